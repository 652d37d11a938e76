#!/bin/bash
TAG=${TAG:-r04_aw}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -n 1)"; }
for rep in 1 2; do
for m in 16 8 4; do run cfg2_m${m}_$rep cfg2 EPIHIP_REALIGN=$m; done
done
for m in 16 8 4; do run cfg2n_m$m cfg2n EPIHIP_REALIGN=$m; done
for m in 16 8 4; do run cfg2u_m$m cfg2u EPIHIP_REALIGN=$m; done
for m in 16 8 4; do run cfg4_m$m cfg4 EPIHIP_REALIGN=$m; done
for m in 16 4; do run cfg5_m$m cfg5 EPIHIP_REALIGN=$m; done
echo done
