#!/bin/bash
TAG=${TAG:-r04_aq}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
W=$R/epialleler_amd/csrc/libepihip_twalk0.so
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -n 1)"; }
run cfg2 cfg2 X=1
run cfg2_w0 cfg2 EPIHIP_LIB=$W
for k in 2 4 8 16; do run cfg2_w$k cfg2 EPIHIP_LIB=$W EPIHIP_CX_WALK=$k; done
run cfg2_again cfg2 X=1
echo done
