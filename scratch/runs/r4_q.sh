#!/bin/bash
TAG=${TAG:-r04_q}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; }
run cfg2_block cfg2 EPIHIP_SPIN_US=0
run cfg2_spin cfg2 X=1
run cfg2_block2 cfg2 EPIHIP_SPIN_US=0
run cfg2_spin2 cfg2 X=1
run cfg4_block cfg4 EPIHIP_SPIN_US=0
run cfg4_spin cfg4 EPIHIP_SPIN_US=20000
echo done
