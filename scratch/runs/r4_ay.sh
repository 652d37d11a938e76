#!/bin/bash
TAG=${TAG:-r04_ay}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; w=$3; k=$4; timeout -k 10 300 python bench.py --workload $wl --steps $k --warmup $w --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"])' 2>&1 | tail -n 1)"; }
run w3 cfg2 3 20
run w300 cfg2 300 20
run w3b cfg2 3 20
run w1000 cfg2 1000 20
run w3_k200 cfg2 3 200
run w2_k10 cfg2 2 10
run w3000_k200 cfg2 3000 200
echo done
