#!/bin/bash
TAG=${TAG:-r04_az}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; w=$3; k=$4; timeout -k 10 300 python bench.py --workload $wl --steps $k --warmup $w --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], d["config"]["setup"][:12])' 2>&1 | tail -n 1)"; }
run w2_k10 cfg2 2 10
run w3_k20 cfg2 3 20
run w300 cfg2 300 20
run w2_k10b cfg2 2 10
run cfg4 cfg4 2 10
run cfg5 cfg5 2 10
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --backend gloo --steps 3 --no-extras --rows 2000000 > gpurun_out/$TAG/gloo2.json 2> gpurun_out/$TAG/gloo2.err; echo "gloo2 rc=$? $(tail -n 1 gpurun_out/$TAG/gloo2.json | cut -c1-150)"
timeout -k 10 500 python bench.py > gpurun_out/$TAG/bench_cfg2.json 2> gpurun_out/$TAG/bench_cfg2.err; echo "default rc=$? $(tail -n 1 gpurun_out/$TAG/bench_cfg2.json | cut -c1-200)"
echo done
