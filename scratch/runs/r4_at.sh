#!/bin/bash
TAG=${TAG:-r04_at}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_sharded.py tests/test_gpu_layout.py -x -q -m gpu > gpurun_out/$TAG/tests.log 2>&1
rc=$?; echo "tests rc=$rc $(tail -n 1 gpurun_out/$TAG/tests.log)"
[ $rc -ne 0 ] && { tail -n 40 gpurun_out/$TAG/tests.log; exit 1; }
timeout -k 10 600 python scratch/outlier_cost.py > gpurun_out/$TAG/outlier_cost.txt 2>&1; tail -n 19 gpurun_out/$TAG/outlier_cost.txt
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -n 1)"; }
run cfg2 cfg2 X=1
run cfg2u cfg2u X=1
run cfg2n cfg2n X=1
run cfg5 cfg5 X=1
echo done
