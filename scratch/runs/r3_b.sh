#!/bin/bash
# round 3: the rewritten one-pass lMHL kernel -- parity first, then cfg4 / cfg4d
TAG=${TAG:-r03_b}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/$TAG/t_parity.log 2>&1; echo "parity rc=$? $(tail -1 gpurun_out/$TAG/t_parity.log)"
timeout -k 10 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_sharded.py tests/test_gpu_bed.py tests/test_gpu_patterns.py tests/test_shim_core.py -m gpu -q > gpurun_out/$TAG/t_var.log 2>&1; echo "variants rc=$? $(tail -1 gpurun_out/$TAG/t_var.log)"
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q > gpurun_out/$TAG/t_full.log 2>&1; echo "fullsize rc=$? $(tail -1 gpurun_out/$TAG/t_full.log)"
for wl in cfg4 cfg4d; do timeout -k 10 280 python bench.py --workload $wl --steps 5 --warmup 1 --no-extras --cpu-sample 0 > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err; echo "$wl rc=$?"; tail -1 gpurun_out/$TAG/bench_$wl.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_all"])'; done
echo r3_b done
