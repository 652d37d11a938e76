#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r2d_tests.log 2>&1
rc=$?; echo "parity rc=$rc"; tail -12 gpurun_out/r2d_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 5 --workload cfg4 --no-extras --cpu-sample 0 > gpurun_out/r2d_bench4.log 2>&1
echo "bench4 rc=$?"; tail -1 gpurun_out/r2d_bench4.log
EPIHIP_MHL_FUSED=0 timeout -k 10 300 python bench.py --steps 5 --workload cfg4 --no-extras --cpu-sample 0 > gpurun_out/r2d_bench4old.log 2>&1
echo "bench4old rc=$?"; tail -1 gpurun_out/r2d_bench4old.log | cut -c1-400
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2d_full.log 2>&1
rc=$?; echo "fullsize rc=$rc"; tail -5 gpurun_out/r2d_full.log
