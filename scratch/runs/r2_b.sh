#!/bin/bash
# new CX tile kernel (position-aligned u8 counters, fused thresholding): parity suite, then bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r2b_tests.log 2>&1
rc=$?; echo "parity rc=$rc"; tail -25 gpurun_out/r2b_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2b_full.log 2>&1
rc=$?; echo "fullsize rc=$rc"; tail -15 gpurun_out/r2b_full.log
timeout -k 10 300 python bench.py --steps 10 > gpurun_out/r2b_bench1.log 2>&1
echo "bench1 rc=$?"; tail -1 gpurun_out/r2b_bench1.log
timeout -k 10 300 python bench.py --steps 10 --workload cfg2cx --no-extras --cpu-sample 0 > gpurun_out/r2b_bench_cx.log 2>&1
echo "benchcx rc=$?"; tail -1 gpurun_out/r2b_bench_cx.log
timeout -k 10 300 python bench.py --steps 5 --workload cfg5 --no-extras --cpu-sample 0 > gpurun_out/r2b_bench5.log 2>&1
echo "bench5 rc=$?"; tail -1 gpurun_out/r2b_bench5.log
