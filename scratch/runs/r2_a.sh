#!/bin/bash
# round 2, first GPU call: >4 GiB full-size tests on the round-1 kernels, new generator, bench with the new sub-records
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2a_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r2a_tests.log
tail -5 gpurun_out/r2a_tests.log
timeout -k 10 300 python bench.py --steps 5 > gpurun_out/r2a_bench1.log 2>&1
echo "bench1 rc=$?"; tail -2 gpurun_out/r2a_bench1.log
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --backend gloo --steps 3 --no-extras --rows 2000000 > gpurun_out/r2a_bench2.log 2>&1
echo "bench2 rc=$?"; tail -3 gpurun_out/r2a_bench2.log
