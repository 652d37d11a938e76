#!/bin/bash
# the GPU suite once under the index-check build (make check), fused and two-kernel lMHL paths
set -o pipefail
mkdir -p gpurun_out
export EPIHIP_LIB=$PWD/epialleler_amd/csrc/libepihip_check.so
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2_check_suite.log 2>&1
echo "check-build suite rc=$?"; tail -4 gpurun_out/r2_check_suite.log
EPIHIP_MHL_FUSED=0 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2_check_twokernel.log 2>&1
echo "check-build two-kernel lMHL rc=$?"; tail -4 gpurun_out/r2_check_twokernel.log
unset EPIHIP_LIB
timeout -k 10 300 python -m pytest tests/test_shim_core.py -x -q -m gpu > gpurun_out/r2_shim.log 2>&1
echo "shim core rc=$?"; tail -3 gpurun_out/r2_shim.log
timeout -k 10 300 python bench.py --workload file --steps 3 --warmup 1 > gpurun_out/r2_file.log 2>&1
echo "file rc=$?"; tail -1 gpurun_out/r2_file.log
