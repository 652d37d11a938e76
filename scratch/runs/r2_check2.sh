#!/bin/bash
# the GPU suite once under the index-check build (make check) on the round's final kernels: default paths, then the
# general CX kernel + two-kernel lMHL path
set -o pipefail
mkdir -p gpurun_out
export EPIHIP_LIB=$PWD/epialleler_amd/csrc/libepihip_check.so
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2_check_suite.log 2>&1
echo "check-build suite rc=$?"; tail -3 gpurun_out/r2_check_suite.log
EPIHIP_MHL_FUSED=0 EPIHIP_CX_LEAN=0 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2_check_general.log 2>&1
echo "check-build general CX kernel + two-kernel lMHL rc=$?"; tail -3 gpurun_out/r2_check_general.log
