#!/bin/bash
# the GPU suite once under the index-check build (make check): every tile kernel verifies its indices before use
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_check
export EPIHIP_LIB=$PWD/epialleler_amd/csrc/libepihip_check.so
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --deselect tests/test_host_api.py::test_no_timing_switches_in_the_product_library > gpurun_out/r04_check/suite.log 2>&1
echo "check-build suite rc=$? $(tail -n 1 gpurun_out/r04_check/suite.log)"
EPIHIP_MHLF_FOLD=0 EPIHIP_MHLF_FOLD_SLOTS=1 EPIHIP_HEAVY_ROWS=300 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04_check/nofold.log 2>&1
echo "check-build, fold slots exhausted rc=$? $(tail -n 1 gpurun_out/r04_check/nofold.log)"
EPIHIP_MHL_FUSED=0 EPIHIP_CX_LEAN=0 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04_check/twokernel.log 2>&1
echo "check-build, two-kernel lMHL + general CX rc=$? $(tail -n 1 gpurun_out/r04_check/twokernel.log)"
EPIHIP_REALIGN=0 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04_check/realign0.log 2>&1
echo "check-build, rows back to back rc=$? $(tail -n 1 gpurun_out/r04_check/realign0.log)"
