#!/bin/bash
# full GPU suite, then the measurement pass
TAG=${TAG:-r03_d}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/$TAG/gpu_tests.txt 2>&1; rc=$?; echo "gpu tests rc=$rc $(tail -1 gpurun_out/$TAG/gpu_tests.txt)"
[ $rc -eq 0 ] || exit 1
TAG=$TAG bash scratch/runs/r3_measure.sh
