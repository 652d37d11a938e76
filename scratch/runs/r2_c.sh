#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -m gpu > gpurun_out/r2c_tests.log 2>&1
rc=$?; echo "parity rc=$rc"; tail -12 gpurun_out/r2c_tests.log
[ $rc -eq 0 ] || exit $rc
ABL="$ABL" bash scratch/runs/r2_ablate.sh
W=cfg2cx ABL="" bash scratch/runs/r2_ablate.sh
W=cfg2u ABL="" bash scratch/runs/r2_ablate.sh
W=cfg5 ABL="" bash scratch/runs/r2_ablate.sh
