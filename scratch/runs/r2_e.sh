#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2e_tests.log 2>&1
rc=$?; echo "gpu suite rc=$rc"; tail -6 gpurun_out/r2e_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload file --steps 3 --warmup 1 > gpurun_out/r2e_file.log 2>&1
echo "file rc=$?"; tail -1 gpurun_out/r2e_file.log
timeout -k 10 300 python bench.py --workload file --steps 3 --warmup 1 --host-threads 1 > gpurun_out/r2e_file1.log 2>&1
echo "file1 rc=$?"; tail -1 gpurun_out/r2e_file1.log
