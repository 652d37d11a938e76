#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/q_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/q_tests.log
[ $rc -eq 0 ] || exit $rc
for w in cfg2 cfg2cx; do
  timeout -k 10 300 python bench.py --steps 10 --workload $w --no-extras --cpu-sample 0 > gpurun_out/q_$w.log 2>&1
  echo "$w: $(tail -1 gpurun_out/q_$w.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"])' 2>&1 | tail -1)"
done
