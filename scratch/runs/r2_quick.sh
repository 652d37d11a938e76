#!/bin/bash
# quick loop: CX/lMHL parity, then the bench lines of the main workloads
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -m gpu > gpurun_out/q_tests.log 2>&1
rc=$?; echo "parity rc=$rc"; tail -4 gpurun_out/q_tests.log
[ $rc -eq 0 ] || exit $rc
for w in ${WL:-cfg2 cfg2cx cfg2u cfg4 cfg5}; do
  timeout -k 10 300 python bench.py --steps 10 --workload $w --no-extras --cpu-sample 0 > gpurun_out/q_$w.log 2>&1
  echo "$w: $(tail -1 gpurun_out/q_$w.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"])' 2>&1 | tail -1)"
done
