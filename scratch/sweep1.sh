#!/bin/bash
# usage: sweep1.sh  (runs on the GPU box)
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout=450 > gpurun_out/t3.log 2>&1; echo exit=$? >> gpurun_out/t3.log; tail -4 gpurun_out/t3.log
for T in 512 1024 2048; do for G in 8 16 32; do
  echo "T=$T G=$G" >> gpurun_out/sweep1.log
  EPIHIP_CX_TILE=$T EPIHIP_CX_GROUP=$G timeout -k 10 120 python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'])" >> gpurun_out/sweep1.log 2>&1
done; done
cat gpurun_out/sweep1.log
