#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout=450 > gpurun_out/t3.log 2>&1; echo exit=$? >> gpurun_out/t3.log; tail -3 gpurun_out/t3.log
b() { echo "$@"; env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
b A=1
for un in 2 8; do
  (cd epialleler_amd/csrc && rm -f per_read.o && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -ffp-contract=off -DEPI_PR_UN=$un" libepihip.so > /dev/null 2>&1)
  for g in 2 4 8; do b EPIHIP_GROUP=$g UN=$un; done
done
