#!/bin/bash
b() { echo "$@"; env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
for cfg in "8 1" "10 2" "12 2" "20 1" "5 4" "6 4"; do
  set -- $cfg
  (cd epialleler_amd/csrc && rm -f per_read.o && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -ffp-contract=off -DEPI_PR_UN=$1" libepihip.so > /dev/null 2>&1)
  b EPIHIP_GROUP=$2 UN=$1
done
