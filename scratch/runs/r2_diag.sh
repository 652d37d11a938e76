#!/bin/bash
for a in $ABL; do
  lib=epialleler_amd/csrc/libepihip_t$a.so
  EPIHIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 5 --workload ${W:-cfg2} --no-extras --cpu-sample 0 > gpurun_out/diag_$a.log 2>&1
  echo "== $a: $(tail -1 gpurun_out/diag_$a.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)"
  grep "cx diag" gpurun_out/diag_$a.log | tail -2
done
