#!/bin/bash
# timing builds of the CX tile kernel on cfg2 (and cfg2cx): kernel ms per build
mkdir -p gpurun_out
W=${W:-cfg2}
for a in 0 $ABL; do
  lib=epialleler_amd/csrc/libepihip_t$a.so
  [ "$a" = 0 ] && lib=epialleler_amd/csrc/libepihip.so
  EPIHIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 10 --workload $W --no-extras --cpu-sample 0 > gpurun_out/abl_$a.log 2>&1
  echo "ablate $a: $(tail -1 gpurun_out/abl_$a.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)"
done
