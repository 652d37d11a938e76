#!/bin/bash
# builds k_cx_tiles with the LDS atomics or the VALU work compiled out (timing experiments) and times the accumulate phase
cd $GRAFT_REPO_ROOT/epialleler_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -ffp-contract=off"
for abl in 1 2 0; do
  rm -f cx_report.o; make -j8 libepihip.so CXXFLAGS="$FL -DEPI_CX_ABL=$abl" > $GRAFT_REPO_ROOT/gpurun_out/abl_build_$abl.log 2>&1 || exit 1
  for ab in 2 0; do
  ( cd $GRAFT_REPO_ROOT && EPIHIP_CX_ABLATE=$ab timeout -k 10 120 python bench.py --workload cfg2cx --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ABL=$abl ablate=$ab', d['ms_per_step'], d['roofline']['kernel_ms_all'])" ) || exit 1
  done
done
