#!/bin/bash
# the wide per-read kernel with its ALU work compiled out (timing experiment)
cd $GRAFT_REPO_ROOT/epialleler_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -ffp-contract=off"
for abl in 1 0; do
  rm -f per_read.o
  if [ $abl = 1 ]; then make -j8 libepihip.so CXXFLAGS="$FL -DEPI_PW_NOALU" > /dev/null 2>&1 || exit 1; else make -j8 libepihip.so > /dev/null 2>&1 || exit 1; fi
  ( cd $GRAFT_REPO_ROOT && timeout -k 10 120 python bench.py --workload cfg2 --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('noalu=$abl', d['ms_per_step'], d['roofline']['kernel_ms_all'])" ) || exit 1
done
