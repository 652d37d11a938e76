#!/bin/bash
# lMHL tile kernel: 256- against 512-thread workgroups (build-time EPI_MHL_WG) on short and long reads
cd $GRAFT_REPO_ROOT/epialleler_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -ffp-contract=off"
for wg in 512 256; do
  rm -f mhl_report.o; make -j8 libepihip.so CXXFLAGS="$FL -DEPI_MHL_WG=$wg" > $GRAFT_REPO_ROOT/gpurun_out/mhl_wg_build.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/mhl_wg_build.log; exit 1; }
  ( cd $GRAFT_REPO_ROOT
    for args in "--rows 1000000 --read-len 10000" "--rows 10000000 --read-len 1000" ""; do timeout -k 10 200 python bench.py --workload cfg4 $args --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('wg=$wg [$args]', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; done ) || exit 1
done
