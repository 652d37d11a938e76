#!/bin/bash
# which LDS operations of k_cx_tiles conflict: PMC pass p2 on timing builds that leave one phase out
TAG=${TAG:-r04_u}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
bash scratch/pmc2.sh ${TAG}_full "p2" > gpurun_out/$TAG/full.log 2>&1; grep -E "cx_tiles" gpurun_out/pmc_${TAG}_full/summary.txt > gpurun_out/$TAG/full.txt
for A in 2 16 32; do
  EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_ta$A.so BENCH_ARGS="--no-selfcheck" bash scratch/pmc2.sh ${TAG}_a$A "p2" > gpurun_out/$TAG/a$A.log 2>&1; grep -E "cx_tiles" gpurun_out/pmc_${TAG}_a$A/summary.txt > gpurun_out/$TAG/a$A.txt
done
rm -rf gpurun_out/pmc_${TAG}_*
for f in full a2 a16 a32; do echo == $f; sed 's/^.\{50\}//' gpurun_out/$TAG/$f.txt; done
