#!/bin/bash
TAG=${TAG:-r03_j}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
L=$R/epialleler_amd/csrc
for v in ab1 ab3 ab4 ab8 ab13; do
export EPIHIP_LIB=$L/libepihip_t$v.so
BENCH_ARGS="--workload cfg4" bash scratch/pmc2.sh ${TAG}_$v "p2" > gpurun_out/$TAG/pmc_$v.log 2>&1; echo "== $v"; grep -i "mhl_fused" gpurun_out/pmc_${TAG}_$v/summary.txt | grep "INSTS_VALU\|INSTS_LDS \|INSTS_SALU" | cut -c50-120
done
rm -rf gpurun_out/pmc_${TAG}_*/p?
echo r3_j done
