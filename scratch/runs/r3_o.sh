#!/bin/bash
TAG=${TAG:-r03_o}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
L=$R/epialleler_amd/csrc
BENCH_ARGS="--workload cfg2" bash scratch/pmc2.sh ${TAG}_base "p2" > gpurun_out/$TAG/pmc_base.log 2>&1; echo "== base"; grep -i "cx_tiles" gpurun_out/pmc_${TAG}_base/summary.txt | grep "INSTS_VALU\|INSTS_LDS \|INSTS_SALU" | cut -c50-120
for v in cx8 cx4 cx2 cx6 cx38; do
export EPIHIP_LIB=$L/libepihip_t$v.so
BENCH_ARGS="--workload cfg2" bash scratch/pmc2.sh ${TAG}_$v "p2" > gpurun_out/$TAG/pmc_$v.log 2>&1; echo "== $v"; grep -i "cx_tiles" gpurun_out/pmc_${TAG}_$v/summary.txt | grep "INSTS_VALU\|INSTS_LDS \|INSTS_SALU" | cut -c50-120
done
rm -rf gpurun_out/pmc_${TAG}_*/p?
echo $TAG done
