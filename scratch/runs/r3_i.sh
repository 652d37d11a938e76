#!/bin/bash
TAG=${TAG:-r03_i}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
L=$R/epialleler_amd/csrc
BENCH_ARGS="--workload cfg4" bash scratch/pmc2.sh ${TAG}_pipe "p1 p2 p5" > gpurun_out/$TAG/pmc_pipe.log 2>&1; grep -i "mhl_fused" gpurun_out/pmc_${TAG}_pipe/summary.txt | cut -c30-120
export EPIHIP_LIB=$L/libepihip_tnofold.so
BENCH_ARGS="--workload cfg4" bash scratch/pmc2.sh ${TAG}_nofold "p1 p2 p5" > gpurun_out/$TAG/pmc_nofold.log 2>&1; grep -i "mhl_fused" gpurun_out/pmc_${TAG}_nofold/summary.txt | cut -c30-120
cp gpurun_out/pmc_${TAG}_*/summary.txt gpurun_out/$TAG/ 2>/dev/null
rm -rf gpurun_out/pmc_${TAG}_*/p?
echo r3_i done
