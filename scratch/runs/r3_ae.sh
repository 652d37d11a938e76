#!/bin/bash
TAG=${TAG:-r03_ae}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
BENCH_ARGS="--workload cfg5" bash scratch/pmc2.sh ${TAG}_cfg5 "p1 p2 p3 p4" > gpurun_out/$TAG/pmc_cfg5.log 2>&1; cp gpurun_out/pmc_${TAG}_cfg5/summary.txt gpurun_out/$TAG/pmc_cfg5.txt; grep "k_cx_tiles" gpurun_out/$TAG/pmc_cfg5.txt | cut -c1-20,50-120
rm -rf gpurun_out/pmc_${TAG}_*
echo $TAG done
