#!/bin/bash
TAG=${TAG:-r03_r}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
BENCH_ARGS="--workload cfg2cx" bash scratch/pmc2.sh ${TAG}_cfg2cx "p1 p2 p3 p4" > gpurun_out/$TAG/pmc_cfg2cx.log 2>&1; cp gpurun_out/pmc_${TAG}_cfg2cx/summary.txt gpurun_out/$TAG/pmc_cfg2cx.txt; grep "cxp_tiles\|cx_gather" gpurun_out/$TAG/pmc_cfg2cx.txt | cut -c1-30,50-120
rm -rf gpurun_out/pmc_${TAG}_*
echo $TAG done
