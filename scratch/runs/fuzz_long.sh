#!/bin/bash
# long randomized GPU-vs-oracle runs on the final kernels: default switches, big uploads, pile-up splitting
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
S=${1:-410000}
FUZZ_BIG=1 timeout -k 10 400 python scratch/fuzz.py 360 $S > gpurun_out/fuzz_m1.log 2>&1; tail -1 gpurun_out/fuzz_m1.log
EPIHIP_HEAVY_ROWS=150 EPIHIP_CX_SLOT=7 EPIHIP_MHL_SLOT=5 timeout -k 10 300 python scratch/fuzz.py 260 $((S+10000)) > gpurun_out/fuzz_m2.log 2>&1; tail -1 gpurun_out/fuzz_m2.log
EPIHIP_MHL_MULTI=1 EPIHIP_MHL_SUMS=64 timeout -k 10 300 python scratch/fuzz.py 260 $((S+20000)) > gpurun_out/fuzz_m3.log 2>&1; tail -1 gpurun_out/fuzz_m3.log
