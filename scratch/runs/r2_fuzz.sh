#!/bin/bash
# round 2: randomized GPU-vs-oracle runs over the new kernels (u8 CX, fused thresholding, fused lMHL) and their fallbacks
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
B=${B:-170}
timeout -k 10 $((B+60)) python scratch/fuzz.py $B 210000 > gpurun_out/fuzz_r2_1.log 2>&1; tail -1 gpurun_out/fuzz_r2_1.log
EPIHIP_HEAVY_ROWS=200 EPIHIP_CX_SLOT=5 EPIHIP_MHL_SLOT=3 timeout -k 10 $((B+60)) python scratch/fuzz.py $B 220000 > gpurun_out/fuzz_r2_2.log 2>&1; tail -1 gpurun_out/fuzz_r2_2.log
EPIHIP_MHL_FUSED=0 EPIHIP_CX_LEAN=0 EPIHIP_CX_SLOT=0 EPIHIP_MHL_SLOT=0 timeout -k 10 $((B+60)) python scratch/fuzz.py $B 230000 > gpurun_out/fuzz_r2_3.log 2>&1; tail -1 gpurun_out/fuzz_r2_3.log
FUZZ_BIG=1 timeout -k 10 $((B+60)) python scratch/fuzz.py $B 240000 > gpurun_out/fuzz_r2_4.log 2>&1; tail -1 gpurun_out/fuzz_r2_4.log
