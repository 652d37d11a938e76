#!/bin/bash
# randomized GPU-vs-oracle runs over the round's final kernels and their fallback variants
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 260 python scratch/fuzz.py 200 110000 > gpurun_out/fuzz_f1.log 2>&1; tail -1 gpurun_out/fuzz_f1.log
EPIHIP_HEAVY_ROWS=200 EPIHIP_CX_SLOT=5 EPIHIP_MHL_SLOT=3 timeout -k 10 260 python scratch/fuzz.py 150 120000 > gpurun_out/fuzz_f2.log 2>&1; tail -1 gpurun_out/fuzz_f2.log
EPIHIP_PR_WIDE=0 EPIHIP_CX_SLOT=0 EPIHIP_MHL_SLOT=0 timeout -k 10 260 python scratch/fuzz.py 100 130000 > gpurun_out/fuzz_f3.log 2>&1; tail -1 gpurun_out/fuzz_f3.log
EPIHIP_PR_RPG=2 EPIHIP_CX_TILE=2048 EPIHIP_CX_WG=1024 EPIHIP_HEAVY_ROWS=700 timeout -k 10 260 python scratch/fuzz.py 100 140000 > gpurun_out/fuzz_f4.log 2>&1; tail -1 gpurun_out/fuzz_f4.log
