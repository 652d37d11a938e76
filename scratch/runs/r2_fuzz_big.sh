#!/bin/bash
# long randomised run weighted to large batches (uploads of 4-90 MiB, thousands of tiles, remembered tile index on the second
# report of every batch) on the round's final library
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
FUZZ_BIG=1 timeout -k 10 520 python scratch/fuzz.py 460 310000 > gpurun_out/fuzz_big_1.log 2>&1; tail -1 gpurun_out/fuzz_big_1.log
FUZZ_BIG=1 EPIHIP_HEAVY_ROWS=300 EPIHIP_CX_SLOT=7 timeout -k 10 520 python scratch/fuzz.py 460 320000 > gpurun_out/fuzz_big_2.log 2>&1; tail -1 gpurun_out/fuzz_big_2.log
