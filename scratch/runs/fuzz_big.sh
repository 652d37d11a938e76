#!/bin/bash
# randomized runs with every fourth case an upload of 4..90 MiB (threaded staging copy, 64 MiB chunks)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
FUZZ_BIG=1 timeout -k 10 560 python scratch/fuzz.py 500 310000 > gpurun_out/fuzz_b1.log 2>&1; tail -1 gpurun_out/fuzz_b1.log
