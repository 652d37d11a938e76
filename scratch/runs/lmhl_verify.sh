#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_v.log 2>&1; tail -3 gpurun_out/t_v.log
grep -q "failed\|VIOLATION\|Aborted" gpurun_out/t_v.log && exit 1
timeout -k 10 400 python scratch/fuzz.py 240 200000 > gpurun_out/fuzz6.log 2>&1; tail -1 gpurun_out/fuzz6.log
EPIHIP_HEAVY_ROWS=300 timeout -k 10 300 python scratch/fuzz.py 120 300000 > gpurun_out/fuzz7.log 2>&1; tail -1 gpurun_out/fuzz7.log
