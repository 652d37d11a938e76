#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
EPIHIP_HEAVY_ROWS=200 timeout -k 10 300 python scratch/fuzz.py 150 50000 > gpurun_out/fuzz2.log 2>&1; tail -1 gpurun_out/fuzz2.log
EPIHIP_MHL_MULTI=1 EPIHIP_HEAVY_ROWS=900 timeout -k 10 300 python scratch/fuzz.py 150 60000 > gpurun_out/fuzz3.log 2>&1; tail -1 gpurun_out/fuzz3.log
EPIHIP_CX_PACKED=0 EPIHIP_CX_TILE=2048 timeout -k 10 300 python scratch/fuzz.py 100 70000 > gpurun_out/fuzz4.log 2>&1; tail -1 gpurun_out/fuzz4.log
EPIHIP_CX_TILE=512 EPIHIP_CX_WG=256 EPIHIP_CX_GROUP=16 timeout -k 10 300 python scratch/fuzz.py 100 80000 > gpurun_out/fuzz5.log 2>&1; tail -1 gpurun_out/fuzz5.log
