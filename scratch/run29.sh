#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t29.log 2>&1; tail -25 gpurun_out/t29.log
