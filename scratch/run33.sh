#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "test_all_byte_values" > gpurun_out/t33.log 2>&1; grep -v "^  File" gpurun_out/t33.log | head -30
