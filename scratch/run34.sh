#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_all_byte_values" > gpurun_out/t34.log 2>&1; grep -v "^  File" gpurun_out/t34.log | grep -i "check\|passed\|failed\|Aborted\|VIOLATION" | head -10
