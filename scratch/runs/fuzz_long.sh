#!/bin/bash
# long randomized GPU-vs-oracle run on the final kernels (default switches), then smoke()
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 420 python scratch/fuzz.py 380 210000 > gpurun_out/fuzz_l1.log 2>&1; tail -1 gpurun_out/fuzz_l1.log
EPIHIP_HEAVY_ROWS=150 timeout -k 10 300 python scratch/fuzz.py 260 220000 > gpurun_out/fuzz_l2.log 2>&1; tail -1 gpurun_out/fuzz_l2.log
