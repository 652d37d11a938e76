#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 280 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/b_ra.log 2>&1; tail -1 gpurun_out/b_ra.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['roofline']['kernel_ms_all'])"; }
run base
EPIHIP_MHL_ABLATE=2048 EPIHIP_MHL_SUMS=64 run nomaxh
