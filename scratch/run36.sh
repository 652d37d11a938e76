#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 280 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/b36.log 2>&1; tail -1 gpurun_out/b36.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
EPIHIP_MHL_ABLATE=32 run launch_only
EPIHIP_MHL_ABLATE=80 run zero_only
EPIHIP_MHL_ABLATE=30 run zero_rowloop
EPIHIP_MHL_ABLATE=64 run zero_emit
