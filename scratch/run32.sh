#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/t32.log 2>&1; tail -5 gpurun_out/t32.log
run() { timeout -k 10 280 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/b32.log 2>&1; tail -1 gpurun_out/b32.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
run base
EPIHIP_MHL_ABLATE=2 run norecs
EPIHIP_MHL_ABLATE=6 run norecs_nointerval
EPIHIP_MHL_ABLATE=14 run noaccum
EPIHIP_MHL_ABLATE=16 run noemit
EPIHIP_MHL_ABLATE=30 run onlyzero
