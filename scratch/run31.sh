#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 280 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/b31.log 2>&1; tail -1 gpurun_out/b31.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
run base
EPIHIP_MHL_ABLATE=1 run noatomic
EPIHIP_MHL_ABLATE=2 run norecs
EPIHIP_MHL_ABLATE=6 run norecs_nointerval
EPIHIP_MHL_ABLATE=14 run noaccum
EPIHIP_MHL_ABLATE=16 run noemit
EPIHIP_MHL_ABLATE=30 run onlyzero
