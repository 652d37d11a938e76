#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --workload cfg2cx --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b38.log 2>&1; tail -1 gpurun_out/b38.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
run full
EPIHIP_CX_ABLATE=1 run noacc
EPIHIP_CX_ABLATE=2 run noemit
EPIHIP_CX_ABLATE=3 run neither
