#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --workload $2 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b42.log 2>&1; tail -1 gpurun_out/b42.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
run normal cfg2
EPIHIP_CX_ABLATE=4 run fixedslot cfg2
run normal cfg2cx
EPIHIP_CX_ABLATE=4 run fixedslot cfg2cx
