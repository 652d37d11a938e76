#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_qa.log 2>&1; tail -2 gpurun_out/t_qa.log
grep -q "failed\|VIOLATION\|Aborted" gpurun_out/t_qa.log && exit 1
for wl in cfg2 cfg2cx cfg4; do timeout -k 10 280 python bench.py --workload $wl --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; done
