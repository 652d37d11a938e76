#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t30.log 2>&1; tail -5 gpurun_out/t30.log
for wl in cfg4 cfg2; do timeout -k 10 280 python bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/b30_$wl.log 2>&1; tail -1 gpurun_out/b30_$wl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:20], d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'], d['roofline']['frac'])"; done
