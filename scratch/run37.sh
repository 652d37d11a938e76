#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t37.log 2>&1; tail -4 gpurun_out/t37.log
grep -q "failed\|VIOLATION\|Aborted" gpurun_out/t37.log && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for wl in cfg2 cfg2cx cfg4 cfg5; do timeout -k 10 280 python bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/b37_$wl.log 2>&1; tail -1 gpurun_out/b37_$wl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:20], d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'], d['roofline']['frac'])"; done
