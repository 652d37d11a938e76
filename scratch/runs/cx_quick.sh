#!/bin/bash
# three back-to-back timings of the two CX workloads (run-to-run spread is ~2 %)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for wl in cfg2 cfg2cx; do timeout -k 10 120 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1; done; done
