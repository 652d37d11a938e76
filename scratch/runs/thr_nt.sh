#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --workload cfg2 --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('nt', d['ms_per_step'], d['roofline']['kernel_ms_all'])"
