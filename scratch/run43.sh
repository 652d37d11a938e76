#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --workload cfg2 --rows 100000000 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/b43.log 2>&1; tail -1 gpurun_out/b43.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3-size', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'], d['roofline']['frac'])"
