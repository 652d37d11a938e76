#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t45.log 2>&1; tail -2 gpurun_out/t45.log
for wl in cfg2 cfg2cx cfg5; do timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b45.log 2>&1; tail -1 gpurun_out/b45.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; done
