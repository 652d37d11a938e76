#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_variants.py -x -q -m gpu > gpurun_out/t_ab.log 2>&1; tail -3 gpurun_out/t_ab.log
grep -q "failed\|VIOLATION\|Aborted" gpurun_out/t_ab.log && exit 1
run() { timeout -k 10 280 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/b_ab.log 2>&1; tail -1 gpurun_out/b_ab.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
run narrow8
EPIHIP_MHL_WPS6=1 run narrow6
EPIHIP_MHL_SUMS=64 run wide
