#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_variants.py -x -q -m gpu > gpurun_out/t35.log 2>&1; tail -4 gpurun_out/t35.log
grep -q "passed" gpurun_out/t35.log || exit 1
grep -q "failed\|VIOLATION\|Aborted" gpurun_out/t35.log && exit 1
run() { timeout -k 10 280 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/b35.log 2>&1; tail -1 gpurun_out/b35.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
run base
