#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bed.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/t44.log 2>&1; tail -4 gpurun_out/t44.log
grep -q "failed\|VIOLATION\|Aborted" gpurun_out/t44.log && exit 1
run() { timeout -k 10 200 python bench.py --workload cfg2 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b44.log 2>&1; tail -1 gpurun_out/b44.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
run lds
EPIHIP_PER_READ=group run group
