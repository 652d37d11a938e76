#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -m gpu > gpurun_out/t_q.log 2>&1; tail -2 gpurun_out/t_q.log
grep -q "failed\|VIOLATION\|Aborted" gpurun_out/t_q.log && exit 1
timeout -k 10 300 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4', d['ms_per_step'], d['roofline']['kernel_ms_all'])"
timeout -k 10 400 python bench.py --workload cfg4 --rows 1000000 --read-len 10000 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lr-lmhl', d['ms_per_step'], d['roofline']['kernel_ms_all'])"
