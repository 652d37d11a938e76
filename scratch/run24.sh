#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/t24.log 2>&1; tail -3 gpurun_out/t24.log
run() { timeout -k 10 200 python bench.py --workload $1 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b24.log 2>&1; tail -1 gpurun_out/b24.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
for wl in cfg2 cfg2cx cfg5; do
run $wl "pk T2048 wg1024"
EPIHIP_CX_PACKED=0 run $wl "u32 T1024 wg1024"
EPIHIP_CX_TILE=1024 EPIHIP_CX_WG=512 run $wl "pk T1024 wg512"
EPIHIP_CX_TILE=1024 run $wl "pk T1024 wg1024"
EPIHIP_CX_WG=512 run $wl "pk T2048 wg512"
done
