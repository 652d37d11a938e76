#!/bin/bash
# software-pipelined k_cx_tiles: parity, then speed per (WG, D)
cd $GRAFT_REPO_ROOT
EPIHIP_CX_PIPE=2 EPIHIP_CX_WG=256 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/t_pipe.log 2>&1; tail -3 gpurun_out/t_pipe.log
grep -q "failed\|VIOLATION\|Aborted\|error" gpurun_out/t_pipe.log && exit 1
for cfg in "512 0" "256 2" "256 3" "512 2"; do
  set -- $cfg
  for wl in cfg2 cfg2cx cfg5; do
    EPIHIP_CX_WG=$1 EPIHIP_CX_PIPE=$2 timeout -k 10 120 python bench.py --workload $wl --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('WG=$1 D=$2', d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
  done
done
