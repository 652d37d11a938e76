#!/bin/bash
# wide per-read kernel: parity, then speed per reads-per-lane-group (EPIHIP_PR_RPG) against the 2-lane layout
cd $GRAFT_REPO_ROOT
for r in 2 3 4; do
EPIHIP_PR_RPG=$r timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_bed.py -x -q -m gpu > gpurun_out/t_prw.log 2>&1; tail -1 gpurun_out/t_prw.log
grep -q "failed\|VIOLATION\|Aborted\|error" gpurun_out/t_prw.log && exit 1
done
for cfg in "0 4" "1 2" "1 3" "1 4"; do
  set -- $cfg
  for wl in cfg2 cfg3; do
    EPIHIP_PR_WIDE=$1 EPIHIP_PR_RPG=$2 timeout -k 10 120 python bench.py --workload $wl --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('wide=$1 rpg=$2', d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
  done
done
