#!/bin/bash
# per-read kernels on long and short reads: wide layout against the 2-lane one
cd $GRAFT_REPO_ROOT
for rl in 10000 2000 600 100 50; do
  rows=$((3000000000 / rl)); if [ $rows -gt 20000000 ]; then rows=20000000; fi
  for w in 0 1; do
    EPIHIP_PR_WIDE=$w timeout -k 10 120 python bench.py --workload cfg2 --read-len $rl --rows $rows --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('L=$rl rows=$rows wide=$w', d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
  done
done
