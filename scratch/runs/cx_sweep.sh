#!/bin/bash
# sweep of tile size / workgroup size for k_cx_tiles on cfg2 and cfg2cx
cd $GRAFT_REPO_ROOT
for cfg in "1024 512" "1024 256" "1024 1024" "512 256" "512 512" "2048 512" "2048 1024"; do
  set -- $cfg
  for wl in cfg2 cfg2cx; do
    EPIHIP_CX_TILE=$1 EPIHIP_CX_WG=$2 timeout -k 10 120 python bench.py --workload $wl --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('T=$1 WG=$2', d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
  done
done
