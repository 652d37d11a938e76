#!/bin/bash
# lMHL pass 1 lane shapes on config 4: G lanes x 16*C bytes (u32 masks for C = 2, u64 for 3 and 4)
cd $GRAFT_REPO_ROOT
for gc in "8,3" "16,2" "8,4" "16,3" "32,2"; do
  EPIHIP_MHL_GROUP=$gc timeout -k 10 200 python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G,C=$gc', d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
done
