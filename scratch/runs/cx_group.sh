#!/bin/bash
# lanes per row (EPIHIP_CX_GROUP): 8 lanes x 10 dwords (default), 16 x 5, 32 x 3 -- fewer cache-line touches per load instruction
cd $GRAFT_REPO_ROOT
for g in 8 16 32; do
  for ab in 0 6; do
    EPIHIP_CX_GROUP=$g EPIHIP_CX_ABLATE=$ab timeout -k 10 120 python bench.py --workload cfg2cx --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G=$g ablate=$ab', d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
  done
done
