#!/bin/bash
# EPIHIP_CX_ABLATE: 1 skip accumulate, 2 skip emit, 4 loads only (6 = loads only, no emit); T = $1 (default 1024), bits = $2
cd $GRAFT_REPO_ROOT
T=${1:-1024}
for ab in ${2:-0 1 2 4 6}; do
  EPIHIP_CX_TILE=$T EPIHIP_CX_ABLATE=$ab timeout -k 10 120 python bench.py --workload cfg2cx --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('T=$T ablate=$ab', d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
done
