#!/bin/bash
cd $GRAFT_REPO_ROOT
for ab in 0 1 2 4 6; do
  EPIHIP_CX_ABLATE=$ab timeout -k 10 120 python bench.py --workload cfg2cx --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ablate=$ab', d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
done
