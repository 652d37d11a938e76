#!/bin/bash
# k_mhl_tiles timing builds: EPIHIP_MHL_ABLATE bits 2 no records, 4 no whole-slice intervals, 8 no histogram adds, 16 no emit
cd $GRAFT_REPO_ROOT
for ab in 0 16 8 24 2 6 14 30; do
  EPIHIP_MHL_ABLATE=$ab timeout -k 10 200 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ablate=$ab', d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
done
