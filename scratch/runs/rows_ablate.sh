#!/bin/bash
# k_mhl_rows timing builds: EPIHIP_MHL_ABLATE bits 8 no cursor atomic, 9 no record writes, 10 no span bits, 11 no max_h atomic
cd $GRAFT_REPO_ROOT
for ab in 0 256 512 1024 2048 3840; do
  EPIHIP_MHL_ABLATE=$ab timeout -k 10 200 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ablate=$ab', d['ms_per_step'], d['roofline']['kernel_ms_all'])" || exit 1
done
