#!/bin/bash
# does k_cx_tiles run faster per row when the batch fits the 256 MB Infinity Cache (re-read right after thresholding)?
cd $GRAFT_REPO_ROOT
for rows in 400000 600000 800000 1200000 2000000 4000000 10000000; do
  timeout -k 10 120 python bench.py --workload cfg2 --rows $rows --steps 30 --warmup 5 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_all']; print('rows=$rows MB=%d' % ($rows*300/1e6), d['ms_per_step'], k, 'us/Mrow thr %.1f cx %.1f' % (k['threshold']*1e3/($rows/1e6), k['cx_tiles']*1e3/($rows/1e6)))" || exit 1
done
