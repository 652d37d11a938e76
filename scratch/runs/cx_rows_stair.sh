#!/bin/bash
# does k_cx_tiles time follow ceil(rows per tile / 64)?  (rounds of 8 waves x 8 rows)
cd $GRAFT_REPO_ROOT
for rows in 7000000 8500000 9000000 9500000 10000000 10500000 12000000 14000000; do
  timeout -k 10 120 python bench.py --workload cfg2cx --rows $rows --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_all']['cx_tiles']; print('rows=$rows', d['ms_per_step'], k, 'us/Mrow', round(k*1e3/($rows/1e6),2))" || exit 1
done
