#!/bin/bash
# lMHL at other template lengths (the one-block lane shapes at 256 threads): same bytes per batch, rows scaled
TAG=${TAG:-r03_w}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
for spec in "100 100000000" "150 80000000" "200 60000000" "250 50000000" "500 25000000" "1000 12000000"; do
  set -- $spec
  timeout -k 10 250 python bench.py --workload cfg4 --read-len $1 --rows $2 --steps 3 --warmup 1 --no-extras --cpu-sample 0 > gpurun_out/$TAG/L$1.json 2> gpurun_out/$TAG/L$1.err
  echo "L=$1 rows=$2: $(tail -1 gpurun_out/$TAG/L$1.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["kernel_ms_all"], round(r["frac"],3))' 2>&1 | tail -1)"
done
echo $TAG done
