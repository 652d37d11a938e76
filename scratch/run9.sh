#!/bin/bash
for wl in cfg2cx cfg4 cfg5; do
  echo "== $wl"
  timeout -k 10 280 python bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 200000 > gpurun_out/b_$wl.log 2>&1; echo rc=$?; tail -1 gpurun_out/b_$wl.log | cut -c1-1500
done
