#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "1024 512" "512 256" "512 512" "2048 1024"; do
  set -- $cfg
  echo "== T=$1 WG=$2"
  EPIHIP_CX_DIAG=1 EPIHIP_CX_TILE=$1 EPIHIP_CX_WG=$2 timeout -k 10 120 python bench.py --workload cfg2cx --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep "cx diag" | tail -2
done
