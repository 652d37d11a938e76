#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --workload $1 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b25.log 2>&1; tail -1 gpurun_out/b25.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
for wl in cfg2 cfg2cx cfg5; do
EPIHIP_CX_TILE=1024 EPIHIP_CX_WG=512 run $wl "pk T1024 wg512"
EPIHIP_CX_TILE=512 EPIHIP_CX_WG=512 run $wl "pk T512 wg512"
EPIHIP_CX_TILE=512 EPIHIP_CX_WG=256 run $wl "pk T512 wg256"
EPIHIP_CX_TILE=1024 EPIHIP_CX_WG=256 run $wl "pk T1024 wg256"
done
EPIHIP_CX_TILE=1024 EPIHIP_CX_WG=512 EPIHIP_CX_GROUP=16 run cfg2 "pk T1024 wg512 g16"
EPIHIP_CX_TILE=1024 EPIHIP_CX_WG=512 EPIHIP_CX_GROUP=64 run cfg5 "pk T1024 wg512 g64"
EPIHIP_CX_TILE=1024 EPIHIP_CX_WG=512 EPIHIP_CX_GROUP=16 run cfg5 "pk T1024 wg512 g16"
