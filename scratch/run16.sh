#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout=450 > gpurun_out/t3.log 2>&1; echo exit=$? >> gpurun_out/t3.log; tail -3 gpurun_out/t3.log
b() { echo "$@"; env "$@" timeout -k 10 120 python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
b EPIHIP_CX_PERSIST=0
b EPIHIP_CX_PERSIST=1 EPIHIP_CX_GROUP=8
b EPIHIP_CX_PERSIST=1 EPIHIP_CX_GROUP=16
b EPIHIP_CX_PERSIST=1 EPIHIP_CX_GROUP=32
b EPIHIP_CX_PERSIST=1 EPIHIP_CX_WG=512 EPIHIP_CX_GROUP=8
b EPIHIP_CX_PERSIST=1 EPIHIP_CX_WG=512 EPIHIP_CX_GROUP=16
