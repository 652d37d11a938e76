#!/bin/bash
TAG=${TAG:-r04_f}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; }
run base cfg2 EPIHIP_CX_WALK=0
run pad4k_nowalk cfg2 EPIHIP_CX_WALK=0 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tpad4k_0.so
for K in 1 2 4 8; do run walk$K cfg2 EPIHIP_CX_WALK=$K; done
echo r4_f done
