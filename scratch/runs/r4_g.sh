#!/bin/bash
TAG=${TAG:-r04_g}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; }
run base cfg2 EPIHIP_CX_WALK=0
run touch cfg2 EPIHIP_CX_WALK=0 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_ttouch_0.so
run touch_walk4 cfg2 EPIHIP_CX_WALK=4 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_ttouch_0.so
run base5 cfg5 EPIHIP_CX_WALK=0
run touch5 cfg5 EPIHIP_CX_WALK=0 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_ttouch_0.so
echo r4_g done
