#!/bin/bash
# round 4, pass c: where the walking CX kernel loses -- occupancy alone (5 waves/SIMD, no walk), walk of 1, T = 1024 walks
TAG=${TAG:-r04_c}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --workload cfg2 --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; }
run base EPIHIP_CX_WALK=0
run walk1 EPIHIP_CX_WALK=1
run wps5_nowalk EPIHIP_CX_WALK=0 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_twps5_0.so
run t1024_nowalk EPIHIP_CX_WALK=0 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt1024_0.so
run t1024_walk4 EPIHIP_CX_WALK=4 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt1024_0.so
run t1024_walk8 EPIHIP_CX_WALK=8 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt1024_0.so
run t1024_walk16 EPIHIP_CX_WALK=16 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt1024_0.so
echo r4_c done
