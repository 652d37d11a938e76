#!/bin/bash
TAG=${TAG:-r04_t}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 2 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; tail -2 gpurun_out/$TAG/$name.err; }
run cfg4 cfg4 X=1
run cfg4_t1280_w320 cfg4 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt1280.so
run cfg4_t1280_w256 cfg4 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt1280w256.so
echo done
