#!/bin/bash
TAG=${TAG:-r04_ap}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -n 1)"; }
run cfg2 cfg2 X=1
run cfg2_t4096 cfg2 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt4096w512_0.so
run cfg2n cfg2n X=1
run cfg2n_t4096 cfg2n EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt4096w512_0.so
run cfg5 cfg5 X=1
run cfg5_t4096 cfg5 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tt4096w512_0.so
echo done
