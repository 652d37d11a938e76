#!/bin/bash
TAG=${TAG:-r04_ab}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; }
run cfg2 cfg2 X=1
run cfg2_wg7 cfg2 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_twg7.so
run cfg2n cfg2n X=1
run cfg2n_wg7 cfg2n EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_twg7.so
run cfg5 cfg5 X=1
run cfg5_wg7 cfg5 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_twg7.so
run cfg2_again cfg2 X=1
run cfg2_wg7_again cfg2 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_twg7.so
echo done
