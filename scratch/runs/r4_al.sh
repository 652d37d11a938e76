#!/bin/bash
TAG=${TAG:-r04_al}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
NIB=$R/epialleler_amd/csrc/libepihip_tnib.so
run() { name=$1; wl=$2; L=$3; rows=$4; shift; shift; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --read-len $L --rows $rows --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -n 1)"; }
for spec in "600 5000000" "1200 2500000" "2400 1250000" "150 20000000"; do
set -- $spec
run n_L$1 cfg2n $1 $2 X=1
run n_L$1_nib cfg2n $1 $2 EPIHIP_LIB=$NIB
done
echo done
