#!/bin/bash
TAG=${TAG:-r04_bb}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
Z=$R/epialleler_amd/csrc/libepihip_tzbuf.so
EPIHIP_LIB=$Z timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$TAG/tests.log 2>&1
rc=$?; echo "tests(zbuf) rc=$rc $(tail -n 1 gpurun_out/$TAG/tests.log)"
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 30 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -n 1)"; }
for rep in 1 2; do
run cfg2_p$rep cfg2 X=1
run cfg2_z$rep cfg2 EPIHIP_LIB=$Z
done
run cfg2n_p cfg2n X=1
run cfg2n_z cfg2n EPIHIP_LIB=$Z
run cfg5_p cfg5 X=1
run cfg5_z cfg5 EPIHIP_LIB=$Z
run cfg2u_p cfg2u X=1
run cfg2u_z cfg2u EPIHIP_LIB=$Z
echo done
