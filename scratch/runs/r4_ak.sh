#!/bin/bash
TAG=${TAG:-r04_ak}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
NIB=$R/epialleler_amd/csrc/libepihip_tnib.so
EPIHIP_LIB=$NIB timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/$TAG/tests.log 2>&1
rc=$?; echo "tests(nib) rc=$rc $(tail -n 1 gpurun_out/$TAG/tests.log)"
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -n 1)"; }
for wl in cfg2n cfg5 cfg2cx; do
run ${wl} $wl X=1
run ${wl}_nib $wl EPIHIP_LIB=$NIB
done
run cfg2n_again cfg2n X=1
run cfg2n_nib_again cfg2n EPIHIP_LIB=$NIB
echo done
