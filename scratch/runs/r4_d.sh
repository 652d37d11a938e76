#!/bin/bash
# round 4, pass d: walking lean CX kernel, arguments copied once per tile
TAG=${TAG:-r04_d}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; }
for K in 0 1 2 3 4 6 8; do run cfg2_k$K cfg2 EPIHIP_CX_WALK=$K; done
for K in 0 4; do run cfg2n_k$K cfg2n EPIHIP_CX_WALK=$K; done
timeout -k 10 900 python -m pytest tests/test_gpu_variants.py -m gpu -q -x > gpurun_out/$TAG/tests_var.log 2>&1; echo "variants rc=$? $(tail -1 gpurun_out/$TAG/tests_var.log)"
echo r4_d done
