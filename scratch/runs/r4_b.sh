#!/bin/bash
# round 4, pass b: walking lean CX kernel -- variants + parity suite + A/B of tiles per workgroup
TAG=${TAG:-r04_b}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_variants.py -m gpu -q -x -k "WALK or env0" > gpurun_out/$TAG/tests_var.log 2>&1; echo "variants rc=$? $(tail -1 gpurun_out/$TAG/tests_var.log)"
for K in 0 2 4 8 16; do EPIHIP_CX_WALK=$K timeout -k 10 200 python bench.py --workload cfg2 --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/bench_cfg2_k$K.json 2> gpurun_out/$TAG/bench_cfg2_k$K.err; echo "cfg2 walk=$K rc=$?: $(tail -1 gpurun_out/$TAG/bench_cfg2_k$K.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; done
for K in 0 4 8; do EPIHIP_CX_WALK=$K timeout -k 10 200 python bench.py --workload cfg2n --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/bench_cfg2n_k$K.json 2> gpurun_out/$TAG/bench_cfg2n_k$K.err; echo "cfg2n walk=$K rc=$?: $(tail -1 gpurun_out/$TAG/bench_cfg2n_k$K.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; done
echo r4_b done
