#!/bin/bash
# round 4, pass o: walking one-pass lMHL kernel -- variants, parity, A/B of tiles per workgroup on config 4
TAG=${TAG:-r04_o}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-extras --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], (d["selfcheck"] or {}).get("ok"))' 2>&1 | tail -1)"; }
for K in 0 1 2 4 8 16; do run cfg4_k$K cfg4 EPIHIP_MHLF_WALK=$K; done
run cfg4d cfg4d X=1
echo done
