#!/bin/bash
TAG=${TAG:-r03_f}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
for wl in cfg2 cfg2p cfg2u; do timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 2 --no-extras --cpu-sample 0 > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err; echo "$wl: $(tail -1 gpurun_out/$TAG/bench_$wl.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)"; done
echo r3_f done
