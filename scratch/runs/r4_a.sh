#!/bin/bash
# round 4, pass a: new tests (bench-stream full-size parity, empty-rank shards) + self-validating bench lines
TAG=${TAG:-r04_a}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -q -x -k "empty or bench_ or sharded_equals" > gpurun_out/$TAG/tests_new.log 2>&1; echo "new tests rc=$? $(tail -1 gpurun_out/$TAG/tests_new.log)"
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/$TAG/bench_cfg2.json 2> gpurun_out/$TAG/bench_cfg2.err; echo "cfg2 rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_cfg2.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["selfcheck"], d.get("tile_hint_off"), d.get("n1_same_stream"))
PY
for wl in cfg4 cfg5 cfg2cx; do timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err; echo "$wl rc=$?: $(tail -1 gpurun_out/$TAG/bench_$wl.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], (d["selfcheck"] or {}).get("ok"), d.get("tile_hint_off"))' 2>&1 | tail -1)"; done
echo r4_a done
