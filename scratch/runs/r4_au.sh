#!/bin/bash
# round 4, pass m: full GPU suite + default bench line on the state after the communicator / table recycling work
TAG=${TAG:-r04_au}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -n 1 gpurun_out/$TAG/tests.log)"
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/$TAG/bench_cfg2.json 2> gpurun_out/$TAG/bench_cfg2.err; echo "cfg2 rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_cfg2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], d["selfcheck"]["ok"], d["sharded_1rank"]["overhead_ms"], d["host_out"], d["cpu_baseline"]["value"], d.get("layout_off"), d["config"]["batch_ms"], d["tile_hint_off"], d["streamed"], d.get("cfg2t"), d.get("cfg2u"))
PY
echo done
