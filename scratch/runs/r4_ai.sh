#!/bin/bash
TAG=${TAG:-r04_ai}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 500 python scratch/upload_cost.py > gpurun_out/$TAG/upload_cost.txt 2>&1; echo "upload_cost rc=$?"; tail -n 6 gpurun_out/$TAG/upload_cost.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/$TAG/bench_cfg2.json 2> gpurun_out/$TAG/bench_cfg2.err; echo "cfg2 rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_cfg2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"], d["selfcheck"]["ok"], d.get("layout_off"), d["config"]["batch_ms"], d["streamed"])
PY
echo done
