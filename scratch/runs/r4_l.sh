#!/bin/bash
TAG=${TAG:-r04_l}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x > gpurun_out/$TAG/tests_sharded.log 2>&1; echo "sharded tests rc=$? $(tail -1 gpurun_out/$TAG/tests_sharded.log)"
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/$TAG/bench_cfg2.json 2> gpurun_out/$TAG/bench_cfg2.err; echo "cfg2 rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_cfg2.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["sharded_1rank"], d["host_out"])
PY

echo done
