#!/bin/bash
# round 3, first call: the GPU suite on the hygiene changes, the default bench line on the new headline stream (uniform
# starts) with host_out / sharded_1rank / cfg2g / cfg2u / cfg2p, cfg4 + cfg4d, and the gloo-2 rehearsal with a real exchange
TAG=${TAG:-r03_a}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
timeout -k 10 500 python bench.py > gpurun_out/$TAG/bench_cfg2.json 2> gpurun_out/$TAG/bench_cfg2.err; echo "bench rc=$?"; tail -1 gpurun_out/$TAG/bench_cfg2.json | cut -c1-300
for wl in cfg4 cfg4d cfg2cx; do timeout -k 10 280 python bench.py --workload $wl --steps 5 --warmup 1 --no-extras --cpu-sample 0 > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err; echo "$wl rc=$?"; tail -1 gpurun_out/$TAG/bench_$wl.json | cut -c1-200; done
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --backend gloo --steps 3 --no-extras --rows 2000000 > gpurun_out/$TAG/bench_gloo2_rehearsal.json 2> gpurun_out/$TAG/bench_gloo2.err; echo "gloo2 rc=$?"; tail -1 gpurun_out/$TAG/bench_gloo2_rehearsal.json | cut -c1-300
echo r3_a done
