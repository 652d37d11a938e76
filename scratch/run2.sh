#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout=450 > gpurun_out/t3.log 2>&1; echo exit=$? >> gpurun_out/t3.log; tail -4 gpurun_out/t3.log
timeout -k 10 120 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'])"
for G in 4 8 16; do echo "thrG=$G"; EPIHIP_GROUP=$G timeout -k 10 120 python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; done
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/p2.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof2/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters.txt 2>&1 || true
