#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 280 python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-sample 200000 > gpurun_out/b_e_cfg4.log 2>&1; tail -1 gpurun_out/b_e_cfg4.log | cut -c1-200
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof9 -- python3 $R/bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-sample 0 > $R/gpurun_out/p9.log 2>&1; echo prof rc=$?
