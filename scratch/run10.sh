#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $R/gpurun_out/p3.log 2>&1
tail -1 $R/gpurun_out/p3.log | cut -c1-300
bash $R/scratch/pmc.sh b
