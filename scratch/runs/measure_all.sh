#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 300 python bench.py > gpurun_out/b_d_cfg2.log 2>&1; tail -1 gpurun_out/b_d_cfg2.log | cut -c1-400
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof6 -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $R/gpurun_out/p6.log 2>&1
bash $R/scratch/pmc.sh d
cd $R
for wl in cfg2cx cfg4 cfg5; do timeout -k 10 280 python bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 200000 > gpurun_out/b_d_$wl.log 2>&1; tail -1 gpurun_out/b_d_$wl.log | cut -c1-120; done
