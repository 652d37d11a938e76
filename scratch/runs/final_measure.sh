#!/bin/bash
# round-1 final measurements: GPU suite, bench lines of every workload, rocprofv3 kernel stats (cfg2, cfg4), PMC passes (cfg2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_h.log 2>&1; tail -2 gpurun_out/t_h.log
grep -q "failed\|VIOLATION\|Aborted" gpurun_out/t_h.log && exit 1
timeout -k 10 300 python bench.py > gpurun_out/b_h_cfg2.log 2>&1 || exit 1; tail -1 gpurun_out/b_h_cfg2.log | cut -c1-200
for wl in cfg2cx cfg3 cfg4 cfg5; do timeout -k 10 280 python bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 200000 > gpurun_out/b_h_$wl.log 2>&1 || exit 1; tail -1 gpurun_out/b_h_$wl.log | cut -c1-160; done
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_h4 -- python3 $R/bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-sample 0 > $R/gpurun_out/p_h4.log 2>&1 || exit 1
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_h2 -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $R/gpurun_out/p_h2.log 2>&1 || exit 1
cd $R && bash scratch/pmc.sh h > gpurun_out/pmc_h.log 2>&1; tail -12 gpurun_out/pmc_h.log | cut -c1-250
echo done
