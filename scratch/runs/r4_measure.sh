#!/bin/bash
# round-4 measurements for profiles/: GPU suite, bench lines of every workload, rocprofv3 kernel stats (cfg2, cfg2cx, cfg4, cfg5),
# PMC passes (all four of cfg2 and cfg4; the two traffic passes of cfg2n, cfg2cx, cfg4d, cfg5)
TAG=${TAG:-r04_z}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/$TAG/gpu_tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 gpurun_out/$TAG/gpu_tests.txt)"
timeout -k 10 500 python bench.py > gpurun_out/$TAG/bench_cfg2.json 2> gpurun_out/$TAG/bench_cfg2.err || exit 1; tail -n 1 gpurun_out/$TAG/bench_cfg2.json | cut -c1-200
for wl in cfg2cx cfg2n cfg4 cfg4d cfg5; do timeout -k 10 280 python bench.py --workload $wl --steps 5 --warmup 1 --no-extras --cpu-sample 200000 > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err || exit 1; tail -n 1 gpurun_out/$TAG/bench_$wl.json | cut -c1-160; done
timeout -k 10 300 python bench.py --workload file --steps 3 --warmup 1 > gpurun_out/$TAG/bench_file.json 2> gpurun_out/$TAG/bench_file.err; tail -n 1 gpurun_out/$TAG/bench_file.json | cut -c1-400
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --backend gloo --steps 3 --no-extras --rows 2000000 > gpurun_out/$TAG/bench_gloo2_rehearsal.json 2> gpurun_out/$TAG/bench_gloo2.err; tail -n 1 gpurun_out/$TAG/bench_gloo2_rehearsal.json | cut -c1-160
cd /tmp && export TMPDIR=/tmp
for wl in ${PROF_WL:-cfg2 cfg2cx cfg4 cfg5}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof_$wl -- python3 $R/bench.py --workload $wl --steps 10 --warmup 2 --no-extras --cpu-sample 0 > $R/gpurun_out/$TAG/prof_$wl.log 2>&1 || exit 1
  cp $(ls $R/gpurun_out/$TAG/prof_$wl/*/*kernel_stats.csv | head -n 1) $R/gpurun_out/$TAG/kernel_stats_$wl.csv
  head -n 4 $R/gpurun_out/$TAG/kernel_stats_$wl.csv | cut -c1-150
done
cd $R
bash scratch/pmc2.sh ${TAG}_cfg2 > gpurun_out/$TAG/pmc_cfg2.log 2>&1; cp gpurun_out/pmc_${TAG}_cfg2/summary.txt gpurun_out/$TAG/pmc_cfg2.txt
BENCH_ARGS="--workload cfg4" bash scratch/pmc2.sh ${TAG}_cfg4 > gpurun_out/$TAG/pmc_cfg4.log 2>&1; cp gpurun_out/pmc_${TAG}_cfg4/summary.txt gpurun_out/$TAG/pmc_cfg4.txt
for wl in cfg2n cfg2cx cfg4d cfg5; do
  BENCH_ARGS="--workload $wl" bash scratch/pmc2.sh ${TAG}_$wl "p3 p4" > gpurun_out/$TAG/pmc_$wl.log 2>&1; cp gpurun_out/pmc_${TAG}_$wl/summary.txt gpurun_out/$TAG/pmc_$wl.txt
done
rm -rf gpurun_out/$TAG/prof_* gpurun_out/pmc_${TAG}_*
echo measure done
