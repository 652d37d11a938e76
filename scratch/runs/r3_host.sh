#!/bin/bash
# host-side timings on the GPU box's CPU: BAM reader phases (16 threads), A/B of huge pages / libdeflate
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r03_host
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
EPIHIP_BAM_TIMING=1 python scratch/bam_phases.py 500000 2>&1 | tail -8
echo "--- no hugepage"
EPIHIP_NO_HUGEPAGE=1 EPIHIP_BAM_TIMING=1 python scratch/bam_phases.py 500000 2>&1 | tail -8
timeout -k 10 300 python bench.py --workload file --steps 3 --warmup 1 2> gpurun_out/r03_host/bench_file.err | tee gpurun_out/r03_host/bench_file.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["ms"])'
EPIHIP_NO_HUGEPAGE=1 timeout -k 10 300 python bench.py --workload file --steps 3 --warmup 1 2> /dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("nohuge", d["ms_per_step"], d["value"], d["config"]["ms"])'
