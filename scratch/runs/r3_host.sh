#!/bin/bash
# host-side timings on the GPU box's CPU: BAM reader phases (16 threads), with and without libdeflate
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r03_host
ls -la /lib/x86_64-linux-gnu/libdeflate* 2>&1 | head -3; nproc; grep -m1 "model name" /proc/cpuinfo
EPIHIP_BAM_TIMING=1 python scratch/bam_phases.py 500000 2>&1 | tail -14
echo "--- zlib only"
EPIHIP_NO_LIBDEFLATE=1 EPIHIP_BAM_TIMING=1 python scratch/bam_phases.py 500000 2>&1 | tail -5
timeout -k 10 300 python bench.py --workload file --steps 3 --warmup 1 2> gpurun_out/r03_host/bench_file.err | tee gpurun_out/r03_host/bench_file.json | cut -c1-600
