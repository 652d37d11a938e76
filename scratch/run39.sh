#!/bin/bash
cd $GRAFT_REPO_ROOT
cp epialleler_amd/csrc/libepihip.so /tmp/lib_orig.so
run() { timeout -k 10 200 python bench.py --workload cfg2 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b39.log 2>&1; tail -1 gpurun_out/b39.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
run normal
cp scratch/libs/libepihip_noalu.so epialleler_amd/csrc/libepihip.so
run noalu
cp /tmp/lib_orig.so epialleler_amd/csrc/libepihip.so
