#!/bin/bash
cd $GRAFT_REPO_ROOT
cp epialleler_amd/csrc/libepihip.so /tmp/lib_orig.so
run() { timeout -k 10 200 python bench.py --workload cfg2 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b40.log 2>&1; tail -1 gpurun_out/b40.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
for un in 3 5 7 14; do cp scratch/libs/libepihip_un$un.so epialleler_amd/csrc/libepihip.so; run un$un; done
cp /tmp/lib_orig.so epialleler_amd/csrc/libepihip.so
