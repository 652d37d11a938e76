#!/bin/bash
cd $GRAFT_REPO_ROOT
cp epialleler_amd/csrc/libepihip.so /tmp/lib_orig.so
run() { timeout -k 10 280 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/b_nu.log 2>&1; tail -1 gpurun_out/b_nu.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['roofline']['kernel_ms_all'])"; }
run nu10
for nu in 5 7 8; do cp scratch/libs/libepihip_mnu$nu.so epialleler_amd/csrc/libepihip.so; run nu$nu; done
cp /tmp/lib_orig.so epialleler_amd/csrc/libepihip.so
