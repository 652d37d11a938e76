#!/bin/bash
cd $GRAFT_REPO_ROOT
cp epialleler_amd/csrc/libepihip.so /tmp/lib_orig.so
run() { timeout -k 10 200 python bench.py --workload $1 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/b22.log 2>&1; tail -1 gpurun_out/b22.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['config']['workload'][:8], d['ms_per_step'], d['roofline']['kernel_ms_all'])"; }
for g in 8 16; do EPIHIP_CX_GROUP=$g run cfg2 "nu10 g$g"; done
for g in 16 32 64; do EPIHIP_CX_GROUP=$g run cfg5 "nu10 g$g"; done
for nu in 5 7 12; do
  cp scratch/libs/libepihip_nu$nu.so epialleler_amd/csrc/libepihip.so
  for g in 8 16; do EPIHIP_CX_GROUP=$g run cfg2 "nu$nu g$g"; done
  for g in 32 64; do EPIHIP_CX_GROUP=$g run cfg5 "nu$nu g$g"; done
done
cp /tmp/lib_orig.so epialleler_amd/csrc/libepihip.so
