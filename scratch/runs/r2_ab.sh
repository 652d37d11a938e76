#!/bin/bash
# A/B of variant libraries (scratch/build_variant.sh): parity subset, then cfg2 (and WL) timings, two rounds to see the noise
cd $GRAFT_REPO_ROOT
D=epialleler_amd/csrc
for v in ${PARITY}; do
  export EPIHIP_LIB=$GRAFT_REPO_ROOT/$D/libepihip_t$v.so
  if true; then timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/ab_par_$v.log 2>&1; echo "$v parity: $(tail -1 gpurun_out/ab_par_$v.log)"; fi
done
for round in 1 2; do
for v in ${VARS}; do
  export EPIHIP_LIB=$GRAFT_REPO_ROOT/$D/libepihip_t$v.so
  for w in ${WL:-cfg2}; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --workload $w --no-extras --cpu-sample 0 > gpurun_out/ab_${v}_$w.log 2>&1
    echo "$v $w: $(tail -1 gpurun_out/ab_${v}_$w.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)"
  done
done
done
