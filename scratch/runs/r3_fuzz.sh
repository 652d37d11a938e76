#!/bin/bash
# round 3: randomized GPU-vs-oracle runs over the reworked one-pass lMHL kernel (both fold variants, slab slots exhausted,
# forced lane shapes) and the CX kernels with the two-perm LUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_fuzz
B=${B:-120}
run() { name=$1; shift; ( for kv in "$@"; do export $kv; done; timeout -k 10 $((B+90)) python scratch/fuzz.py $B ${SEED} > gpurun_out/r03_fuzz/$name.log 2>&1; echo "$name: rc=$? $(tail -1 gpurun_out/r03_fuzz/$name.log | cut -c1-200)" ); }
SEED=310000 run defaults
SEED=320000 run nofold_slots1 EPIHIP_MHLF_FOLD=0 EPIHIP_MHLF_FOLD_SLOTS=1 EPIHIP_HEAVY_ROWS=300
SEED=330000 run fold EPIHIP_MHLF_FOLD=1 EPIHIP_MHL_SLOT=3 EPIHIP_CX_SLOT=5
SEED=340000 run nofold EPIHIP_MHLF_FOLD=0 EPIHIP_HEAVY_ROWS=5000
SEED=350000 FUZZ_BIG=1 run big FUZZ_BIG=1
echo fuzz done
