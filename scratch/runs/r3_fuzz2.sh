#!/bin/bash
# second fuzz pass of the round: forced lane shapes of the one-pass lMHL kernel, the general CX kernel, more seeds
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_fuzz2
B=${B:-180}
run() { name=$1; shift; ( for kv in "$@"; do export $kv; done; timeout -k 10 $((B+90)) python scratch/fuzz.py $B ${SEED} > gpurun_out/r03_fuzz2/$name.log 2>&1; echo "$name: rc=$? $(tail -1 gpurun_out/r03_fuzz2/$name.log | cut -c1-200)" ); }
SEED=410000 run defaults2
SEED=420000 run shape_16_4 EPIHIP_MHLF_SHAPE=16,4
SEED=430000 run shape_32_3_2 EPIHIP_MHLF_SHAPE=32,3,2 EPIHIP_MHLF_FOLD=0 EPIHIP_MHLF_FOLD_SLOTS=3
SEED=440000 run shape_64_2 EPIHIP_MHLF_SHAPE=64,2 EPIHIP_CX_LEAN=0
SEED=450000 run heavy EPIHIP_HEAVY_ROWS=120 EPIHIP_MHL_SLOT=0 EPIHIP_CX_SLOT=0
echo fuzz2 done
