#!/bin/bash
# round 4 (CX kernel refactored onto the windowed LDS layout; library communicator): randomized GPU-vs-oracle runs over the reworked one-pass lMHL kernel (both fold variants, slab slots exhausted,
# forced lane shapes) and the CX kernels with the two-perm LUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_fuzz5
B=${B:-120}
run() { name=$1; shift; ( for kv in "$@"; do export $kv; done; timeout -k 10 $((B+90)) python scratch/fuzz.py $B ${SEED} > gpurun_out/r04_fuzz5/$name.log 2>&1; echo "$name: rc=$? $(tail -n 1 gpurun_out/r04_fuzz5/$name.log | cut -c1-200)" ); }
SEED=414000 run defaults
SEED=424000 run nofold_slots1 EPIHIP_MHLF_FOLD=0 EPIHIP_MHLF_FOLD_SLOTS=1 EPIHIP_HEAVY_ROWS=300
SEED=434000 run fold EPIHIP_MHLF_FOLD=1 EPIHIP_MHL_SLOT=3 EPIHIP_CX_SLOT=5
SEED=444000 run cx_general EPIHIP_CX_LEAN=0 EPIHIP_HEAVY_ROWS=200 EPIHIP_TILE_HINT=0
SEED=454000 FUZZ_BIG=1 run big FUZZ_BIG=1
SEED=464000 run realign0 EPIHIP_REALIGN=0
SEED=474000 run realign4 EPIHIP_REALIGN=4 EPIHIP_CX_LEAN=0
echo fuzz done
