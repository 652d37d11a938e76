#!/bin/bash
# debugging aid + A/B timings of the one-pass lMHL kernel (variant libraries from scratch/build_variant.sh, FILE=mhl_fused)
TAG=${TAG:-r03_c}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 300 python scratch/debug_mhl.py > gpurun_out/$TAG/debug.log 2>&1; echo "debug rc=$?"; head -60 gpurun_out/$TAG/debug.log
D=epialleler_amd/csrc
one() { # name, lib (or ""), extra env
  name=$1; lib=$2; shift 2
  ( [ -n "$lib" ] && export EPIHIP_LIB=$R/$D/$lib; for kv in "$@"; do export $kv; done
    timeout -k 10 200 python bench.py --workload cfg4 --steps 5 --warmup 1 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)" )
}
one base ""
one a1_noemit libepihip_ta1.so
one a2_loads libepihip_ta2.so
one a4_noruns libepihip_ta4.so
one w6 libepihip_tw6.so
one t2k libepihip_tt2k.so
one shape16x2 "" EPIHIP_MHLF_SHAPE=16,2
one shape8x4 "" EPIHIP_MHLF_SHAPE=8,4
one base2 ""
echo r3_c done
