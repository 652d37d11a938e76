#!/bin/bash
# lMHL one-pass kernel: mismatch check, the GPU suite, A/B of workgroup / tile shapes and finer ablations, PMC of the base
TAG=${TAG:-r03_d}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 300 python scratch/debug_mhl.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | head -30
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
D=epialleler_amd/csrc
one() { name=$1; lib=$2; shift 2
  ( [ -n "$lib" ] && export EPIHIP_LIB=$R/$D/$lib; for kv in "$@"; do export $kv; done
    timeout -k 10 200 python bench.py --workload cfg4 --steps 5 --warmup 1 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)" )
}
one base ""
one a4_noruns libepihip_ta4.so
one a8_non8 libepihip_ta8.so
one v256 libepihip_tv256.so
one v256w8 libepihip_tv256w8.so
one t2k512 libepihip_tt2k512.so
one base2 ""
BENCH_ARGS="--workload cfg4" bash scratch/pmc2.sh ${TAG}_cfg4 "p1 p2" > gpurun_out/$TAG/pmc_cfg4.log 2>&1; grep -i "mhl_fused" gpurun_out/pmc_${TAG}_cfg4/summary.txt | cut -c30-120
echo r3_d done
