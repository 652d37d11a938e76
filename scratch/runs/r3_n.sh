#!/bin/bash
TAG=${TAG:-r03_n}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 300 python scratch/debug_mhl.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -12
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_sharded.py -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
one() { name=$1; shift
  ( for kv in "$@"; do export $kv; done
    timeout -k 10 200 python bench.py --workload ${WL:-cfg4} --steps 5 --warmup 1 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)" )
}
L=$R/epialleler_amd/csrc
one base
one prev EPIHIP_LIB=$L/libepihip_tprev.so
one base_b
WL=cfg4d one cfg4d
BENCH_ARGS="--workload cfg4" bash scratch/pmc2.sh ${TAG}_base "p2" > gpurun_out/$TAG/pmc_base.log 2>&1; grep -i "mhl_fused" gpurun_out/pmc_${TAG}_base/summary.txt | grep "INSTS_VALU\|INSTS_LDS \|INSTS_SALU" | cut -c50-120
rm -rf gpurun_out/pmc_${TAG}_*/p?
echo $TAG done
