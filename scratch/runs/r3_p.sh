#!/bin/bash
TAG=${TAG:-r03_p}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_sharded.py tests/test_gpu_bed.py -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
one() { name=$1; shift
  ( for kv in "$@"; do export $kv; done
    timeout -k 10 200 python bench.py --workload ${WL:-cfg2} --steps 20 --warmup 3 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)" )
}
L=$R/epialleler_amd/csrc
one base
one prev EPIHIP_LIB=$L/libepihip_tprev.so
one base_b
WL=cfg2u one cfg2u
WL=cfg2u one cfg2u_prev EPIHIP_LIB=$L/libepihip_tprev.so
WL=cfg5 one cfg5
WL=cfg5 one cfg5_prev EPIHIP_LIB=$L/libepihip_tprev.so
BENCH_ARGS="--workload cfg2" bash scratch/pmc2.sh ${TAG}_base "p2" > gpurun_out/$TAG/pmc_base.log 2>&1; grep -i "cx_tiles" gpurun_out/pmc_${TAG}_base/summary.txt | grep "INSTS_VALU\|INSTS_LDS \|INSTS_SALU" | cut -c50-120
rm -rf gpurun_out/pmc_${TAG}_*/p?
echo $TAG done
