#!/bin/bash
TAG=${TAG:-r03_x}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
L=$R/epialleler_amd/csrc
one() { name=$1; shift
  ( for kv in "$@"; do export $kv; done
    timeout -k 10 200 python bench.py --workload ${WL:-cfg2cx} --steps 10 --warmup 2 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)" )
}
one base
one noemit EPIHIP_LIB=$L/libepihip_tcx16.so
export EPIHIP_LIB=$L/libepihip_tcx16.so
BENCH_ARGS="--workload cfg2cx" bash scratch/pmc2.sh ${TAG}_noemit "p2" > gpurun_out/$TAG/pmc.log 2>&1; grep "cxp_tiles" gpurun_out/pmc_${TAG}_noemit/summary.txt | grep "INSTS_VALU\|INSTS_LDS " | cut -c50-120
rm -rf gpurun_out/pmc_${TAG}_*/p?
echo $TAG done
