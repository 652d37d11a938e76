#!/bin/bash
TAG=${TAG:-r03_aa}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
L=$R/epialleler_amd/csrc
EPIHIP_LIB=$L/libepihip_tcxp2k.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fixture or ragged or toys or pileup or synth_medium" > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
one() { name=$1; shift
  ( for kv in "$@"; do export $kv; done
    timeout -k 10 200 python bench.py --workload ${WL:-cfg2cx} --steps 10 --warmup 2 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)" )
}
one base
one t2k EPIHIP_LIB=$L/libepihip_tcxp2k.so
one base_b
one t2k_b EPIHIP_LIB=$L/libepihip_tcxp2k.so
echo $TAG done
