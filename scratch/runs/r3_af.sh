#!/bin/bash
TAG=${TAG:-r03_af}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -q -x -k "long or fixture or ragged or variants" > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/$TAG/fullsize.log 2>&1; echo "fullsize rc=$? $(tail -1 gpurun_out/$TAG/fullsize.log)"
L=$R/epialleler_amd/csrc
one() { name=$1; shift
  ( for kv in "$@"; do export $kv; done
    timeout -k 10 250 python bench.py --workload ${WL:-cfg5} --steps 5 --warmup 1 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"])' 2>&1 | tail -1)" )
}
one base
one prev EPIHIP_LIB=$L/libepihip_tprev.so
one base_b
echo $TAG done
