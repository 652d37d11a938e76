#!/bin/bash
TAG=${TAG:-r03_y}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_bed.py -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
L=$R/epialleler_amd/csrc
one() { name=$1; shift
  ( for kv in "$@"; do export $kv; done
    timeout -k 10 200 python bench.py --workload ${WL:-cfg2cx} --steps 10 --warmup 2 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)" )
}
one base
one prev EPIHIP_LIB=$L/libepihip_tprev.so
one base_b
WL=cfg2 one cfg2
WL=cfg2 one cfg2_prev EPIHIP_LIB=$L/libepihip_tprev.so
echo $TAG done
