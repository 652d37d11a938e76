#!/bin/bash
TAG=${TAG:-r03_ah}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
L=$R/epialleler_amd/csrc
EPIHIP_LIB=$L/libepihip_tpipe.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests(pipe) rc=$? $(tail -1 gpurun_out/$TAG/tests.log)"
one() { name=$1; shift
  ( for kv in "$@"; do export $kv; done
    timeout -k 10 250 python bench.py --workload ${WL:-cfg2} --steps ${ST:-20} --warmup 2 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"])' 2>&1 | tail -1)" )
}
one base
one pipe EPIHIP_LIB=$L/libepihip_tpipe.so
one base_b
one pipe_b EPIHIP_LIB=$L/libepihip_tpipe.so
WL=cfg2n one cfg2n
WL=cfg2n one cfg2n_pipe EPIHIP_LIB=$L/libepihip_tpipe.so
WL=cfg5 ST=5 one cfg5
WL=cfg5 ST=5 one cfg5_pipe EPIHIP_LIB=$L/libepihip_tpipe.so
echo $TAG done
