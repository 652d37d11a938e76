#!/bin/bash
TAG=${TAG:-r03_u}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
L=$R/epialleler_amd/csrc
EPIHIP_LIB=$L/libepihip_tt2048.so timeout -k 10 300 python scratch/debug_mhl.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -6
one() { name=$1; shift
  ( for kv in "$@"; do export $kv; done
    timeout -k 10 200 python bench.py --workload ${WL:-cfg4} --steps 5 --warmup 1 --no-extras --cpu-sample 0 > gpurun_out/$TAG/ab_$name.json 2> gpurun_out/$TAG/ab_$name.err
    echo "$name: $(tail -1 gpurun_out/$TAG/ab_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)" )
}
one base
one t2048 EPIHIP_LIB=$L/libepihip_tt2048.so
one base_b
echo $TAG done
