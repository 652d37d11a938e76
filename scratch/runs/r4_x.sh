#!/bin/bash
TAG=${TAG:-r04_x}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 2 --no-extras --cpu-sample 0 --no-selfcheck > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"])' 2>&1 | tail -1)"; }
run full cfg4 X=1
for A in 1 2 4 8; do run ablate$A cfg4 EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tmab$A.so; done
echo done
