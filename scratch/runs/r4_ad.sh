#!/bin/bash
TAG=${TAG:-r04_ad}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
run() { name=$1; wl=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-extras --cpu-sample 0 --no-selfcheck > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; echo "$name rc=$?: $(tail -n 1 gpurun_out/$TAG/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_all"], d["roofline"]["frac"])' 2>&1 | tail -n 1)"; }
for wl in cfg5 cfg2n cfg2; do
run ${wl}_base $wl X=1
for a in 1 8 9; do run ${wl}_a$a $wl EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_tal$a.so; done
done
echo done
