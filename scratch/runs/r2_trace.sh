#!/bin/bash
# kernel trace of a few cfg2 steps: per-kernel start/end to see the gaps between the kernels of one step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/trace; mkdir -p $R/gpurun_out/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace/p -- python3 $R/bench.py --steps 6 --warmup 2 --no-extras --cpu-sample 0 ${BENCH_ARGS} > $R/gpurun_out/trace/log.txt 2>&1
f=$(ls $R/gpurun_out/trace/p/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    prev_end = e
PY
rm -rf $R/gpurun_out/trace/p
