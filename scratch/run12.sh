#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout=450 > gpurun_out/t3.log 2>&1; echo exit=$? >> gpurun_out/t3.log; tail -25 gpurun_out/t3.log
timeout -k 10 120 python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_all'])"
python - <<'PY'
# amplicon-like stress: 4M reads of 300 B piled on 3 amplicons; with and without splitting
import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import epialleler_amd as ea
rng = np.random.default_rng(1)
n = 4_000_000
L = 300
start = np.sort(rng.choice([1000, 1200, 500000], n) + rng.integers(0, 40, n)).astype(np.int32)
xm = (0x10 | rng.choice(np.array([12, 12, 12, 12, 10, 14, 15, 7], np.uint8), n * L)).astype(np.uint8)
off = np.arange(n + 1, dtype=np.int64) * L
bam = ea.ProcessedBam.from_arrays(xm, off, np.ones(n, np.int32), rng.integers(1, 3, n).astype(np.int32), start)
for hv in ("1000000000", "16384"):
    os.environ["EPIHIP_HEAVY_ROWS"] = hv
    r = ea.rcpp_cx_report(bam, None, "Z"); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = ea.rcpp_cx_report(bam, None, "Z", as_device=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("heavy_rows", hv, "rows", r.nrow, "ms", round(dt * 1e3, 2), "meth", int(r["meth"].sum()), "unmeth", int(r["unmeth"].sum()))
PY
