"""Latency of the first reports on a fresh resident batch (allocations, pool sizing), config 2 size."""
import time
import torch
import epialleler_amd as ea
from epialleler_amd import synth
for kw, name in ((dict(), "defaults (threshold + CG)"), (dict(threshold_reads=False, report_context="CX"), "CX, no threshold")):
    bam = synth.generate_device(n_total=10_000_000, read_len=300, row_first=0, n=10_000_000, device=0)
    torch.cuda.synchronize()
    ts = []
    for i in range(3):
        t0 = time.perf_counter()
        r = ea.generateCytosineReport(bam, as_device=True, **kw)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        del r
    print("%s: calls 1-3 = %.1f, %.1f, %.1f ms" % (name, *ts), flush=True)
    bam.close(); del bam
bam = synth.generate_device(n_total=10_000_000, read_len=300, row_first=0, n=10_000_000, device=0)
torch.cuda.synchronize()
ts = []
for i in range(3):
    t0 = time.perf_counter(); r = ea.generateMhlReport(bam, as_device=True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); del r
print("generateMhlReport (10 M): calls 1-3 = %.1f, %.1f, %.1f ms" % tuple(ts))
