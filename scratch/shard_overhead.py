"""Per-step time of the sharded driver on one rank (no collectives) against the plain single-GPU call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import epialleler_amd as ea
from epialleler_amd import distributed as D, synth
bam = synth.generate_device(n_total=10_000_000, read_len=300, device=0)
eng = D.HipShardEngine(bam)
def t(fn, n=20):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("plain   %.3f ms" % t(lambda: ea.generateCytosineReport(bam, as_device=True)))
print("sharded %.3f ms" % t(lambda: D.sharded_cytosine_report(eng, gather=False, levels=bam.levels)))
# with two artificial shared tiles (slab path + finish) on one rank
import numpy as np
first, last = eng.key_range()
keys = np.array([first, first + 1], dtype=np.int64); owned = np.ones(2, dtype=np.int32)
from epialleler_amd.api import CONTEXT_TO_BASES as C
c = C["CG"]
thr = (c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
def shared_step():
    eng.cx_accumulate_fused(thr, c["ctx_meth"], keys, owned)
    return eng.cx_finish(c["ctx_meth"])
print("sharded with 2 shared tiles, fused (no collective) %.3f ms" % t(shared_step))
keys = np.arange(first, first + 14, dtype=np.int64); owned = np.ones(14, dtype=np.int32)
print("sharded with 14 shared tiles, fused (no collective) %.3f ms" % t(shared_step))
