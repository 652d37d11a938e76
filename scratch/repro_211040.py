"""Reproduces the fuzz failure at seed 211040 (one CX row missing) and prints where the tables differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, synth_np
from oracle import oracle as orc
import epialleler_amd as ea
seed = 211040
rng = np.random.default_rng(seed)
kind = int(rng.integers(0, 6))
t = synth_np.generate(seed=seed, n_total=int(rng.integers(1000, 30000)), read_len=int(rng.choice([100, 300, 301, 2000])))
n = t["off"].size - 1
print("kind", kind, "n", n, "len", int(t["off"][1] - t["off"][0]))
bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
ctxn = str(rng.choice(["CG", "CHG", "CHH", "CxG", "CX"])); c = H.CONTEXT_TO_BASES[ctxn]
mn, mb, mo = int(rng.integers(0, 4)), float(rng.choice([0.0, 0.3, 0.5, 1.0])), float(rng.choice([0.0, 0.1, 1.0]))
want_p = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], mn, mb, mo)
p = want_p if rng.random() < 0.6 else None
rctx = str(rng.choice(["Z", "X", "H", "ZX", "ZXH"]))
print("ctx", ctxn, "pass", p is not None, "rctx", rctx)
for rep_i in range(3):
    got = dict(ea.rcpp_cx_report(bam, p, rctx))
    want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, rctx)
    kg = set(zip(got["rname"].tolist(), got["pos"].tolist(), got["strand"].tolist()))
    kw = set(zip(want["rname"].tolist(), want["pos"].tolist(), want["strand"].tolist()))
    print("call", rep_i, "rows", len(got["pos"]), len(want["pos"]), "missing", sorted(kw - kg)[:5], "extra", sorted(kg - kw)[:5])
    for (r, ps, sd) in sorted(kw - kg)[:3]:
        i = np.where((want["rname"] == r) & (want["pos"] == ps) & (want["strand"] == sd))[0][0]
        print("  wanted row:", {k: int(want[k][i]) for k in want}, "tile(1024) index", (ps) // 1024, "pos%1024", ps % 1024)

def show(tab, lo):
    m = (tab["rname"] == 4) & (tab["pos"] >= lo)
    return [tuple(int(tab[k][i]) for k in ("pos", "strand", "context", "meth", "unmeth")) for i in np.where(m)[0]]
print("got ", show(got, 64440))
print("want", show(want, 64440))
# the same question on the tail of the batch alone
for first in (25582, 25570, 25400, 20000):
    sl = slice(first, n)
    o0 = int(t["off"][first])
    tb = ea.ProcessedBam.from_arrays(t["xm"][o0:], t["off"][first:] - o0, t["rname"][sl], t["strand"][sl], t["start"][sl])
    pp = p[sl] if p is not None else None
    g2 = dict(ea.rcpp_cx_report(tb, pp, rctx))
    w2 = orc.cx_report(t["xm"][o0:], t["off"][first:] - o0, t["rname"][sl], t["strand"][sl], t["start"][sl], pp, rctx)
    print("tail from", first, "rows", len(g2["pos"]), len(w2["pos"]), "off0&3", o0 & 3)
    tb.close()
# no thresholding
g3 = dict(ea.rcpp_cx_report(bam, None, rctx)); w3 = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, rctx)
print("no pass: rows", len(g3["pos"]), len(w3["pos"]))
print("pass of last rows", None if p is None else p[-5:])

# raw counters of the last tile through the shared-slab path
from epialleler_amd import distributed as D
eng = D.HipShardEngine(bam)
first, last = eng.key_range()
keys = np.array([last], dtype=np.int64); owned = np.ones(1, dtype=np.int32)
import torch
pt = None if p is None else torch.as_tensor(p.astype(np.int32)).cuda()
slab = eng.cx_accumulate(pt, "H", keys, owned).cpu().numpy().reshape(16, -1)
print("key", hex(last), "T", slab.shape[1])
for pos in range(968, 976):
    print(pos, "strand+ planes", slab[0:8, pos].tolist(), "strand- planes", slab[8:16, pos].tolist())
cols = eng.cx_finish("H")
print("rows after finish", cols.shape)
