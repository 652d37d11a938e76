"""debug: the lMHL report of the three shards of the wgs_tail case, many times, against the oracle (single process)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, synth_np
from oracle import oracle as orc
import epialleler_amd as ea

t = synth_np.with_long_tail(synth_np.generate(n_total=9000, read_len=300, n_chr=3), 61, 1100, first=7)
n = t["off"].size - 1
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    for rank in range(3):
        lo, hi = n * rank // 3, n * (rank + 1) // 3
        off = t["off"][lo:hi + 1]
        sub = dict(xm=t["xm"][int(off[0]):int(off[-1])], off=off - off[0], rname=t["rname"][lo:hi], strand=t["strand"][lo:hi], start=t["start"][lo:hi])
        bam = ea.ProcessedBam.from_arrays(sub["xm"], sub["off"], sub["rname"], sub["strand"], sub["start"])
        try:
            for hmax in (0, 2):
                got = dict(ea.rcpp_mhl_report(bam, "Zz", hmax, 0, 0.1))
                want = orc.mhl_report(sub["xm"], sub["off"], sub["rname"], sub["strand"], sub["start"], "Zz", hmax, 0, 0.1)
                for k in want:
                    g, w = np.asarray(got[k]), np.asarray(want[k])
                    if g.shape != w.shape or not np.array_equal(g, w, equal_nan=(g.dtype.kind == "f")):
                        bad += 1
                        i = int(np.argmax(g[:min(len(g), len(w))] != w[:min(len(g), len(w))])) if g.shape == w.shape else -1
                        print("MISMATCH it=%d rank=%d hmax=%d col=%s shapes %s %s first index %d got %s want %s pos got %s want %s" % (
                            it, rank, hmax, k, g.shape, w.shape, i, g[i:i + 3], w[i:i + 3], np.asarray(got["pos"])[i:i + 3], np.asarray(want["pos"])[i:i + 3]), flush=True)
                        break
        finally:
            bam.close()
print("done, mismatches:", bad)
