"""Prints the first rows in which the GPU lMHL report differs from the oracle, column by column (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H, synth_np
from oracle import oracle as orc
import epialleler_amd as ea

def run(name, t, hmax=0, hmin=0, moo=0.1):
    bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
    got = ea.rcpp_mhl_report(bam, "Zz", hmax, hmin, moo)
    want = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", hmax, hmin, moo)
    bam.close()
    if got["pos"].shape != want["pos"].shape:
        print(name, "ROW COUNT", got["pos"].shape, want["pos"].shape); return
    bad = np.zeros(want["pos"].size, bool)
    for k in want:
        a, b = got[k], want[k]
        if a.dtype.kind == "f":
            bad |= a.view(np.uint64) != b.view(np.uint64)
        else:
            bad |= a != b
    idx = np.nonzero(bad)[0]
    print(name, "rows", want["pos"].size, "mismatching", idx.size)
    for i in idx[:12]:
        print("   row", i, {k: (got[k][i], want[k][i]) for k in want})

rng = np.random.default_rng(3)
run("synth2000", synth_np.generate(n_total=2000, read_len=300))
run("synth_gap", synth_np.generate(n_total=2000, read_len=300, gap_from=150, gap_len=50))
run("amplicon010", H.bam("amplicon010meth.bam"))
run("random", synth_np.random_templates(rng, 3000, 0, 700, 3, 9000))
