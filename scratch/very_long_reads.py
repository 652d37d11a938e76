"""Reads of up to 200 kb (longer than the 64 KiB limit of the wide per-read kernel and of one k_mhl_rows block) against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, synth_np
from oracle import oracle as orc
import epialleler_amd as ea
rng = np.random.default_rng(5)
for n, lo, hi, span in ((40, 1000, 200000, 500000), (300, 0, 90000, 2000000), (5, 65535, 65537, 1000)):
    t = synth_np.random_templates(rng, n, lo, hi, 2, span, p_garbage=0.01)
    bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
    c = H.CONTEXT_TO_BASES["CG"]
    got = ea.rcpp_threshold_reads(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
    want = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
    assert np.array_equal(got.astype(np.int32), want)
    gb = ea.rcpp_get_xm_beta(bam, "Z", "z")
    assert np.array_equal(gb.view(np.uint64), orc.get_xm_beta(t["xm"], t["off"], "Z", "z").view(np.uint64))
    for p in (None, want):
        H.assert_reports_equal(dict(ea.rcpp_cx_report(bam, p, "ZXH")), orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, "ZXH"))
    H.assert_reports_equal(dict(ea.rcpp_mhl_report(bam, "Zz", 0, 0, 0.1)),
                           orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1), float_cols=("length", "lmhl"))
    print("ok", n, "reads up to", int(np.diff(t["off"]).max()), "bytes")
    bam.close()
