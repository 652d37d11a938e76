"""Child process of test_gpu_variants.py: CX and lMHL reports of a few template sets against the oracle with whatever
EPIHIP_CX_* switches the parent put in the environment (they are read once per process)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H          # noqa: E402
import synth_np              # noqa: E402
from oracle import oracle as orc   # noqa: E402
import epialleler_amd as ea  # noqa: E402


def main():
    rng = np.random.default_rng(77)
    sets = [synth_np.random_templates(rng, 4000, 0, 700, 3, 9000),
            synth_np.random_templates(rng, 6000, 100, 400, 1, 40),                       # pile-up (heavy when EPIHIP_HEAVY_ROWS is low)
            synth_np.random_templates(rng, 60, 3000, 9000, 2, 30000, alphabet="......hhxzzZZZHXuU-"),
            H.bam("capture.bam")]
    allb = np.arange(256, dtype=np.uint8)                                                # every byte value, piled 40 deep
    t = H.templates_from_xm(["z"] * 40, [1] * 40, [1] * 40)
    t["xm"] = np.tile(allb, 40)
    t["off"] = (np.arange(41, dtype=np.int64) * 256)
    t["strand"] = (1 + (np.arange(40) & 1)).astype(np.int32)
    sets.append(t)
    # short reads (up to 305 bytes: the four-lane shapes of the lean CX kernel and the two-block shape of the one-pass lMHL
    # kernel): WGS-like over three reference sequences, sparse with gaps between tiles, one 6000-row pile-up, and a pile-up
    # inside WGS-like rows
    sets.append(synth_np.generate(n_total=20000, read_len=300, n_chr=3))
    sets.append(synth_np.random_templates(rng, 5000, 0, 305, 2, 120000))
    sets.append(synth_np.random_templates(rng, 6000, 100, 300, 1, 40))
    sets.append(synth_np.generate_uniform(n_total=12000, mean_len=300, n_chr=2, ragged=False, gap_every=0, pileup=(3000, 900)))
    for t in sets:
        bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"], t.get("levels"))
        try:
            c = H.CONTEXT_TO_BASES["CG"]
            p = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
            for ctx, pv in (("Z", None), ("ZXH", p), ("ZX", p)):
                got = ea.rcpp_cx_report(bam, pv, ctx)
                want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], pv, ctx)
                H.assert_reports_equal(dict(got), want)
            for ctx in ("Z", "ZXH"):                                      # thresholding fused into the tile kernel
                got, gp = ea.cytosine_report_fused(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1,
                                                   ctx, return_pass=True)
                assert np.array_equal(gp.astype(np.int32), p)
                H.assert_reports_equal(dict(got), orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, ctx))
            for hmax, hmin, moo in ((0, 0, 0.1), (3, 2, 1.0)):
                got = ea.rcpp_mhl_report(bam, "Zz", hmax, hmin, moo)
                want = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", hmax, hmin, moo)
                H.assert_reports_equal(dict(got), want, float_cols=("length", "lmhl"))
        finally:
            bam.close()
    print("variant ok", ea._lib.load().epi_tile_positions())


if __name__ == "__main__":
    main()
