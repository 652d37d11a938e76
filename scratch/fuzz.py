"""One-off randomized comparison of the GPU path with the oracle over many shapes (not part of the test suite)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H          # noqa: E402
import synth_np              # noqa: E402
from oracle import oracle as orc   # noqa: E402
import epialleler_amd as ea  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t_end = time.time() + budget
it = 0
alphabets = [None, "......hhxzzZZZHXuU-", "zZ", "zZ.", "zzzzZ....", "ZZZZZZZZz.", "hHxXzZuU.+-", "....-----zZ"]
while time.time() < t_end:
    seed = seed0 + it
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 6))
    if os.environ.get("FUZZ_BIG") and it % 4 == 0:
        kind = 6
    if kind == 0:      # short ragged
        t = synth_np.random_templates(rng, int(rng.integers(1, 4000)), 0, int(rng.integers(1, 700)), int(rng.integers(1, 6)),
                                      int(rng.integers(10, 20000)), p_garbage=float(rng.choice([0, 0, 0.05, 0.3])),
                                      alphabet=alphabets[int(rng.integers(0, len(alphabets)))])
    elif kind == 1:    # pile-up
        t = synth_np.random_templates(rng, int(rng.integers(100, 8000)), 20, int(rng.integers(40, 500)), int(rng.integers(1, 3)),
                                      int(rng.integers(2, 300)), alphabet=alphabets[int(rng.integers(0, len(alphabets)))])
    elif kind == 2:    # long reads (multi-block lMHL pass 1)
        t = synth_np.random_templates(rng, int(rng.integers(1, 120)), int(rng.integers(0, 3000)), int(rng.integers(3000, 20000)),
                                      int(rng.integers(1, 4)), int(rng.integers(100, 60000)),
                                      alphabet=alphabets[int(rng.integers(0, len(alphabets)))])
    elif kind == 3:    # medium reads around the single-block limit of k_mhl_rows (2 KiB) and lane-width switches
        top = int(rng.choice([200, 240, 250, 370, 380, 500, 760, 1000, 1010, 2030, 2040, 2050, 3000]))
        t = synth_np.random_templates(rng, int(rng.integers(1, 1500)), max(top - 40, 0), top, int(rng.integers(1, 4)),
                                      int(rng.integers(100, 30000)), alphabet=alphabets[int(rng.integers(0, len(alphabets)))])
    elif kind == 4:    # positions near tile multiples and large coordinates
        t = synth_np.random_templates(rng, int(rng.integers(1, 2000)), 1, int(rng.integers(2, 400)), int(rng.integers(1, 3)), 3000)
        base = int(rng.choice([1, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 10 ** 6, 2 ** 31 - 4000]))
        t["start"] = (t["start"].astype(np.int64) + base - 1).astype(np.int32)
    elif kind == 6:    # uploads of 4 MiB and more (threaded staging copy, 64 MiB chunks), odd sizes
        L = int(rng.choice([101, 150, 299, 301, 1001]))
        t = synth_np.generate(seed=seed, n_total=int(rng.integers(4 << 20, 90 << 20)) // L, read_len=L)
        cut = int(rng.integers(0, 7))                        # drop a few bytes of the last read: any len % 4, len % 4096
        if cut and t["off"][-1] - t["off"][-2] > cut:
            t["off"] = t["off"].copy(); t["off"][-1] -= cut
            t["xm"] = t["xm"][:int(t["off"][-1])]
    elif kind == 5 and it % 2:   # ragged, gapped templates at uniform-random starts (bench cfg2u's model), small
        t = synth_np.generate_uniform(seed=seed, n_total=int(rng.integers(1000, 30000)))
    else:              # the bench generator's model, small
        t = synth_np.generate(seed=seed, n_total=int(rng.integers(1000, 30000)), read_len=int(rng.choice([100, 300, 301, 2000])))
    if kind in (0, 1, 3, 5) and rng.random() < 0.4 and t["off"].size > 2:      # a tail of long templates (lane shapes follow the bulk of the rows)
        every = int(rng.choice([3, 17, 97, 501, 4001]))
        t = synth_np.with_long_tail(t, every, int(rng.choice([330, 400, 650, 1000, 1100, 2500, 4000, 9000])), first=int(rng.integers(0, every)))
    n = t["off"].size - 1
    bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
    try:
        ctxn = str(rng.choice(["CG", "CHG", "CHH", "CxG", "CX"]))
        c = H.CONTEXT_TO_BASES[ctxn]
        mn, mb, mo = int(rng.integers(0, 4)), float(rng.choice([0.0, 0.3, 0.5, 1.0])), float(rng.choice([0.0, 0.1, 1.0]))
        got = ea.rcpp_threshold_reads(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], mn, mb, mo)
        want = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], mn, mb, mo)
        assert np.array_equal(got.astype(np.int32), want), ("threshold", seed)
        gb = ea.rcpp_get_xm_beta(bam, c["ctx_meth"], c["ctx_unmeth"])
        assert np.array_equal(gb.view(np.uint64), orc.get_xm_beta(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"]).view(np.uint64)), ("beta", seed)
        p = want if rng.random() < 0.6 else None
        rctx = str(rng.choice(["Z", "X", "H", "ZX", "ZXH"]))
        H.assert_reports_equal(dict(ea.rcpp_cx_report(bam, p, rctx)),
                               orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, rctx))
        if n:                                                # the one-pass entry point (thresholding inside the tile kernel)
            rctx2 = str(rng.choice(["Z", "X", "H", "ZX", "ZXH"]))
            H.dirty_allocator(bam)
            rep2, p2 = ea.cytosine_report_fused(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], mn, mb, mo,
                                                rctx2, return_pass=True)
            assert np.array_equal(p2.astype(np.int32), want), ("fused pass", seed)
            H.assert_reports_equal(dict(rep2), orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], want, rctx2))
        if not (kind == 6 and n > 200000):                   # (the lMHL oracle is slow on the big cases)
            hctx = str(rng.choice(["Zz", "Xx", "Hh", "ZzXx", "ZzXxHh"]))
            hmax, hmin, moo = int(rng.choice([0, 0, 1, 3, 50])), int(rng.choice([0, 0, 2, 5])), float(rng.choice([0.1, 0.0, 1.0]))
            H.assert_reports_equal(dict(ea.rcpp_mhl_report(bam, hctx, hmax, hmin, moo)),
                                   orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], hctx, hmax, hmin, moo),
                                   float_cols=("length", "lmhl"))
    except Exception:
        print("FAILED at seed", seed, "kind", kind, "n", n, flush=True)
        raise
    finally:
        bam.close()
    it += 1
    if it % 50 == 0:
        print("ok", it, "cases", flush=True)
print("fuzz done:", it, "cases, all equal to the oracle")
