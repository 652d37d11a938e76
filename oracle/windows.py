"""Full-size parity through windows -- TEST INFRASTRUCTURE ONLY (tests/, bench.py's self-check).

The oracle cannot run 10 M reads in test time.  What it can do: take a window of consecutive rows out of a
device-resident batch, run the reference algorithm on those rows alone, and compare with the rows of the full-size
report that lie where no read outside the window can reach (one read length in from the window's ends).  The rule per
(rname, pos, strand) is local, so inside that range the two tables must be identical, bit for bit
(src/rcpp_cx_report.cpp:108-131, src/rcpp_mhl_report.cpp:138-198).
"""
import numpy as np

from . import oracle as orc

THR_CG = ("Z", "z", "XH", "xh", 2, 0.5, 0.1)        # generateCytosineReport's thresholding defaults (R/generateCytosineReport.R:164-171)


def window_rows(dev, lo, hi):
    """Host copy of rows [lo, hi) of a device-resident batch (dict of torch tensors xm/off/rname/strand/start)."""
    off = dev["off"][lo:hi + 1].cpu().numpy()
    xm = dev["xm"][int(off[0]):int(off[-1])].cpu().numpy()
    return {"xm": xm, "off": off - off[0], "rname": dev["rname"][lo:hi].cpu().numpy(),
            "strand": dev["strand"][lo:hi].cpu().numpy(), "start": dev["start"][lo:hi].cpu().numpy(),
            "byte0": int(off[0])}


def oracle_for(kind, threshold=False, letters="Z"):
    """The oracle call that corresponds to a bench workload / an R-level report call."""
    if kind == "mhl":
        return lambda w: orc.mhl_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], "Zz", 0, 0, 0.1)

    def cx(w):
        p = orc.threshold_reads(w["xm"], w["off"], *THR_CG) if threshold else None
        return orc.cx_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], p, letters)
    return cx


def _np(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)


def check_windows(rep, dev, n, oracle_fn, float_cols=(), L=300, wrows=20000, starts=None, min_rows=300):
    """`rep`: the full-size report (numpy columns or torch tensors on the device); `dev`: the batch's device columns.
    Checks strict (rname, pos, strand) order of the whole table, then every window.  Returns a summary dict; raises
    AssertionError with the first difference."""
    dev_side = hasattr(rep["pos"], "cpu")
    if dev_side:
        import torch
        key = rep["rname"].to(torch.int64) * (1 << 33) + rep["pos"].to(torch.int64) * 2 + (rep["strand"].to(torch.int64) - 1)
        assert bool((key[1:] > key[:-1]).all()), "report rows are not in strict (rname, pos, strand) order"
        kpos = key >> 1                                     # rname * 2^32 + pos
    else:
        key = rep["rname"].astype(np.int64) * (1 << 33) + rep["pos"].astype(np.int64) * 2 + (rep["strand"] - 1)
        assert np.all(np.diff(key) > 0), "report rows are not in strict (rname, pos, strand) order"
        kpos = key >> 1
    del key
    wrows = min(wrows, n)
    starts = list(starts) if starts is not None else [0, n // 3 + 17, n - wrows]
    done = []
    for lo in starts:
        lo = max(0, min(int(lo), n - wrows))
        hi = lo + wrows
        w = window_rows(dev, lo, hi)
        want = oracle_fn(w)
        # rows of the window are complete only where no read outside [lo, hi) can reach: trim one read length
        r0, r1 = int(w["rname"][0]), int(w["rname"][-1])
        p_lo = int(w["start"][0]) + L if lo > 0 else -1
        p_hi = int(w["start"][-1]) - 1 if hi < n else 2 ** 31 - 1
        k_lo, k_hi = (r0 << 32) + p_lo, (r1 << 32) + p_hi
        kw = (want["rname"].astype(np.int64) << 32) + want["pos"]
        mw = (kw >= k_lo) & (kw <= k_hi)
        if dev_side:
            import torch
            a = int(torch.searchsorted(kpos, torch.tensor([k_lo], device=kpos.device, dtype=torch.int64), right=False))
            b = int(torch.searchsorted(kpos, torch.tensor([k_hi], device=kpos.device, dtype=torch.int64), right=True))
        else:
            a = int(np.searchsorted(kpos, k_lo, side="left"))
            b = int(np.searchsorted(kpos, k_hi, side="right"))
        assert int(mw.sum()) >= min_rows, "window at row %d holds only %d comparable rows" % (lo, int(mw.sum()))
        for c in want:
            got, exp = _np(rep[c][a:b]), want[c][mw]
            assert got.shape == exp.shape, "window at row %d, column %s: %d rows, oracle %d" % (lo, c, got.shape[0], exp.shape[0])
            if c in float_cols:
                same = np.array_equal(got.view(np.uint64), exp.view(np.uint64))
            else:
                same = np.array_equal(got, exp)
            if not same:
                bad = int(np.flatnonzero(got.view(np.uint64) != exp.view(np.uint64))[0]) if c in float_cols else int(np.flatnonzero(got != exp)[0])
                raise AssertionError("window at row %d, column %s differs from the oracle at table row %d: got %r, oracle %r"
                                     % (lo, c, a + bad, got[bad], exp[bad]))
        done.append({"row_lo": lo, "rows": wrows, "byte_offset": w["byte0"], "table_rows_compared": int(mw.sum())})
    return {"ok": True, "windows": done}
