"""The engine's own row layout (csrc/layout.hip): an uploaded batch, and an adopted one after epi_batch_realign, keeps its rows
at offsets congruent to their start position modulo 16 -- the bytes of every row unchanged, filler between them -- and no
result depends on it."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
import synth_np
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ea():
    import epialleler_amd
    return epialleler_amd


def _view(ea, bam):
    """(xm bytes, off[n+1], len[n]) of the batch as the kernels read it, copied to the host"""
    import torch
    lib = ea._lib.load()
    h = bam.batch()
    xm, off, ln, nb = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    ea._lib.check(lib.epi_batch_view(h, C.byref(xm), C.byref(off), C.byref(ln), C.byref(nb)))
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def fetch(ptr, dtype, count):
        out = np.empty(count, dtype=dtype)
        if count:
            assert hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0
        return out
    return fetch(xm, np.uint8, nb.value), fetch(off, np.int64, bam.n + 1), fetch(ln, np.int32, bam.n), lib.epi_batch_layout(h)


def _check_layout(t, xm, off, ln, modulus):
    n = len(t["start"])
    want_len = np.diff(t["off"]).astype(np.int32)
    assert np.array_equal(ln, want_len)
    assert np.all(off[:-1] % modulus == t["start"].astype(np.int64) % modulus)
    assert np.all(off[:-1] + ln <= off[1:]) and np.all(off[1:] - off[:-1] - ln < modulus)
    for x in list(range(min(n, 50))) + list(range(max(n - 50, 0), n)) + list(range(0, n, max(n // 200, 1))):
        assert np.array_equal(xm[off[x]:off[x] + ln[x]], t["xm"][t["off"][x]:t["off"][x + 1]]), x
    fill = np.ones(len(xm), dtype=bool)                       # everything that is not a row is filler
    for x in range(n):
        fill[off[x]:off[x] + ln[x]] = False
    assert np.all(xm[fill] == 0xFB)


@pytest.mark.parametrize("case", ["ragged", "tiny", "long", "empty_rows"])
def test_uploaded_batch_is_position_congruent(ea, case):
    rng = np.random.default_rng(11)
    if case == "ragged":
        t = synth_np.random_templates(rng, 3000, 0, 400, 3, 20000)
    elif case == "tiny":
        t = synth_np.random_templates(rng, 500, 1, 15, 2, 300)
    elif case == "long":
        t = synth_np.random_templates(rng, 40, 3000, 9000, 2, 30000)
    else:
        t = synth_np.random_templates(rng, 800, 0, 3, 1, 100)
    bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
    try:
        xm, off, ln, layout = _view(ea, bam)
        assert layout == 16
        _check_layout(t, xm, off, ln, 16)
        c = H.CONTEXT_TO_BASES["CG"]
        p = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
        got = ea.rcpp_threshold_reads(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
        assert np.array_equal(np.asarray(got).astype(np.int32), p)
        beta = ea.rcpp_get_xm_beta(bam, c["ctx_meth"], c["ctx_unmeth"])
        assert np.array_equal(np.asarray(beta).view(np.uint64), orc.get_xm_beta(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"]).view(np.uint64))
        H.assert_reports_equal(dict(ea.rcpp_cx_report(bam, p, "Z")), orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, "Z"))
        H.assert_reports_equal(dict(ea.rcpp_mhl_report(bam, "Zz", 0, 0, 0.1)),
                               orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1), float_cols=("length", "lmhl"))
    finally:
        bam.close()


def test_adopted_batch_zero_copy_and_realigned(ea):
    import torch
    from epialleler_amd._lib import EpihipError
    rng = np.random.default_rng(12)
    t = synth_np.random_templates(rng, 5000, 20, 320, 2, 40000)
    nb = int(t["off"][-1])

    def adopt(realign):
        xm = torch.full(((nb + 15) // 16 * 16 + 64,), 0xFB, dtype=torch.uint8, device="cuda:0")
        xm[:nb] = torch.from_numpy(t["xm"]).cuda()
        return ea.ProcessedBam.from_device(xm, nb, torch.from_numpy(t["off"]).cuda(), torch.from_numpy(t["rname"]).cuda(),
                                           torch.from_numpy(t["strand"]).cuda(), torch.from_numpy(t["start"]).cuda(), realign=realign)
    want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, "Z")
    a = adopt(False)
    try:
        xm, off, ln, layout = _view(ea, a)
        assert layout == 0 and np.array_equal(off, t["off"]) and np.array_equal(xm[:nb], t["xm"])       # the caller's own memory
        H.assert_reports_equal(dict(ea.rcpp_cx_report(a, None, "Z")), want)
        with pytest.raises(EpihipError, match="before the first report"):
            ea._lib.check(ea._lib.load().epi_batch_realign(a.batch(), None))
    finally:
        a.close()
    b = adopt(True)
    try:
        xm, off, ln, layout = _view(ea, b)
        assert layout == 16
        _check_layout(t, xm, off, ln, 16)
        b.drop_source()                                        # the engine reads its own copy: the adopted tensors may go
        assert "xm" not in b.dev
        torch.cuda.empty_cache()
        H.assert_reports_equal(dict(ea.rcpp_cx_report(b, None, "Z")), want)
        H.assert_reports_equal(dict(ea.rcpp_mhl_report(b, "Zz", 0, 0, 0.1)),
                               orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1), float_cols=("length", "lmhl"))
    finally:
        b.close()
