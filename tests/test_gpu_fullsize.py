"""BASELINE-size checks through size-independent properties (the oracle cannot run
10 M reads in test time): determinism, strict (rname,pos,strand) order, and exact
agreement with the oracle on windows of the genome cut out of the full-size input."""
import numpy as np
import pytest

import helpers as H
from oracle import oracle as orc
from oracle import windows as W

pytestmark = pytest.mark.gpu


def _window_rows(bam, lo, hi):
    """Host copy of rows [lo,hi) of a device-resident batch."""
    return W.window_rows(bam.dev, lo, hi)


def _check_windows(rep, bam, n, oracle_fn, float_cols=(), L=300, wrows=20000, starts=None):
    """Strict row order of the whole table + exact agreement with the oracle on row windows (oracle/windows.py)."""
    return W.check_windows(rep, bam.dev, n, oracle_fn, float_cols=float_cols, L=L, wrows=wrows, starts=starts)


def test_config2_cytosine_report_10M():
    """BASELINE config 2: 10 M PE150 templates, generateCytosineReport on one MI355X."""
    import epialleler_amd as ea
    from epialleler_amd import synth
    n = 10_000_000
    bam = synth.generate_device(n_total=n, read_len=300)
    c = H.CONTEXT_TO_BASES["CG"]
    rep = ea.generateCytosineReport(bam, threshold_reads=True)
    rep2 = ea.generateCytosineReport(bam, threshold_reads=True)
    for k in rep:
        assert np.array_equal(rep[k], rep2[k])          # idempotent / deterministic
    assert 5_000_000 < rep.nrow < 9_000_000
    def oracle_fn(w):
        p = orc.threshold_reads(w["xm"], w["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
        return orc.cx_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], p, "Z")
    _check_windows(rep, bam, n, oracle_fn)
    cx = ea.generateCytosineReport(bam, threshold_reads=False, report_context="CX")
    _check_windows(cx, bam, n, lambda w: orc.cx_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], None, "ZXH"))
    bam.close()


def test_config4_mhl_report_scaled():
    """BASELINE config 4 shape (generateMhlReport on PE150), at 5 M templates to bound test time."""
    import epialleler_amd as ea
    from epialleler_amd import synth
    n = 5_000_000
    bam = synth.generate_device(n_total=n, read_len=300, seed=7)
    rep = ea.generateMhlReport(bam)
    _check_windows(rep, bam, n, lambda w: orc.mhl_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], "Zz", 0, 0, 0.1),
                   float_cols=("length", "lmhl"))
    bam.close()


def test_config5_long_reads_scaled():
    """BASELINE config 5 shape: 10 kb templates (one read per wavefront group), 100 k templates."""
    import epialleler_amd as ea
    from epialleler_amd import synth
    n = 100_000
    bam = synth.generate_device(n_total=n, read_len=10000, seed=11)
    rep = ea.generateCytosineReport(bam, threshold_reads=False)
    key = rep["rname"].astype(np.int64) * (1 << 33) + rep["pos"].astype(np.int64) * 2 + (rep["strand"] - 1)
    assert np.all(np.diff(key) > 0)
    w = _window_rows(bam, 0, 3000)
    want = orc.cx_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], None, "Z")
    lim = int(w["start"][-1]) - 1
    mw = want["pos"] <= lim
    mg = (rep["rname"] == 1) & (rep["pos"] <= lim)
    for c in want:
        assert np.array_equal(rep[c][mg], want[c][mw]), c
    bam.close()


# ---- past 4 GiB of xm: 64-bit byte offsets, 32-bit row-relative arithmetic, block indices of long reads ---------------
# (the windows at the END of the batch sit behind offsets > 2^32, 2^33, 2^34)

def test_config3_cytosine_report_100M_one_gpu():
    """BASELINE config 3 on one GPU: 100 M PE150 templates = 30 GB of xm, threshold.reads=TRUE."""
    import epialleler_amd as ea
    from epialleler_amd import synth
    n = 100_000_000
    bam = synth.generate_device(n_total=n, read_len=300, seed=3)
    assert bam.nbytes > 2 ** 34
    c = H.CONTEXT_TO_BASES["CG"]
    rep = ea.generateCytosineReport(bam, threshold_reads=True)
    rep2 = ea.generateCytosineReport(bam, threshold_reads=True, as_device=True)
    for k in rep:
        assert np.array_equal(rep[k], rep2[k].cpu().numpy())    # deterministic
    del rep2
    assert 50_000_000 < rep.nrow < 90_000_000
    def oracle_fn(w):
        p = orc.threshold_reads(w["xm"], w["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
        return orc.cx_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], p, "Z")
    # rows 14.4 M and 28.7 M are where the byte offset crosses 2^32 and 2^33
    _check_windows(rep, bam, n, oracle_fn, starts=(0, 14_316_000, 28_632_000, 57_260_000, n - 20000))
    bam.close()


def test_config4_mhl_report_50M():
    """BASELINE config 4 at full size: 50 M PE150 templates (15 GB), generateMhlReport defaults."""
    import epialleler_amd as ea
    from epialleler_amd import synth
    n = 50_000_000
    bam = synth.generate_device(n_total=n, read_len=300, seed=9)
    rep = ea.generateMhlReport(bam)
    rep2 = ea.generateMhlReport(bam, as_device=True)
    for k in rep:
        a, b = rep[k], rep2[k].cpu().numpy()
        assert np.array_equal(a.view(np.uint64) if a.dtype == np.float64 else a, b.view(np.uint64) if b.dtype == np.float64 else b)
    del rep2
    _check_windows(rep, bam, n, lambda w: orc.mhl_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], "Zz", 0, 0, 0.1),
                   float_cols=("length", "lmhl"), starts=(0, 14_316_000, 28_632_500, n - 20000))
    bam.close()


def test_config5_long_reads_12GB():
    """BASELINE config 5 shape past 4 GiB: 1.2 M templates of 10 kb (12 GB): CX and lMHL, windows from the end."""
    import epialleler_amd as ea
    from epialleler_amd import synth
    n = 1_200_000
    bam = synth.generate_device(n_total=n, read_len=10000, seed=13)
    assert bam.nbytes > 2 ** 33
    rep = ea.generateCytosineReport(bam, threshold_reads=False)
    _check_windows(rep, bam, n, lambda w: orc.cx_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], None, "Z"),
                   L=10000, wrows=1500, starts=(0, 429_000, 859_000, n - 1500))
    del rep
    m = ea.generateMhlReport(bam)
    _check_windows(m, bam, n, lambda w: orc.mhl_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], "Zz", 0, 0, 0.1),
                   float_cols=("length", "lmhl"), L=10000, wrows=1000, starts=(430_000, n - 1000))
    bam.close()


def test_uniform_stream_cytosine_report():
    """The SURVEY-8d-conformant stream (uniform-random starts, ragged lengths, gapped templates; bench cfg2u) at 3 M
    templates: thresholded CG report and lMHL against the oracle on windows."""
    import epialleler_amd as ea
    from epialleler_amd import synth
    n = 3_000_000
    bam = synth.generate_device_uniform(n_total=n, seed=21)
    c = H.CONTEXT_TO_BASES["CG"]
    rep = ea.generateCytosineReport(bam, threshold_reads=True)
    def oracle_fn(w):
        p = orc.threshold_reads(w["xm"], w["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
        return orc.cx_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], p, "Z")
    _check_windows(rep, bam, n, oracle_fn, L=360)
    m = ea.generateMhlReport(bam)
    _check_windows(m, bam, n, lambda w: orc.mhl_report(w["xm"], w["off"], w["rname"], w["strand"], w["start"], "Zz", 0, 0, 0.1),
                   float_cols=("length", "lmhl"), L=360)
    bam.close()


# ---- the streams bench.py times (SURVEY 8d's literal wording: uniform-random starts sorted on the device, fixed L) --------

def _uniform(n, L, seed=42, **kw):
    from epialleler_amd import synth
    return synth.generate_device_uniform(n_total=n, mean_len=L, seed=seed, ragged=False, gap_every=0, **kw)


def test_bench_cfg2_stream_10M():
    """bench.py's headline batch itself (cfg2: seed 42, 10 M templates, L = 300, 4 chromosomes): thresholded CG report,
    un-thresholded CG and CX reports (cfg2n / cfg2cx) against the oracle on windows."""
    import epialleler_amd as ea
    n = 10_000_000
    bam = _uniform(n, 300)
    rep = ea.generateCytosineReport(bam, threshold_reads=True, as_device=True)
    assert 5_000_000 < rep.nrow < 9_000_000
    _check_windows(rep, bam, n, W.oracle_for("cx", True, "Z"))
    del rep
    rep = ea.generateCytosineReport(bam, threshold_reads=False, as_device=True)
    _check_windows(rep, bam, n, W.oracle_for("cx", False, "Z"))
    del rep
    rep = ea.generateCytosineReport(bam, threshold_reads=False, report_context="CX", as_device=True)
    _check_windows(rep, bam, n, W.oracle_for("cx", False, "ZXH"))
    bam.close()


def test_bench_cfg2_stream_three_chromosomes():
    """The N > 1 stream of bench.py (3 chromosomes: every cut of 2 / 4 / 8 equal row ranges lies inside a chromosome), on one GPU."""
    import epialleler_amd as ea
    n = 10_000_000
    bam = _uniform(n, 300, n_chr=3)
    rep = ea.generateCytosineReport(bam, threshold_reads=True, as_device=True)
    _check_windows(rep, bam, n, W.oracle_for("cx", True, "Z"), starts=(0, n // 3 - 10_000, 2 * n // 3 - 10_000, n - 20000))
    bam.close()


def test_bench_cfg4_stream_50M():
    """bench.py's cfg4 batch (50 M templates, 15 GB): generateMhlReport defaults; windows where the byte offset crosses
    2^32 and 2^33 and at the end of the batch."""
    import epialleler_amd as ea
    n = 50_000_000
    bam = _uniform(n, 300)
    rep = ea.generateMhlReport(bam, as_device=True)
    _check_windows(rep, bam, n, W.oracle_for("mhl"), float_cols=("length", "lmhl"),
                   starts=(0, 14_316_000, 28_632_500, n - 20000))
    bam.close()


def test_bench_cfg5_stream_full_50GB():
    """bench.py's cfg5 batch at its full size: 5 M templates of 10 kb = 50 GB of xm, byte offsets beyond 2^35; the
    un-thresholded CG report against the oracle on windows at the start, behind 2^32 / 2^34 / 2^35 bytes and at the end."""
    import epialleler_amd as ea
    n = 5_000_000
    bam = _uniform(n, 10000)
    assert bam.nbytes > 2 ** 35
    rep = ea.generateCytosineReport(bam, threshold_reads=False, as_device=True)
    res = _check_windows(rep, bam, n, W.oracle_for("cx", False, "Z"), L=10000, wrows=1500,
                         starts=(0, 429_500, 1_718_000, 3_440_000, n - 1500))
    assert res["windows"][3]["byte_offset"] > 2 ** 35
    bam.close()
