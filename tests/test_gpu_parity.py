"""Parity tests proper: the HIP engine (through the C ABI of libepihip.so) against
the CPU oracle, bit-exact for every integer column and for the float64 columns
(per-read beta, lMHL length/lmhl are compared bitwise; the stated tolerance is 0).
Run on an MI355X with `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
import synth_np
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

C2B = H.CONTEXT_TO_BASES
ALL_CTX = ("CG", "CHG", "CHH", "CxG", "CX")


@pytest.fixture(scope="module")
def ea():
    import epialleler_amd
    return epialleler_amd


def pb(ea, t, levels=None):
    return ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"], levels or t.get("levels"))


def o_thr(t, ctx="CG", min_n=2, min_beta=0.5, max_oo=0.1):
    c = C2B[ctx]
    return orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"],
                               min_n, min_beta, max_oo)


def check_all(ea, t, pass_variants=True, mhl=True, contexts=ALL_CTX):
    """threshold + beta + cx + mhl on one template set against the oracle."""
    bam = pb(ea, t)
    try:
        for ctx in contexts:
            c = C2B[ctx]
            got = ea.rcpp_threshold_reads(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
            want = o_thr(t, ctx)
            assert np.array_equal(got.astype(np.int32), want), ("threshold", ctx)
            gb = ea.rcpp_get_xm_beta(bam, c["ctx_meth"], c["ctx_unmeth"])
            wb = orc.get_xm_beta(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"])
            assert np.array_equal(gb.view(np.uint64), wb.view(np.uint64)), ("beta", ctx)     # tolerance 0 (<= 1e-6 required)
            passes = [None]
            if pass_variants:
                passes.append(want)
            for p in passes:
                got = ea.rcpp_cx_report(bam, p, c["ctx_meth"])
                want_r = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, c["ctx_meth"])
                H.assert_reports_equal(dict(got), want_r)
            # thresholding fused into the tile kernel (generateCytosineReport's default path): same table, same flags
            for rctx in (ctx, "CX" if ctx == "CG" else "CG"):
                letters = C2B[rctx]["ctx_meth"]
                H.dirty_allocator(bam)                   # the flag column must be written for every row, empty reads too
                got, gp = ea.cytosine_report_fused(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1,
                                                   letters, return_pass=True)
                assert np.array_equal(gp.astype(np.int32), want), ("fused pass", ctx)
                H.assert_reports_equal(dict(got), orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], want, letters))
            if mhl:
                for hmax, hmin, moo in ((0, 0, 0.1), (1, 0, 0.1), (3, 2, 1.0)):
                    got = ea.rcpp_mhl_report(bam, c["ctx_meth"] + c["ctx_unmeth"], hmax, hmin, moo)
                    want_m = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"],
                                            c["ctx_meth"] + c["ctx_unmeth"], hmax, hmin, moo)
                    H.assert_reports_equal(dict(got), want_m, float_cols=("length", "lmhl"))
    finally:
        bam.close()


# ---- reference fixtures ------------------------------------------------------------------------

@pytest.mark.parametrize("name,kw", [
    ("amplicon010meth.bam", {}),                      # BASELINE config 1
    ("amplicon000meth.bam", {}),
    ("amplicon100meth.bam", {}),
    ("capture.bam", {}),
    ("capture.bam", dict(min_mapq=30, min_baseq=20)),
    ("capture.bam", dict(trim=3)),
    ("dragen-se-unsort-xg-xm.bam", {}),
    ("dragen-pe-namesort-xg-xm.bam", {}),
])
def test_fixture_bams(ea, name, kw):
    t = H.bam(name, **kw)
    check_all(ea, t)


def test_capture_golden_numbers_through_gpu(ea):
    # the reference's own known-answer values, straight from the GPU path
    b = H.bam("capture.bam")
    bam = pb(ea, b)
    cg = ea.generateCytosineReport(bam)
    cx = ea.generateCytosineReport(bam, threshold_reads=False, report_context="CX")
    ev = lambda pre: H.expected_values("generateCytosineReport", pre)
    assert [cg.nrow, 6] == ev("dim(cg.report)") and [cx.nrow, 6] == ev("dim(cx.report)")
    assert [int(cg["meth"].sum())] == ev("sum(cg.report$meth)") and [int(cg["unmeth"].sum())] == ev("sum(cg.report$unmeth)")
    assert [int(cx["meth"].sum())] == ev("sum(cx.report$meth)") and [int(cx["unmeth"].sum())] == ev("sum(cx.report$unmeth)")
    m = ea.generateMhlReport(bam)
    assert int(m["coverage"].sum()) == 20219
    np.testing.assert_allclose([m["length"].sum(), m["lmhl"].sum()], [229119.960, 2666.456], rtol=2e-7)
    m1 = ea.generateMhlReport(bam, max_haplotype_window=1)
    cgu = ea.generateCytosineReport(bam, threshold_reads=False)
    assert np.array_equal(m1["lmhl"], cgu["meth"] / (cgu["meth"] + cgu["unmeth"]))
    assert cx.levels["context"][6] == "CG" and cx.levels["strand"] == ("+", "-") and cx.levels["rname"] == tuple(b["levels"])


def test_config1_bam_file_to_report(ea):
    """BASELINE config 1 end to end: BAM on disk -> host packer -> HBM -> report (478 rows, 632 / 6449)."""
    import os
    path = os.path.join(H.GOLDEN, "bam", "amplicon010meth.bam")
    rep = ea.generateCytosineReport(path)
    assert [rep.nrow, int(rep["meth"].sum()), int(rep["unmeth"].sum())] == [478, 632, 6449]
    t = H.bam("amplicon010meth.bam")
    H.assert_reports_equal(dict(rep), orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], o_thr(t), "Z"))
    m = ea.generateMhlReport(path, max_outofcontext_beta=1)
    assert int(m["coverage"].sum()) == 7081
    q = ea.generateCytosineReport(os.path.join(H.GOLDEN, "bam", "capture.bam"), min_mapq=30, min_baseq=20)
    assert [q.nrow, int(q["meth"].sum()), int(q["unmeth"].sum())] == [15197, 4830, 15062]


# ---- toy / edge cases ----------------------------------------------------------------------------

def test_toys(ea):
    check_all(ea, H.templates_from_xm(["ZZZzzZZZ", "ZZzzzzZZ"] * 3, [1, 2, 3, 4, 5, 6], [1, 2] * 3))
    check_all(ea, H.templates_from_xm(["zZZZ"], [1], [1]))
    check_all(ea, H.templates_from_xm(["Z.hZz", "Z.hZz", "..hZz"], [1, 1, 1], [1, 1, 1]))
    check_all(ea, H.templates_from_xm(["Zz", "xz"], [1, 1], [1, 1]))
    rng = np.random.default_rng(3)
    xms = ["Z" * 10] + ["".join(rng.permutation(list("Zzzzzzzzzz"))) for _ in range(999)]
    check_all(ea, H.templates_from_xm(xms, [1] * 1000, [1] * 1000))


def test_empty_and_degenerate(ea):
    empty = {"xm": np.zeros(0, np.uint8), "off": np.zeros(1, np.int64), "rname": np.zeros(0, np.int32),
             "strand": np.zeros(0, np.int32), "start": np.zeros(0, np.int32)}
    bam = pb(ea, empty)
    assert ea.rcpp_threshold_reads(bam, "Z", "z", "XH", "xh", 2, 0.5, 0.1).size == 0
    assert ea.rcpp_get_xm_beta(bam, "Z", "z").size == 0
    assert ea.rcpp_cx_report(bam, None, "Z").nrow == 0
    assert ea.rcpp_mhl_report(bam, "Zz", 0, 0, 0.1).nrow == 0
    # rows of length zero, rows of only filler, a single byte
    t = H.templates_from_xm(["", "--", "z", "", "Z-z", ""], [5, 5, 6, 7, 7, 9], [1, 2, 1, 2, 1, 1])
    check_all(ea, t)
    # everything is '.' (no rows at all) and everything is filler
    check_all(ea, H.templates_from_xm(["....", "...."], [1, 3], [1, 2]))
    check_all(ea, H.templates_from_xm(["++--", "-+-+"], [1, 3], [1, 2]))


@pytest.mark.parametrize("depth", [254, 255, 256, 257, 300, 600])
def test_u8_counter_limit(ea, depth):
    """Single-context CX reports keep u8 counters per position and drop the u16 copy when no position is covered by
    more than 255 rows (row x + 255 starts behind the end of row x): exactly at, just above and far above that depth,
    with rows piled on one position, staggered by one base, and pile-ups next to ordinary coverage."""
    rng = np.random.default_rng(depth)
    xm1 = "Z" * 7 + "z" + "." * 20 + "ZxZ" + "h" * 5
    piled = H.templates_from_xm([xm1] * depth, [100] * depth, [1 + (i % 2) for i in range(depth)])
    check_all(ea, piled, mhl=False, contexts=("CG",))
    same_strand = H.templates_from_xm([xm1] * depth, [100] * depth, [1] * depth)
    check_all(ea, same_strand, mhl=False, contexts=("CG", "CHG"))
    staggered = H.templates_from_xm([xm1] * depth, [5 + i // 8 for i in range(depth)], [1] * depth)   # 8 rows per start, ~36 long
    check_all(ea, staggered, mhl=False, contexts=("CG",))
    letters = np.frombuffer(b"....hhxzZZ.-", np.uint8)                 # a pile-up in the middle of ordinary coverage
    xs = ["".join(map(chr, rng.choice(letters, int(rng.integers(20, 90))))) for _ in range(400)] + [xm1] * depth
    starts = [int(v) for v in rng.integers(1, 5000, 400)] + [2500] * depth
    strands = [int(v) for v in rng.integers(1, 3, 400 + depth)]
    check_all(ea, H.templates_from_xm(xs, starts, strands), mhl=False, contexts=("CG",))


def test_batch_changed_under_a_remembered_tile_count(ea):
    """The tile count of a batch is remembered between reports (it saves a host round trip inside the index build) and
    verified at the next synchronisation: a batch whose device buffers were rewritten in place is reported, not
    mis-tiled, and the call after that works on the new contents."""
    from epialleler_amd._lib import EpihipError
    rng = np.random.default_rng(5)
    import torch
    t = synth_np.random_templates(rng, 3000, 30, 120, 1, 20000)
    nb = int(t["off"][-1])
    xm = torch.zeros((nb + 15) // 16 * 16, dtype=torch.uint8, device="cuda:0")
    xm[:nb] = torch.from_numpy(t["xm"]).cuda()
    bam = ea.ProcessedBam.from_device(xm, nb, torch.from_numpy(t["off"]).cuda(), torch.from_numpy(t["rname"]).cuda(),
                                      torch.from_numpy(t["strand"]).cuda(), torch.from_numpy(t["start"]).cuda(),
                                      realign=False)         # strictly zero-copy: the engine reads the caller's columns every time
    try:
        for _ in range(2):                                   # the second call runs on the remembered count
            H.assert_reports_equal(dict(ea.rcpp_cx_report(bam, None, "Z")),
                                   orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, "Z"))
        bam.dev["start"][1500:] += 100000                    # still sorted; far more tiles than before
        t2 = dict(t)
        t2["start"] = t["start"].copy()
        t2["start"][1500:] += 100000
        with pytest.raises(EpihipError, match="changed"):
            ea.rcpp_cx_report(bam, None, "Z")
        H.assert_reports_equal(dict(ea.rcpp_cx_report(bam, None, "Z")),
                               orc.cx_report(t2["xm"], t2["off"], t2["rname"], t2["strand"], t2["start"], None, "Z"))
    finally:
        bam.close()


@pytest.mark.parametrize("every,long_len", [(97, 1000), (500, 2500), (13, 400), (301, 7000)])
def test_tail_of_long_templates(ea, every, long_len):
    # The lane shape of the tile kernels follows the bulk of the rows (length histogram), not the longest one: rows longer than
    # the shape holds go slice by slice -- with fused thresholding twice (class totals, then calls) -- and take their wavefront
    # step along; candidate rows that end in front of a tile are skipped.  Short rows + a tail, every function, every route.
    rng = np.random.default_rng(every * 7 + long_len)
    t = synth_np.with_long_tail(synth_np.random_templates(rng, 6000, 100, 310, 2, 60000), every, long_len, first=int(rng.integers(0, every)))
    check_all(ea, t, contexts=("CG", "CHH"))
    t = synth_np.with_long_tail(synth_np.generate_uniform(n_total=9000, mean_len=300, n_chr=2, ragged=False, gap_every=0), every, long_len)
    check_all(ea, t, contexts=("CG",))


def test_ragged_random(ea):
    rng = np.random.default_rng(11)
    for n, mx, span in ((1, 50, 100), (7, 40, 60), (300, 400, 3000), (2000, 700, 20000), (500, 33, 400)):
        check_all(ea, synth_np.random_templates(rng, n, 0, mx, 3, span))


@pytest.mark.parametrize("mean_len", [20, 100, 250, 600, 1300, 4000])
def test_per_read_kernels_every_group_size(ea, mean_len):
    """The wide per-read kernel picks 2..64 lanes per read from the mean read length; ragged lengths up to 2.5x the mean
    make some reads longer than one round of a group's loads (the tail loop) and some empty."""
    rng = np.random.default_rng(mean_len)
    n = max(200, 400000 // mean_len)
    t = synth_np.random_templates(rng, n, 0, int(2.5 * mean_len), 2, 50000, p_garbage=0.02)
    bam = pb(ea, t)
    try:
        for ctx in ("CG", "CHG", "CHH", "CxG", "CX"):
            c = C2B[ctx]
            for min_n, min_beta, max_oo in ((2, 0.5, 0.1), (1, 0.0, 1.0), (5, 0.9, 0.0)):
                got = ea.rcpp_threshold_reads(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], min_n, min_beta, max_oo)
                want = o_thr(t, ctx, min_n, min_beta, max_oo)
                assert np.array_equal(got.astype(np.int32), want), ("threshold", ctx, min_n)
            gb = ea.rcpp_get_xm_beta(bam, c["ctx_meth"], c["ctx_unmeth"])
            wb = orc.get_xm_beta(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"])
            assert np.array_equal(gb.view(np.uint64), wb.view(np.uint64)), ("beta", ctx)
        # a class string with a repeated letter (weight 2) takes the general kernel
        got = ea.rcpp_threshold_reads(bam, "ZZ", "z", "XH", "xh", 2, 0.5, 0.1)
        want = orc.threshold_reads(t["xm"], t["off"], "ZZ", "z", "XH", "xh", 2, 0.5, 0.1)
        assert np.array_equal(got.astype(np.int32), want)
    finally:
        bam.close()


def test_upload_keeps_the_last_bytes(ea):
    """Regression: the threaded staging copy of epi_batch_upload split len bytes into 4 parts of len/4 rounded to 4 KiB and
    dropped the last len % 4 bytes whenever len/4 was a multiple of 4096 (found by scratch/fuzz.py: one CX row short)."""
    for nbytes in (4 * 1024 * 4096 + 3, 4 * 1200 * 4096 + 1, 2 * 64 * 1024 * 1024 + 4 * 512 * 4096 + 2):
        n = nbytes // 1000
        lens = np.full(n, 1000, np.int64)
        lens[-1] += nbytes - int(lens.sum())
        off = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        xm = np.full(nbytes, 0x1C, np.uint8)              # '.' everywhere
        xm[-3:] = 0x17                                    # the last bases of the last read: 'Z'
        t = dict(xm=xm, off=off, rname=np.ones(n, np.int32), strand=np.ones(n, np.int32),
                 start=(1 + 10 * np.arange(n)).astype(np.int32))
        bam = pb(ea, t)
        try:
            gb = ea.rcpp_get_xm_beta(bam, "Z", "z")
            assert gb[-1] == 1.0 and gb[:-1].max() == 0.0, nbytes
            got = ea.rcpp_cx_report(bam, None, "Z")
            assert got["pos"].tolist()[-3:] == [int(t["start"][-1]) + int(lens[-1]) - 3 + i for i in range(3)], nbytes
        finally:
            bam.close()


def test_tile_boundaries_and_large_positions(ea):
    rng = np.random.default_rng(5)
    # starts straddling multiples of the tile sizes (512/1024), near 2^31, and at position 0/1
    for base in (1, 1000, 1024 * 7 - 150, 512 * 33 - 10, 2 ** 31 - 3000):
        t = synth_np.random_templates(rng, 200, 1, 300, 2, 1500)
        t["start"] = (t["start"].astype(np.int64) + base - 1).astype(np.int32)
        check_all(ea, t, mhl=True, contexts=("CG", "CX"))


def test_all_byte_values_and_odd_context_strings(ea):
    # any nibble can reach the kernels (e.g. nibble 9 counts twice in coverage, lower-casing maps 4->'.')
    rng = np.random.default_rng(17)
    t = synth_np.random_templates(rng, 400, 0, 200, 2, 800, p_garbage=0.3)
    check_all(ea, t)
    bam = pb(ea, t)
    # duplicated / unusual letters in context strings (the reference counts duplicates twice)
    for cm, cu, om, ou in (("ZZ", "z", "XH", "xh"), ("Zz", "zZ", "", ""), ("U", "u", "Z", "z"), (".", "-", "+", "h")):
        got = ea.rcpp_threshold_reads(bam, cm, cu, om, ou, 3, 0.4, 0.3)
        want = orc.threshold_reads(t["xm"], t["off"], cm, cu, om, ou, 3, 0.4, 0.3)
        assert np.array_equal(got.astype(np.int32), want)
        gb = ea.rcpp_get_xm_beta(bam, cm, cu)
        assert np.array_equal(gb.view(np.uint64), orc.get_xm_beta(t["xm"], t["off"], cm, cu).view(np.uint64))
    for ctx in ("Z", "ZX", "H", "ZXHU", "z", ".", "Zz"):
        for p in (None, (rng.random(400) < 0.5).astype(np.int32)):
            H.assert_reports_equal(dict(ea.rcpp_cx_report(bam, p, ctx)),
                                   orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, ctx))
    for ctx in ("Zz", "ZzXx", "Z", "HhUu", "Xx."):
        H.assert_reports_equal(dict(ea.rcpp_mhl_report(bam, ctx, 0, 0, 0.1)),
                               orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], ctx, 0, 0, 0.1),
                               float_cols=("length", "lmhl"))
    bam.close()


def test_pass_na_counts_as_true(ea):
    t = H.templates_from_xm(["Zz.", "zZ.", "ZZz"], [1, 2, 2], [1, 1, 2])
    p = np.asarray([0, np.iinfo(np.int32).min, 1], np.int32)          # FALSE, NA, TRUE
    bam = pb(ea, t)
    H.assert_reports_equal(dict(ea.rcpp_cx_report(bam, p, "Z")),
                           orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, "Z"))


def test_deep_pileup_and_many_rnames(ea):
    rng = np.random.default_rng(23)
    # amplicon-like: 6000 reads on the same ~400 positions (one tile gets everything)
    t = synth_np.random_templates(rng, 6000, 100, 400, 1, 40)
    check_all(ea, t, contexts=("CG", "CX"))
    # many reference sequences with a handful of reads each, some rname ids unused
    t = synth_np.random_templates(rng, 1500, 10, 200, 400, 3000)
    t["rname"] = (t["rname"] * 3).astype(np.int32)
    check_all(ea, t, contexts=("CG",))


@pytest.fixture
def hook_env(ea, monkeypatch):
    """Sets EPIHIP_* test hooks inside this process: the library reads its switches once, so it is told to re-read
    them after every change (epi_options_reload) and once more when the environment has been restored."""
    lib = ea._lib.load()

    def setenv(name, value):
        monkeypatch.setenv(name, value)
        lib.epi_options_reload()
    yield setenv
    monkeypatch.undo()
    lib.epi_options_reload()


def test_heavy_tiles_are_split(ea, hook_env):
    """Ultra-deep tiles are set aside and split over many workgroups (k_cx_heavy); force that path on small data."""
    hook_env("EPIHIP_HEAVY_ROWS", "300")
    rng = np.random.default_rng(31)
    t = synth_np.random_templates(rng, 5000, 50, 400, 2, 60)            # two pile-ups, every tile heavy
    check_all(ea, t, mhl=True, contexts=("CG", "CX"))
    t = synth_np.random_templates(rng, 3000, 0, 700, 3, 9000)           # a mix of heavy and ordinary tiles
    check_all(ea, t, mhl=True, contexts=("CG", "CX"))
    check_all(ea, H.bam("amplicon010meth.bam"), mhl=True, contexts=("CG", "CX"))
    hook_env("EPIHIP_HEAVY_ROWS", "70")
    check_all(ea, H.bam("amplicon010meth.bam"), mhl=True, contexts=("CG",))


def test_lmhl_sum_width_and_lowered_heavy_threshold(ea):
    """lMHL pass 2 keeps its sums in u32 LDS arrays when rows x S(largest h) stays below 2^31; deep tiles are then split
    earlier (heavy path) so that this holds for every chunk.  Haplotypes of ~100 sites on 6000-deep pile-ups sit exactly
    in that regime; ~250 sites push the same data to the u64 kernel."""
    rng = np.random.default_rng(41)
    for max_len, alphabet in ((300, "zZ."), (300, "zzZ"), (700, "zZ")):
        t = synth_np.random_templates(rng, 6000, max_len - 50, max_len, 2, 80, alphabet=alphabet)
        bam = pb(ea, t)
        for hmax in (0, 7):
            got = ea.rcpp_mhl_report(bam, "Zz", hmax, 0, 1.0)
            want = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", hmax, 0, 1.0)
            H.assert_reports_equal(dict(got), want, float_cols=("length", "lmhl"))
        bam.close()


def test_two_kernel_lmhl_index_boundaries(ea, hook_env):
    """The three index computations of the two-kernel lMHL path ("index bounds"): reads whose last
    byte is the last position of a 512-position tile and the last byte of the batch, reads that end one position
    before / behind that, a long read (per-block records) that fills its last 2 KiB record block to the final byte,
    and stretches that run up to the last byte -- with the one-pass kernel switched off, plain and wavefront-per-read."""
    def reads(lens, ends):
        xms = ["".join("Zz"[(i * 7 + k) % 5 == 0] if (i + k) % 3 else "." for i in range(n)) for k, n in enumerate(lens)]
        xms = [x[:-3] + "ZZZ" for x in xms]                              # a stretch up to the last byte
        return H.templates_from_xm(xms, [e - n + 1 for n, e in zip(lens, ends)], [1 + (k & 1) for k in range(len(lens))])
    for multi in ("", "1"):
        hook_env("EPIHIP_MHL_FUSED", "0")
        if multi:
            hook_env("EPIHIP_MHL_MULTI", "1")
        for lens, ends in (([100, 300, 511, 512, 513, 700], [1023, 1024, 1535, 2047, 2048, 2559]),   # ends on / around tile ends
                           ([2048, 4096, 6144, 2047, 2049], [4095, 8191, 12287, 12288, 16383]),         # whole record blocks
                           ([5000] * 4, [8191, 8192, 8193, 10239])):
            t = reads(lens, ends)
            bam = pb(ea, t)
            for hmax, hmin in ((0, 0), (3, 2)):
                got = ea.rcpp_mhl_report(bam, "Zz", hmax, hmin, 1.0)
                want = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", hmax, hmin, 1.0)
                H.assert_reports_equal(dict(got), want, float_cols=("length", "lmhl"))
            bam.close()


def test_long_reads(ea):
    rng = np.random.default_rng(29)
    t = synth_np.random_templates(rng, 40, 5000, 12000, 2, 30000, alphabet="......hhxzzZZZHXuU-")
    check_all(ea, t, contexts=("CG", "CX"))
    # test_generateMhlReport.R:102-122: two 10000-base reads
    xms = ["".join(rng.choice(list("Zzzzzzzzzz"), 10000)) for _ in range(2)]
    t = H.templates_from_xm(xms, [1, 1], [1, 1])
    bam = pb(ea, t)
    m = ea.rcpp_mhl_report(bam, "Zz", 1, 0, 0.1)
    cg = ea.rcpp_cx_report(bam, None, "Z")
    assert [int(m["coverage"].sum()), m["length"].sum()] == [20000, 100000000]
    assert np.array_equal(m["lmhl"], cg["meth"] / (cg["meth"] + cg["unmeth"]))


def test_unsorted_rows_are_rejected(ea):
    t = H.templates_from_xm(["zz", "zz", "zz"], [1, 1, 1], [1, 1, 1])
    t["start"] = np.asarray([1, 10, 2], np.int32)
    bam = pb(ea, t)
    with pytest.raises(ea.EpihipError) as ei:
        ea.rcpp_cx_report(bam, None, "Z")
    assert ei.value.code == 3
    with pytest.raises(ea.EpihipError):
        ea.rcpp_mhl_report(bam, "Zz", 0, 0, 0.1)
    # per-read functions do not need sorted rows
    assert ea.rcpp_threshold_reads(bam, "Z", "z", "", "", 0, 0.0, 1.0).size == 3
    t["start"] = np.asarray([1, 2, 3], np.int32)
    t["strand"] = np.asarray([1, 3, 1], np.int32)
    with pytest.raises(ea.EpihipError):
        ea.rcpp_cx_report(pb(ea, t), None, "Z")


def test_match_arg_and_defaults(ea):
    t = H.templates_from_xm(["Zz"], [1], [1])
    with pytest.raises(ValueError):
        ea.generateCytosineReport(pb(ea, t), threshold_context="CpG")
    r = ea.generateCytosineReport(pb(ea, t), threshold_reads=False)
    assert r.nrow == 2 and list(r.keys()) == ["rname", "strand", "pos", "context", "meth", "unmeth"]


# ---- synthetic workload (BASELINE configs 2-5 shape) -----------------------------------------------

def test_synth_matches_numpy_mirror(ea):
    import torch
    from epialleler_amd import synth
    for kw in (dict(n_total=3000, read_len=300), dict(n_total=1000, read_len=97, n_chr=3, depth=7, gap_from=40, gap_len=11),
               dict(n_total=5000, read_len=300, row_first=1234, n=2000)):
        bam = synth.generate_device(**kw)
        ref = synth_np.generate(**kw)
        d = bam.dev
        assert np.array_equal(d["xm"][:bam.nbytes].cpu().numpy(), ref["xm"])
        for k in ("off", "rname", "strand", "start"):
            assert np.array_equal(d[k].cpu().numpy(), ref[k]), k
        bam.close()
    torch.cuda.synchronize()


def test_uniform_synth_matches_numpy_mirror(ea):
    """The SURVEY-8d-conformant generator (uniform starts sorted, ragged lengths, gapped templates; bench cfg2u)."""
    from epialleler_amd import synth
    for kw in (dict(n_total=4000), dict(n_total=9000, n_chr=3, row_first=2500, n=5000, gap_every=2, gap_len=31),
               dict(n_total=1500, mean_len=120, depth=9, gap_every=0), dict(n_total=5000, n_chr=3, gap_every=0, ragged=False)):
        bam = synth.generate_device_uniform(**kw)
        ref = synth_np.generate_uniform(**kw)
        d = bam.dev
        assert np.array_equal(d["xm"][:bam.nbytes].cpu().numpy(), ref["xm"])
        for k in ("off", "rname", "strand", "start"):
            assert np.array_equal(d[k].cpu().numpy(), ref[k]), k
        bam.close()


def test_uniform_synth_medium_parity(ea):
    check_all(ea, synth_np.generate_uniform(n_total=20000), contexts=("CG", "CX"))


@pytest.mark.parametrize("pile", [300, 700, 20000])
def test_pileup_inside_a_wgs_stream(ea, pile):
    """One amplicon-like pile-up in WGS-like data (bench cfg2p / cfg4d): the lean CX kernel keeps every ordinary tile and
    lists only the tiles around the pile-up for the general kernel (a 20 000-row pile-up additionally takes the heavy-tile
    split); the one-pass lMHL kernel folds its u8 counters there.  Fixed-length and ragged / gapped rows."""
    n = 30000 + pile
    check_all(ea, synth_np.generate_uniform(n_total=n, gap_every=0, ragged=False, pileup=(n // 2 + 77, pile)), mhl=True, contexts=("CG", "CX"))
    if pile == 700:
        check_all(ea, synth_np.generate_uniform(n_total=n, pileup=(5000, pile), seed=9), mhl=True, contexts=("CG",))


@pytest.mark.parametrize("kw", [dict(n_total=20000, read_len=300), dict(n_total=6000, read_len=300, gap_from=150, gap_len=50),
                                dict(n_total=300, read_len=10000, n_chr=2)])
def test_synth_medium_parity(ea, kw):
    t = synth_np.generate(**kw)
    check_all(ea, t, contexts=("CG", "CX"))


def test_host_drop_in_entry_points(ea):
    """The four host-pointer functions an R shim binds, called as C."""
    from epialleler_amd import _lib
    lib = _lib.load()
    t = H.bam("amplicon010meth.bam")
    n = t["off"].size - 1
    vp = lambda a: C.c_void_p(a.ctypes.data)
    out = np.zeros(n, np.int32)
    _lib.check(lib.epi_threshold_reads(vp(t["xm"]), vp(t["off"]), n, b"Z", b"z", b"XH", b"xh", 2, 0.5, 0.1, vp(out)))
    assert np.array_equal(out, o_thr(t)) and int(out.sum()) == 48
    beta = np.zeros(n, np.float64)
    _lib.check(lib.epi_get_xm_beta(vp(t["xm"]), vp(t["off"]), n, b"Z", b"z", vp(beta)))
    assert np.array_equal(beta, orc.get_xm_beta(t["xm"], t["off"], "Z", "z"))
    tab = _lib.CxTable()
    _lib.check(lib.epi_cx_report(vp(t["xm"]), vp(t["off"]), vp(t["rname"]), vp(t["strand"]), vp(t["start"]), vp(out), n, b"Z", C.byref(tab)))
    got = {k: np.ctypeslib.as_array(getattr(tab, k), shape=(tab.nrow,)).copy() for k in ("rname", "strand", "pos", "context", "meth", "unmeth")}
    lib.epi_cx_table_free(C.byref(tab))
    H.assert_reports_equal(got, orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], out, "Z"))
    assert [got["pos"].size, int(got["meth"].sum()), int(got["unmeth"].sum())] == [478, 632, 6449]     # BASELINE config 1
    mt = _lib.MhlTable()
    _lib.check(lib.epi_mhl_report(vp(t["xm"]), vp(t["off"]), vp(t["rname"]), vp(t["strand"]), vp(t["start"]), n, b"Zz", 0, 0, 0.1, C.byref(mt)))
    gm = {k: np.ctypeslib.as_array(getattr(mt, k), shape=(mt.nrow,)).copy() for k in ("rname", "strand", "pos", "context", "coverage", "length", "lmhl")}
    lib.epi_mhl_table_free(C.byref(mt))
    H.assert_reports_equal(gm, orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1),
                           float_cols=("length", "lmhl"))
    # error path: NULL offsets
    assert lib.epi_threshold_reads(None, None, 1, b"Z", b"z", b"", b"", 0, 0.0, 0.0, vp(out)) == 1
    assert b"bad arguments" in lib.epi_last_error()


# ---- long-read (MM/ML) BAM file -> report on the GPU, against the reference's own expected tables ----------------

def test_long_read_bam_file_to_report(ea, tmp_path):
    import test_long_read as LR
    for case in LR.CASES:
        LR.check_case(case, lambda p, **kw: ea.preprocessBam(p, **kw),
                      lambda bam, ctx: dict(ea.rcpp_cx_report(bam, None, ctx)), tmp_path)
    # and the R-level entry point with its long-read arguments (R/generateCytosineReport.R:164-208)
    case = LR.CASES[0]
    path = LR.case_bam(case, str(tmp_path / "lr0.bam"))
    rep = ea.generateCytosineReport(path, threshold_reads=False, report_context="CX", min_prob=160, highest_prob=False)
    want = case["reports"][1]["checks"][0]["value"]
    assert rep["meth"].tolist() == want["meth"] and rep["pos"].tolist() == want["pos"]
