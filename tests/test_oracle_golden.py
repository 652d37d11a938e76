"""Pins the CPU oracle (oracle/epi_oracle.c + oracle/bamio.py) against every
known-answer value the reference's own RUnit tests hold for the hot path
(tests/golden/expected.json, extracted from reference inst/unitTests/*.R) on the
reference's BAM fixtures (tests/golden/bam/, data files copied from
reference inst/extdata/).  CPU only."""
import numpy as np
import pytest

import helpers as H
from oracle import oracle as orc

C2B = H.CONTEXT_TO_BASES


def thr(b, context="CG", min_n=2, min_beta=0.5, max_oo=0.1):
    c = C2B[context]
    return orc.threshold_reads(b["xm"], b["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"],
                               c["ooctx_unmeth"], min_n, min_beta, max_oo)


def cyt(b, threshold_reads=True, threshold_context="CG", report_context=None, **kw):
    """generateCytosineReport(): R/generateCytosineReport.R:164-208."""
    report_context = report_context or threshold_context
    p = thr(b, threshold_context, **kw) if threshold_reads else None
    return orc.cx_report(b["xm"], b["off"], b["rname"], b["strand"], b["start"], p,
                         C2B[report_context]["ctx_meth"])


def mhl(b, context="CG", hmax=0, hmin=0, max_oo=0.1):
    """generateMhlReport(): R/generateMhlReport.R:170-197."""
    c = C2B[context]
    return orc.mhl_report(b["xm"], b["off"], b["rname"], b["strand"], b["start"],
                          c["ctx_meth"] + c["ctx_unmeth"], hmax, hmin, max_oo)


def ev(prefix, nth=0, section="generateCytosineReport"):
    return H.expected_values(section, prefix, nth)


def test_preprocess_dims():
    # inst/unitTests/test_preprocessBam.R:11-15 -- dim == c(2968, 4)
    b = H.bam("capture.bam")
    assert b["npushed"] == 2968 == H.expected()["survey_probe"]["capture"]["templates"]
    a = H.bam("amplicon010meth.bam")
    assert a["npushed"] == 500 and a["xm"].size == 190455
    assert H.bam("dragen-pe-namesort-xg-xm.bam")["npushed"] == 100
    assert H.bam("dragen-se-unsort-xg-xm.bam")["npushed"] == 100


def test_empty_bam_rejected():
    with pytest.raises(ValueError):
        H.bam("empty.bam")


def test_capture_default_reports():
    # test_generateCytosineReport.R:1-90
    b = H.bam("capture.bam")
    cg = cyt(b)
    cx = cyt(b, threshold_reads=False, report_context="CX")
    assert np.unique(cx["rname"].astype(np.int64) << 32 | cx["pos"]).size * 1 >= 0
    key = cx["rname"].astype(np.int64) * (1 << 33) + cx["pos"].astype(np.int64) * 2 + (cx["strand"] - 1)
    assert np.all(np.diff(key) > 0)                       # sorted, no duplicated (rname,pos,strand)
    assert [int((cx["strand"] == 1).sum()), int((cx["strand"] == 2).sum())] == ev("as.numeric(table(cx.report$strand)")
    assert [int((cx["context"] == k).sum()) for k in (2, 6, 7)] == ev("as.numeric(table(cx.report$context)")
    for s, pre in ((1, 'as.numeric(table(cx.report[strand=="+"]'), (2, 'as.numeric(table(cx.report[strand=="-"]')):
        assert [int(((cx["context"] == k) & (cx["strand"] == s)).sum()) for k in (2, 6, 7)] == ev(pre)
    assert [cg["pos"].size, 6] == ev("dim(cg.report)")
    assert [cx["pos"].size, 6] == ev("dim(cx.report)")
    assert [int(cg["meth"].sum())] == ev("sum(cg.report$meth)")
    assert [int(cg["unmeth"].sum())] == ev("sum(cg.report$unmeth)")
    assert [int(cx["meth"].sum())] == ev("sum(cx.report$meth)")
    assert [int(cx["unmeth"].sum())] == ev("sum(cx.report$unmeth)")
    for name, code in (("CG", 7), ("CHG", 6), ("CHH", 2)):
        for col in ("meth", "unmeth"):
            want = ev('cx.report[context=="%s", sum(%s)' % (name, col))
            assert H.group_sums(cx, col, code) == want, (name, col)
    assert int(thr(b).sum()) == H.expected()["survey_probe"]["capture"]["pass"]


def test_capture_quality_filtered_reports():
    # test_generateCytosineReport.R:117-210 (min.mapq=30, min.baseq=20)
    b = H.bam("capture.bam", min_mapq=30, min_baseq=20)
    cg = cyt(b)
    cx = cyt(b, threshold_reads=False, report_context="CX")
    assert [cg["pos"].size, 6] == ev("dim(cg.quality)")
    assert [cx["pos"].size, 6] == ev("dim(cx.quality)")
    assert [int((cx["context"] == k).sum()) for k in (2, 6, 7)] == ev("as.numeric(table(cx.quality$context)")
    assert [int(cg["meth"].sum())] == ev("sum(cg.quality$meth)")
    assert [int(cg["unmeth"].sum())] == ev("sum(cg.quality$unmeth)")
    assert [int(cx["meth"].sum())] == ev("sum(cx.quality$meth)")
    assert [int(cx["unmeth"].sum())] == ev("sum(cx.quality$unmeth)")
    for name, code in (("CG", 7), ("CHG", 6), ("CHH", 2)):
        for col, rcol in (("meth", "meth"), ("unmeth", "unmeth"), ("pos", "as.numeric(pos)")):
            want = ev('cx.quality[context=="%s", sum(%s)' % (name, rcol))
            assert H.group_sums(cx, col, code) == want, (name, col)


def test_trim_changes_counts_not_keys():
    # test_generateCytosineReport.R:95-114 and 235-259
    for name, trim in (("capture.bam", 3), ("dragen-se-unsort-xg-xm.bam", 1)):
        a = cyt(H.bam(name, trim=trim), threshold_reads=False, report_context="CX")
        b = cyt(H.bam(name), threshold_reads=False, report_context="CX")
        ka = set(zip(a["rname"].tolist(), a["strand"].tolist(), a["pos"].tolist(), a["context"].tolist()))
        kb = set(zip(b["rname"].tolist(), b["strand"].tolist(), b["pos"].tolist(), b["context"].tolist()))
        assert ka <= kb
        assert not (a["pos"].size == b["pos"].size and all(np.array_equal(a[k], b[k]) for k in a))


def test_single_end_report():
    # test_generateCytosineReport.R:215-233
    b = H.bam("dragen-se-unsort-xg-xm.bam")
    cx = cyt(b, threshold_reads=False, report_context="CX")
    assert [cx["pos"].size, 6] == ev("dim(cx.single)")
    assert [int((cx["context"] == k).sum()) for k in (2, 6, 7)] == ev("as.numeric(table(cx.single$context)")
    assert [int(cx["meth"].sum()), int(cx["unmeth"].sum())] == ev("c(sum(cx.single$meth)")


def test_survey_probe_values():
    sp = H.expected()["survey_probe"]
    a = H.bam("amplicon010meth.bam")
    assert int(thr(a).sum()) == sp["amplicon010meth"]["pass"]
    r = cyt(a)
    assert [r["pos"].size, int(r["meth"].sum()), int(r["unmeth"].sum())] == sp["amplicon010meth"]["cg_thr"]
    assert [int((r["strand"] == 1).sum()), int((r["strand"] == 2).sum())] == sp["amplicon010meth"]["cg_thr_strand_rows"]
    r = cyt(a, threshold_reads=False)
    assert [r["pos"].size, int(r["meth"].sum()), int(r["unmeth"].sum())] == sp["amplicon010meth"]["cg_nothr"]
    r = cyt(a, threshold_reads=False, report_context="CX")
    assert [r["pos"].size, int(r["meth"].sum()), int(r["unmeth"].sum())] == sp["amplicon010meth"]["cx_nothr"]
    assert [int((r["context"] == k).sum()) for k in (2, 6, 7)] == sp["amplicon010meth"]["cx_ctx_rows"]
    # skip.duplicates=TRUE identical
    d = cyt(H.bam("amplicon010meth.bam", skip_duplicates=True))
    H.assert_reports_equal(d, cyt(a))
    for nm in ("amplicon100meth", "amplicon000meth"):
        b = H.bam(nm + ".bam")
        assert int(thr(b).sum()) == sp[nm]["pass"]
        r = cyt(b)
        assert [r["pos"].size, int(r["meth"].sum()), int(r["unmeth"].sum())] == sp[nm]["cg_thr"]
    b = H.bam("dragen-pe-namesort-xg-xm.bam")
    r = cyt(b, threshold_reads=False, report_context="CX")
    assert [r["pos"].size, int(r["meth"].sum()), int(r["unmeth"].sum())] == sp["dragen-pe-namesort-xg-xm"]["cx_nothr"]
    assert [int((r["context"] == k).sum()) for k in (2, 6, 7)] == sp["dragen-pe-namesort-xg-xm"]["cx_ctx_rows"]


def _mhl_sums(m):
    p, n = m["strand"] == 1, m["strand"] == 2
    return ([int(m["coverage"].sum()), int(m["coverage"][p].sum()), int(m["coverage"][n].sum())],
            [m["length"].sum(), m["lmhl"].sum()], [m["length"][p].sum(), m["lmhl"][p].sum()],
            [m["length"][n].sum(), m["lmhl"][n].sum()])


def _check_mhl_block(m, nth):
    S = "generateMhlReport"
    cov, tot, pl, mi = _mhl_sums(m)
    assert cov == H.expected_values(S, "c(sum(mhl.report$coverage), sum(mhl.report[strand", nth)
    # RUnit::checkEquals tolerance is 1.5e-8 relative; the literals carry 6-9 significant digits
    np.testing.assert_allclose(tot, H.expected_values(S, "c(sum(mhl.report$length), sum(mhl.report$lmhl))", nth), rtol=2e-7)
    np.testing.assert_allclose(pl, H.expected_values(S, 'c(sum(mhl.report[strand=="+"]$length)', nth), rtol=2e-7)
    np.testing.assert_allclose(mi, H.expected_values(S, 'c(sum(mhl.report[strand=="-"]$length)', nth), rtol=2e-7)
    assert H.group_sums_all_ctx(m, "pos") == H.expected_values(S, "mhl.report[, sum(as.numeric(pos))", nth)


def test_mhl_capture():
    # test_generateMhlReport.R:7-38
    b = H.bam("capture.bam")
    m1 = mhl(b, hmax=1)
    cg = cyt(b, threshold_reads=False)
    assert np.array_equal(m1["lmhl"], cg["meth"] / (cg["meth"] + cg["unmeth"]))      # identical()
    _check_mhl_block(mhl(b), 0)


def test_mhl_amplicon():
    # test_generateMhlReport.R:40-100
    a = H.bam("amplicon010meth.bam")
    _check_mhl_block(mhl(a, max_oo=1), 1)
    _check_mhl_block(mhl(a), 2)
    b = H.bam("amplicon100meth.bam", min_mapq=30, min_baseq=20)
    m = mhl(b, hmin=1, hmax=1, max_oo=1)
    cg = cyt(b, threshold_reads=False)
    assert m["lmhl"].size == cg["pos"].size
    beta = cg["meth"] / (cg["meth"] + cg["unmeth"])
    assert np.mean(np.abs(m["lmhl"] - beta)) / np.mean(np.abs(m["lmhl"])) < 0.022992   # RUnit tolerance semantics


def test_mhl_simulated_long_reads():
    # test_generateMhlReport.R:102-122: two 10000-base reads, XM ~ Z:z = 1:9, XG=CT, pos 1
    rng = np.random.default_rng(7)
    xms = ["".join(rng.choice(list("Zzzzzzzzzz"), 10000)) for _ in range(2)]
    t = H.templates_from_xm(xms, [1, 1], [1, 1])
    m = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 1, 0, 0.1)
    cg = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, "Z")
    assert [int(m["coverage"].sum()), m["length"].sum()] == H.expected_values("generateMhlReport", "c(sum(mhl.report$coverage), sum(mhl.report$length))")
    assert np.array_equal(m["lmhl"], cg["meth"] / (cg["meth"] + cg["unmeth"]))


def test_simulated_toys():
    # test_simulateBam.R:53-70: pos=1:6, XM recycled, XG recycled CT/AG
    xms = ["ZZZzzZZZ", "ZZzzzzZZ"] * 3
    t = H.templates_from_xm(xms, [1, 2, 3, 4, 5, 6], [1, 2] * 3)
    r = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, "Z")
    assert [r["pos"].size, 6] == H.expected_values("simulateBam", "dim(cg.beta)")
    assert [int(r["meth"].sum()), int(r["unmeth"].sum())] == H.expected_values("simulateBam", "c(sum(cg.beta$meth)")
    # test_simulateBam.R:72-87: 1000 reads at pos 1, first fully methylated, rest one Z in ten
    rng = np.random.default_rng(3)
    xms = ["Z" * 10] + ["".join(rng.permutation(list("Zzzzzzzzzz"))) for _ in range(999)]
    t = H.templates_from_xm(xms, [1] * 1000, [1] * 1000)
    r = cyt(t)
    assert [int(r["meth"].sum()), int(r["unmeth"].sum())] == H.expected_values("simulateBam", "c(sum(cg.vef$meth)")


def test_survey_toy_kats():
    # SURVEY.md 8c / Appendix A KATs
    t = H.templates_from_xm(["zZZZ"], [1], [1])
    m = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1)
    assert m["lmhl"].tolist() == [0, .5, .5, .5] and m["length"].tolist() == [4.0] * 4
    t = H.templates_from_xm(["ZZZzzZZZ", "ZZzzzzZZ"], [1, 2], [1, 2])
    r = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, "ZXH")
    assert list(zip(r["pos"].tolist(), r["strand"].tolist())) == \
        [(1, 1)] + [(p, s) for p in range(2, 9) for s in (1, 2)] + [(9, 2)]
    t = H.templates_from_xm(["Z.hZz", "Z.hZz", "..hZz"], [1, 1, 1], [1, 1, 1])
    m = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1)
    assert m["pos"].tolist() == [1, 4, 5] and m["coverage"].tolist() == [2, 3, 3]
    np.testing.assert_array_equal(m["length"], [4.0, 8 / 3, 8 / 3])
    np.testing.assert_array_equal(m["lmhl"], [8 / 24, 9 / 24, 0.0])
    m = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 1, 0, 0.1)
    np.testing.assert_array_equal(m["lmhl"], [2 / 3, 1.0, 0.0])
    # threshold-failing reads are lower-cased: reads Z.. / .z. / .zh all failing -> (pos 2, CG, 0, 2)
    t = H.templates_from_xm(["Z..", ".z.", ".zh"], [1, 1, 1], [1, 1, 1])
    r = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], np.zeros(3, np.int32), "Z")
    assert (r["pos"].tolist(), r["context"].tolist(), r["meth"].tolist(), r["unmeth"].tolist()) == ([2], [7], [0], [2])
    t = H.templates_from_xm(["Zz", "xz"], [1, 1], [1, 1])
    r = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, "Z")
    assert (r["pos"].tolist(), r["meth"].tolist(), r["unmeth"].tolist()) == ([2], [0], [2])


def test_bed_report_vef_pins_threshold():
    # test_generateBedReport.R:12-83 (amplicon reports): VEF per amplicon pins rcpp_threshold_reads
    S = "generateBedReport"
    for kw, vef_nth, tot_pre, minus_pre in (({}, 0, "sum(amplicon.report[", "sum(amplicon.report$`nreads-`)"),
                                            (dict(min_mapq=30, min_baseq=20), 0, "sum(quality.report[", "sum(quality.report$`nreads-`)")):
        b = H.bam("amplicon010meth.bam", **kw)
        bed = H.read_bed("amplicon.bed", b["levels"])
        match = H.match_amplicon(b, bed, 1)
        p = thr(b)
        groups = list(range(1, len(bed) + 1)) + [0]          # matched rows then NA
        vef = [p[match == g].sum() / (match == g).sum() for g in groups]
        want = H.expected_values(S, "quality.report$VEF" if kw else "amplicon.report$VEF")
        np.testing.assert_allclose(vef, want, rtol=1e-9)
        assert [int(b["strand"].size)] == H.expected_values(S, tot_pre)
        assert [int((b["strand"] == 2).sum())] == H.expected_values(S, minus_pre)


def test_bed_ecdf_pins_beta():
    # test_generateBedEcdf.R:8-26: ecdf(beta)(0.5) per amplicon, context then out-of-context
    want1 = [0.916666666667, 1, 0.885245901639, 1]
    want2 = [0.916666666667, 1, 0.885245901639, 1, 0.946236559140, 1, 0.892857142857, 1, 0.868131868132, 1]
    for kw, rows, want in (({}, [1, 2], want1), (dict(min_mapq=30, min_baseq=20), [1, 2, 3, 4, 0], want2)):
        b = H.bam("amplicon010meth.bam", **kw)
        bed = H.read_bed("amplicon.bed", b["levels"])
        match = H.match_amplicon(b, bed, 1)
        c = C2B["CG"]
        cb = orc.get_xm_beta(b["xm"], b["off"], c["ctx_meth"], c["ctx_unmeth"])
        ob = orc.get_xm_beta(b["xm"], b["off"], c["ooctx_meth"], c["ooctx_unmeth"])
        got = []
        for g in rows:
            got += [np.mean(cb[match == g] <= 0.5), np.mean(ob[match == g] <= 0.5)]
        np.testing.assert_allclose(got, want, atol=1e-8)


def test_oracle_reference_flush_semantics_unsorted():
    # unsorted rows: the restatement keeps the reference's flush-on-gap behaviour (duplicated positions)
    t = H.templates_from_xm(["zz", "zz"], [1, 1], [1, 1])
    t["start"] = np.asarray([10, 1], np.int32)
    r = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, "Z")
    assert r["pos"].tolist() == [1, 2, 10, 11]      # second read starts before max_pos: no flush, map keeps order
    t["start"] = np.asarray([1, 10, 1], np.int32)
    t2 = H.templates_from_xm(["zz", "zz", "zz"], [1, 1, 1], [1, 1, 1])
    t2["start"] = np.asarray([1, 10, 1], np.int32)
    r = orc.cx_report(t2["xm"], t2["off"], t2["rname"], t2["strand"], t2["start"], None, "Z")
    assert r["pos"].tolist() == [1, 2, 1, 2, 10, 11]   # flush at the gap, then pos 1-2 appear again
