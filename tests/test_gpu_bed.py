"""generateBedReport / generateAmpliconReport / generateCaptureReport / generateBedEcdf through the GPU
path, against the known-answer values of the reference's tests
(inst/unitTests/test_generateBedReport.R, test_generateBedEcdf.R) -- they pin rcpp_threshold_reads,
rcpp_get_xm_beta and rcpp_match_amplicon/_capture."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
BAM = os.path.join(H.GOLDEN, "bam")
S = "generateBedReport"


def test_amplicon_and_capture_reports():
    import epialleler_amd as ea
    amp_bam, amp_bed = os.path.join(BAM, "amplicon010meth.bam"), os.path.join(BAM, "amplicon.bed")
    rep = ea.generateAmpliconReport(amp_bam, amp_bed)
    assert [rep.nrow, len(rep)] == H.expected_values(S, "dim(amplicon.report)")                     # c(5, 9)
    assert [int(np.nansum(rep["nreads-"]))] == H.expected_values(S, "sum(amplicon.report$`nreads-`)")
    assert [int(np.nansum(rep["nreads+"]) + np.nansum(rep["nreads-"]))] == H.expected_values(S, "sum(amplicon.report[")
    np.testing.assert_allclose(rep["VEF"], H.expected_values(S, "amplicon.report$VEF"), rtol=1e-9)
    assert list(rep.keys()) == ["seqnames", "start", "end", "width", "strand", "amplicon", "nreads+", "nreads-", "VEF"]
    assert rep["seqnames"][-1] is None and rep["amplicon"][0] == "CpG00-13"
    nothr = ea.generateAmpliconReport(amp_bam, amp_bed, threshold_reads=False)
    assert [nothr.nrow, len(nothr)] == H.expected_values(S, "dim(nothreshold.report)") and np.all(np.isnan(nothr["VEF"]))
    q = ea.generateAmpliconReport(amp_bam, amp_bed, min_mapq=30, min_baseq=20)
    assert [int(np.nansum(q["nreads-"]))] == H.expected_values(S, "sum(quality.report$`nreads-`)")
    assert [int(np.nansum(q["nreads+"]) + np.nansum(q["nreads-"]))] == H.expected_values(S, "sum(quality.report[")
    np.testing.assert_allclose(q["VEF"], H.expected_values(S, "quality.report$VEF"), rtol=1e-9)
    assert np.sum(rep["VEF"][:4]) == np.sum(q["VEF"][:4]) and rep["VEF"][4] != q["VEF"][4]
    cap = ea.generateCaptureReport(os.path.join(BAM, "capture.bam"), os.path.join(BAM, "capture.bed"))
    assert [cap.nrow, len(cap)] == H.expected_values(S, "dim(capture.report)")                       # c(565, 9)
    assert [int(np.nansum(cap["nreads-"]))] == H.expected_values(S, "sum(capture.report$`nreads-`")
    assert [int(np.nansum(cap["nreads+"]) + np.nansum(cap["nreads-"]))] == H.expected_values(S, "sum(capture.report[")
    same = ea.generateBedReport(os.path.join(BAM, "capture.bam"), os.path.join(BAM, "capture.bed"), bed_type="capture")
    for k in cap:
        a, b = cap[k], same[k]
        assert np.array_equal(a, b) or (a.dtype.kind == "f" and np.array_equal(a, b, equal_nan=True)), k


def test_matching_equals_restatement():
    import epialleler_amd as ea
    from epialleler_amd import bed as B
    for name, bedf, typ in (("amplicon010meth.bam", "amplicon.bed", "amplicon"), ("capture.bam", "capture.bed", "capture")):
        t = H.bam(name)
        bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"], t["levels"])
        bd = ea.readBed(os.path.join(BAM, bedf))
        got = B._match_target(bam, bd, typ, 1, 1).cpu().numpy()
        rows = [(t["levels"].index(c) + 1, int(s), int(e)) for c, s, e in zip(bd.chrom, bd.start, bd.end)]
        lens = np.diff(t["off"])
        want = np.full(got.size, -2 ** 31, np.int64)
        for x in range(got.size):                                   # src/rcpp_match_target.cpp:30-44 / 63-76
            rs = int(t["start"][x]); re_ = rs + int(lens[x]) - 1
            for i, (c, s, e) in enumerate(rows):
                hit = t["rname"][x] == c and ((abs(rs - s) <= 1 or abs(re_ - e) <= 1) if typ == "amplicon"
                                              else (min(re_, e) - max(rs, s) + 1 >= 1))
                if hit:
                    want[x] = i + 1
                    break
        assert np.array_equal(got, want), name


def test_bed_ecdf():
    import epialleler_amd as ea
    amp_bam, amp_bed = os.path.join(BAM, "amplicon010meth.bam"), os.path.join(BAM, "amplicon.bed")
    e = ea.generateBedEcdf(amp_bam, amp_bed, bed_rows=[1, 2])
    vals = [f(0.5) for d in e.values() for f in (d["context"], d["out.of.context"])]
    np.testing.assert_allclose(vals, [0.916666666667, 1, 0.885245901639, 1], atol=1e-8)           # test_generateBedEcdf.R:8-12
    assert list(e.keys()) == ["chr17:43125624-43126026", "chr17:43125270-43125640"]
    e = ea.generateBedEcdf(amp_bam, amp_bed, bed_rows=None, min_mapq=30, min_baseq=20)
    vals = [f(0.5) for d in e.values() for f in (d["context"], d["out.of.context"])]
    np.testing.assert_allclose(vals, [0.916666666667, 1, 0.885245901639, 1, 0.946236559140, 1, 0.892857142857, 1,
                                      0.868131868132, 1], atol=1e-8)                                   # :21-26
    assert list(e.keys())[-1] is None


def test_bed_report_file_has_fwrite_na_fields(tmp_path):
    """generateBedReport(report.file=...): the file data.table::fwrite would write -- NA fields empty (the last row
    collects the reads that match no amplicon: NA seqnames/start/end/width/strand/name), counts without a fraction."""
    import epialleler_amd as ea
    amp_bam, amp_bed = os.path.join(BAM, "amplicon010meth.bam"), os.path.join(BAM, "amplicon.bed")
    out = tmp_path / "amp.tsv"
    assert ea.generateAmpliconReport(amp_bam, amp_bed, report_file=str(out)) is None
    rep = ea.generateAmpliconReport(amp_bam, amp_bed)
    lines = out.read_text().rstrip("\n").split("\n")
    assert lines[0].split("\t") == list(rep.keys()) and len(lines) == 1 + rep.nrow
    last = lines[-1].split("\t")
    assert last[:6] == [""] * 6 and last[6] == "%d" % rep["nreads+"][-1] and last[7] == "%d" % rep["nreads-"][-1]
    assert float(last[8]) == pytest.approx(rep["VEF"][-1], rel=1e-14)
    first = lines[1].split("\t")
    assert first[0] == rep["seqnames"][0] and first[1] == "%d" % rep["start"][0] and first[4] == "*" and "nan" not in out.read_text().lower()
    nothr = tmp_path / "nothr.tsv"
    ea.generateAmpliconReport(amp_bam, amp_bed, threshold_reads=False, report_file=str(nothr))
    assert all(l.split("\t")[8] == "" for l in nothr.read_text().rstrip("\n").split("\n")[1:])      # VEF is NA without thresholding


def test_placeholder_template_is_harmless():
    """A paired-end file without a usable pair gives the reference's placeholder template (strand code 0, no bases,
    src/rcpp_read_bam.cpp:155): reports are empty tables, the BED report counts it on neither strand."""
    import epialleler_amd as ea
    t = {"xm": np.zeros(0, np.uint8), "off": np.zeros(2, np.int64), "rname": np.ones(1, np.int32),
         "strand": np.zeros(1, np.int32), "start": np.ones(1, np.int32)}
    bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"], levels=["chr17"])
    assert ea.generateCytosineReport(bam).nrow == 0 and ea.generateMhlReport(bam).nrow == 0
    bed = ea.Bed(["chr17"], [1], [100])
    rep = ea.generateBedReport(bam, bed, bed_type="capture")
    assert rep.nrow == 1 and np.isnan(rep["nreads+"][0]) and np.isnan(rep["nreads-"][0])
    bam.close()
