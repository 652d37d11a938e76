"""CPU-only checks of the host layer and the C-ABI library: the context table,
argument handling, that libepihip.so loads and exports every symbol declared in
include/epihip.h, and that the product path fails loudly without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from epialleler_amd import _lib
    _lib.build()                      # hipcc cross-compiles for gfx950 without a GPU
    return _lib.load()


def test_context_table_matches_reference():
    import epialleler_amd as ea
    assert ea.CONTEXT_TO_BASES == H.CONTEXT_TO_BASES
    assert ea.CONTEXT_LEVELS[1] == "CHH" and ea.CONTEXT_LEVELS[5] == "CHG" and ea.CONTEXT_LEVELS[6] == "CG"


def test_library_exports_every_declared_symbol(lib):
    from epialleler_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "epihip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(epi_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (epi_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.epi_version() >= 100 and lib.epi_tile_positions() == 2048 and lib.epi_cx_tile_positions(b"ZXH") == 1024


def test_no_oracle_in_product_path():
    # the product must never import or link the CPU oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "epialleler_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower().replace("no cpu", ""), os.path.join(dirpath, f)


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import epialleler_amd as ea
    t = H.templates_from_xm(["Zz"], [1], [1])
    bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
    with pytest.raises(ea.EpihipError) as ei:
        ea.rcpp_cx_report(bam, None, "Z")
    assert ei.value.code == 5 and "no CPU fallback" in str(ei.value)
    # the host-pointer entry points fail the same way
    import ctypes as C
    out = np.zeros(1, np.int32)
    rc = lib.epi_threshold_reads(C.c_void_p(t["xm"].ctypes.data), C.c_void_p(t["off"].ctypes.data), 1,
                                 b"Z", b"z", b"", b"", 0, 0.0, 1.0, C.c_void_p(out.ctypes.data))
    assert rc == 5


def test_processed_bam_validation():
    import epialleler_amd as ea
    t = H.templates_from_xm(["Zz", "zz"], [1, 2], [1, 2])
    with pytest.raises(ValueError):
        ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"][:1], t["strand"], t["start"])
    with pytest.raises(ValueError):
        ea.ProcessedBam.from_arrays(t["xm"][:-1], t["off"], t["rname"], t["strand"], t["start"])
    b = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"], levels=["chr1"])
    assert b.n == 2 and b.nbytes == 4 and ea.preprocessBam(b) is b
    with pytest.raises(ValueError):
        ea.preprocessBam("some.bam")                      # "Unable to open BAM file for reading"


def test_write_report_tsv(tmp_path):
    import epialleler_amd as ea
    rep = ea.Report({"rname": np.asarray([1, 1], np.int32), "strand": np.asarray([1, 2], np.int32),
                     "pos": np.asarray([10, 11], np.int32), "context": np.asarray([7, 2], np.int32),
                     "meth": np.asarray([3, 0], np.int32), "unmeth": np.asarray([1, 5], np.int32)}, ["chr1"])
    p = tmp_path / "r.tsv"
    ea.writeReport(rep, str(p))
    assert p.read_text() == "rname\tstrand\tpos\tcontext\tmeth\tunmeth\nchr1\t+\t10\tCG\t3\t1\nchr1\t-\t11\tCHH\t0\t5\n"
    import gzip
    ea.writeReport(rep, str(tmp_path / "r.tsv.gz"), gzip=True)
    assert gzip.open(tmp_path / "r.tsv.gz", "rt").read() == p.read_text()


def test_write_report_na_and_many_rows(tmp_path):
    """fwrite conventions: NA_integer_, factor codes outside the levels and NaN are empty fields (na = ""); many rows
    (several formatting chunks, several threads) come out in order; gzip output is a valid multi-member file."""
    import gzip
    import epialleler_amd as ea
    n = 200_003
    rng = np.random.default_rng(5)
    cov = rng.integers(0, 50, n).astype(np.int32)
    cov[7] = -2 ** 31
    lm = rng.random(n)
    lm[3] = np.nan
    lm[4] = 0.5
    lm[5] = 2.0
    lm[6] = np.inf
    ctx = rng.choice(np.asarray([2, 6, 7], np.int32), n)
    ctx[9] = 5
    rep = ea.Report({"rname": np.ones(n, np.int32), "strand": (1 + (np.arange(n) & 1)).astype(np.int32),
                     "pos": np.arange(n, dtype=np.int32) - 5, "context": ctx, "coverage": cov, "lmhl": lm}, ["chrA"])
    p = tmp_path / "big.tsv"
    ea.writeReport(rep, str(p), nthreads=5)
    lines = p.read_text().split("\n")
    assert lines[0] == "rname\tstrand\tpos\tcontext\tcoverage\tlmhl" and len(lines) == n + 2 and lines[-1] == ""
    want = lambda i: "\t".join(["chrA", "+-"[i & 1], str(i - 5), {2: "CHH", 5: "NA5", 6: "CHG", 7: "CG"}[int(ctx[i])],
                                 "" if cov[i] == -2 ** 31 else str(int(cov[i])), "" if np.isnan(lm[i]) else ("Inf" if np.isinf(lm[i]) else "%.15g" % lm[i])])
    for i in list(range(12)) + [65535, 65536, 65537, 131071, 131072, n - 1]:
        assert lines[1 + i] == want(i), i
    assert lines[1 + 3].endswith("\t") and lines[1 + 4].endswith("\t0.5") and lines[1 + 5].endswith("\t2")
    ea.writeReport(rep, str(tmp_path / "big.tsv.gz"), gzip=True, nthreads=3)
    assert gzip.open(tmp_path / "big.tsv.gz", "rt").read() == p.read_text()
    # a table with a text column takes the row-by-row path with the same conventions
    rep2 = ea.Report({"strand": np.asarray([1, 2], np.int32), "beta": np.asarray([np.nan, 0.25]),
                      "pattern": np.asarray(["00AB", None], object)})
    ea.writeReport(rep2, str(tmp_path / "pat.tsv"))
    assert (tmp_path / "pat.tsv").read_text() == "strand\tbeta\tpattern\n+\t\t00AB\n-\t0.25\t\n"


def test_write_report_double_format(tmp_path):
    """The double formatting of the native writer, pinned against Python's `%.15g` (what round 1's writer produced; the
    reference holds no written-file fixture, so data.table::fwrite's own text is parity unpinned): huge and tiny
    magnitudes, the 1e15 switch from integer to %.15g form, negative zero (prints as 0), a repeating fraction."""
    import epialleler_amd as ea
    vals = np.asarray([1e300, -1e300, 2.0 ** 63, -2.0 ** 63, 1e15, 1e15 - 1, -(1e15 - 1), 999999999999999.5, 1e-300, 1.0 / 3.0,
                       -0.0, 0.0, 123456789.125, 2.5e-7, 1e16, 4.0, -17.0], np.float64)
    rep = ea.Report({"pos": np.arange(vals.size, dtype=np.int32), "lmhl": vals})
    p = tmp_path / "f.tsv"
    ea.writeReport(rep, str(p), nthreads=2)
    lines = p.read_text().split("\n")[1:-1]
    assert len(lines) == vals.size
    for i, v in enumerate(vals):
        want = "0" if v == 0 else "%.15g" % v
        assert lines[i] == "%d\t%s" % (i, want), (i, lines[i], want)


def test_no_timing_switches_in_the_product_library():
    """Timing builds (phases skipped: wrong results by design) are a compile-time make target; the shipped library
    must not contain the environment switches of round 1."""
    from epialleler_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    for name in (b"EPIHIP_CX_ABLATE", b"EPIHIP_MHL_ABLATE", b"EPIHIP_CX_DIAG"):
        assert name not in data
