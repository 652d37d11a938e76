"""Host-side producer (epi_preprocess_bam: zlib BGZF reader + the reference's template packers, C++)
against the oracle-side restatement (oracle/bamio.py) on the reference's BAM fixtures, byte for byte,
plus the good/bad-file behaviour pinned by inst/unitTests/test_preprocessBam.R.  CPU only."""
import os

import numpy as np
import pytest

import helpers as H

BAM = os.path.join(H.GOLDEN, "bam")


@pytest.fixture(scope="module")
def ea():
    from epialleler_amd import _lib
    _lib.build()
    import epialleler_amd
    return epialleler_amd


@pytest.mark.parametrize("name,kw", [
    ("capture.bam", {}),
    ("capture.bam", dict(min_mapq=30, min_baseq=20)),
    ("capture.bam", dict(trim=3, nthreads=4)),
    ("amplicon010meth.bam", dict(skip_duplicates=True)),
    ("amplicon000meth.bam", dict(min_baseq=5, min_mapq=5)),
    ("amplicon100meth.bam", dict(trim=(2, 5))),
    ("dragen-se-unsort-xg-xm.bam", dict(trim=1)),
    ("dragen-se-unsort-xg-xm.bam", dict(min_baseq=30)),
    ("dragen-pe-namesort-xg-xm.bam", {}),
    ("capture.bam", dict(nthreads=7)),                   # paired-end ranges packed by several threads
    ("amplicon010meth.bam", dict(nthreads=3)),
    ("dragen-se-unsort-xg-xm.bam", dict(nthreads=5)),
])
def test_packer_matches_oracle(ea, name, kw):
    b = ea.preprocessBam(os.path.join(BAM, name), **kw)
    okw = {k: v for k, v in kw.items() if k != "nthreads"}
    o = H.bam(name, **okw)
    for k in ("xm", "off", "rname", "strand", "start"):
        assert np.array_equal(b.host[k], o[k]), k
    assert list(b.levels) == list(o["levels"]) and b.nrecs == o["nrecs"] and b.npushed == o["npushed"]
    # sorted by (rname, start), stable: what setorder(rname, start) leaves (R/internal.R:195)
    key = b.host["rname"].astype(np.int64) * (1 << 32) + b.host["start"]
    assert np.all(np.diff(key) >= 0)


def test_dims_pinned_by_reference_tests(ea):
    # test_preprocessBam.R:11-15, 32-36, 38-42: dim == c(2968,4) / c(500,4)
    assert ea.preprocessBam(os.path.join(BAM, "capture.bam")).n == 2968
    assert ea.preprocessBam(os.path.join(BAM, "amplicon010meth.bam"), skip_duplicates=True).n == 500
    q = ea.preprocessBam(os.path.join(BAM, "capture.bam"), min_mapq=30, min_baseq=20, nthreads=0)
    c = ea.preprocessBam(os.path.join(BAM, "capture.bam"))
    assert q.n == 2968 and not np.array_equal(q.host["xm"], c.host["xm"])
    for k in ("rname", "strand", "start"):
        assert np.array_equal(q.host[k], c.host[k])
    assert ea.preprocessBam(c) is c                       # already preprocessed: returned untouched (:20-23)
    assert c.paired and not ea.preprocessBam(os.path.join(BAM, "dragen-se-unsort-xg-xm.bam"), skip_duplicates=True).paired


@pytest.mark.parametrize("name,kw,needle", [
    ("empty.bam", {}, "Empty file"),                                        # :55-57
    ("dragen-pe-namesort-xg.bam", {}, "No XM tags"),                        # :70-72
    ("dragen-pe-unsort-xg-xm.bam", {}, "not sorted by name"),               # :75-77
    ("dragen-se-unsort-xg.bam", {}, "No XM tags"),                          # :124-126
    ("bwameth-se-unsort-yd.bam", {}, "YD tags"),                            # :114-116
    ("bsmap-se-unsort-zs.bam", {}, "ZS tags"),                              # :119-121
    ("dragen-pe-namesort-xg-xm.bam", dict(paired=False), "endness"),        # :129-131
    ("dragen-se-unsort-xg-xm.bam", dict(paired=True), "endness"),           # :134-136
    ("no-such-file.bam", {}, "Unable to open"),
])
def test_bad_files_raise(ea, name, kw, needle):
    with pytest.raises(ValueError) as ei:
        ea.preprocessBam(os.path.join(BAM, name), **kw)
    assert needle in str(ei.value)
