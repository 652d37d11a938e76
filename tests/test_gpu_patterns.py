"""extractPatterns on the GPU (epi_batch_extract_patterns through the C ABI): the reference's 43 expected values, full
equality with the oracle's table (hashes included), the equivalences at the end of test_extractPatterns.R and random
targets on random templates."""
import os

import numpy as np
import pytest

import helpers as H
import synth_np
import test_extract_patterns as TP
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
BAM = os.path.join(H.GOLDEN, "bam")


@pytest.fixture(scope="module")
def ea():
    import epialleler_amd
    return epialleler_amd


def gpu_table(ea, bam, bed, **kw):
    bed = bed if ":" in bed else os.path.join(BAM, bed)
    return TP.table_from_report(ea.extractPatterns(os.path.join(BAM, bam), bed, **kw))


def same_table(a, b):
    assert a["positions"] == b["positions"] and a["pattern"] == b["pattern"]
    for k in ("strand", "start", "end", "nbase", "cells"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(np.asarray(a["beta"], np.float64).view(np.uint64), np.asarray(b["beta"], np.float64).view(np.uint64))


@pytest.fixture(scope="module")
def gpu_tables(ea):
    return {n: gpu_table(ea, **kw) for n, kw in TP.CALLS.items()}


@pytest.mark.parametrize("k", range(len(TP.GOLD)))
def test_gpu_reproduces_reference_pattern_values(gpu_tables, k):
    assert TP.evaluate(gpu_tables, TP.GOLD[k]["expr"]) == TP.GOLD[k]["value"], TP.GOLD[k]["expr"]


def test_gpu_tables_equal_oracle_tables(gpu_tables):
    for n, kw in TP.CALLS.items():
        same_table(gpu_tables[n], TP.oracle_patterns(**kw))


def test_reference_equivalences(ea):
    # test_extractPatterns.R:262-307: duplicated / out-of-target highlight positions change nothing; a missing BED row
    # gives the empty table
    base = dict(bam="capture.bam", bed="chr17:61864583-61864585")
    same_table(gpu_table(ea, highlight_positions=[61864584, 61864584, 61864584], **base),
               gpu_table(ea, highlight_positions=[61864584, 61864586], **base))
    same_table(gpu_table(ea, **base), gpu_table(ea, bed_row=[1, 2, 3, 4, 5], highlight_positions=[1, 2, -61864584], **base))
    assert not ea.extractPatterns(os.path.join(BAM, "capture.bam"), "chr17:61864583-61864585", bed_row=2)


def test_random_targets(ea):
    rng = np.random.default_rng(99)
    for it in range(40):
        t = synth_np.random_templates(rng, int(rng.integers(1, 1500)), 0, int(rng.integers(1, 500)), int(rng.integers(1, 4)),
                                      int(rng.integers(50, 4000)), p_garbage=float(rng.choice([0, 0.1])))
        bam = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
        try:
            for _ in range(4):
                rn = int(rng.integers(1, 4))
                ts = int(rng.integers(1, 4000)); te = ts + int(rng.integers(0, 600))
                ctx = str(rng.choice(["Zz", "ZzXx", "HhXxZz", "Hh"]))
                clip = bool(rng.integers(0, 2)); ro = int(rng.integers(0, 3)); mo = int(rng.integers(1, 30))
                freq = float(rng.choice([0.0, 0.01, 0.2]))
                hl = sorted({int(p) for p in rng.integers(ts, te + 1, size=int(rng.integers(0, 4)))})
                rep = ea.rcpp_extract_patterns(bam, rn, ts, te, mo, ctx, freq, clip, ro, hl)
                o = orc.extract_patterns(t["xm"], t["off"], t["rname"], t["strand"], t["start"], rn, ts, te, mo, ctx, freq, clip, ro, hl)
                want = TP.table_from(o["strand"], o["start"], o["end"], o["nbase"], o["beta"], ["%016X" % int(v) for v in o["fnv"]],
                                     o["positions"], o["cells"])
                same_table(TP.table_from_report(rep), want)
        finally:
            bam.close()
