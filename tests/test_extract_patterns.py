"""extractPatterns / rcpp_extract_patterns: every numeric expectation of the reference's
inst/unitTests/test_extractPatterns.R (tests/golden/expected.json "extractPatterns", 43 values) checked against the
oracle restatement here, and against the GPU path in test_gpu_patterns.py."""
import json
import os

import numpy as np
import pytest

import helpers as H
from oracle import oracle as orc

GOLD = json.load(open(os.path.join(H.GOLDEN, "expected.json")))["extractPatterns"]
NA = -2 ** 31
LEVELS = ("NA1", "H", "A", "C", "NA5", "X", "Z", "NA8", "NA9", "h", "G", "T", "N", "x", "z", "NA16")
OFFSET = {"CG": 1, "CHG": 2, "CHH": 0, "CxG": 0, "CX": 0}

# the calls of test_extractPatterns.R (inputs only), by the name the R test gives the result
CALLS = {
    "noclip.patterns": dict(bam="amplicon010meth.bam", bed="amplicon.bed", bed_row=2),
    "clip.patterns": dict(bam="amplicon010meth.bam", bed="amplicon.bed", bed_row=2, clip_patterns=True),
    "exact.patterns": dict(bam="amplicon010meth.bam", bed="chr17:43124895-43126001", clip_patterns=True),
    "nooffset.patterns": dict(bam="amplicon010meth.bam", bed="chr17:43124895-43126001", clip_patterns=True, strand_offset=0),
    "cxg.patterns": dict(bam="amplicon010meth.bam", bed="chr17:43124895-43126001", extract_context="CxG", clip_patterns=True),
    "cx.patterns": dict(bam="amplicon010meth.bam", bed="chr17:43124895-43126001", extract_context="CX", clip_patterns=True),
    "capture.patterns": dict(bam="capture.bam", bed="chr20:57266125-57268185"),
    "snv.patterns": dict(bam="capture.bam", bed="chr17:61864583-61864585", highlight_positions=[61864584]),
}


def target_of(bed, bed_row, levels):
    if ":" in bed:
        chrom, rng = bed.rsplit(":", 1)
        a, b = rng.split("-")
        return levels.index(chrom) + 1, int(a), int(b)
    return H.read_bed(bed, levels)[bed_row - 1]


def oracle_patterns(bam, bed, bed_row=1, extract_context="CG", clip_patterns=False, strand_offset=None, highlight_positions=(),
                    match_min_overlap=1, min_context_freq=0.01):
    """What extractPatterns() does around rcpp_extract_patterns (R/extractPatterns.R:107-143, R/internal.R:683-714)."""
    t = H.bam(bam)
    rn, ts, te = target_of(bed, bed_row, list(t["levels"]))
    c = H.CONTEXT_TO_BASES[extract_context]
    off = OFFSET[extract_context] if strand_offset is None else strand_offset
    hl = sorted({p for p in highlight_positions if ts <= p <= te})
    o = orc.extract_patterns(t["xm"], t["off"], t["rname"], t["strand"], t["start"], rn, ts, te, match_min_overlap,
                             c["ctx_meth"] + c["ctx_unmeth"], min_context_freq, clip_patterns, off, hl)
    return table_from(o["strand"], o["start"], o["end"], o["nbase"], o["beta"], ["%016X" % int(v) for v in o["fnv"]],
                      o["positions"], o["cells"])


def table_from(strand, start, end, nbase, beta, pattern, positions, cells):
    return {"strand": np.asarray(strand), "start": np.asarray(start), "end": np.asarray(end), "nbase": np.asarray(nbase),
            "beta": np.asarray(beta), "pattern": list(pattern), "positions": [int(p) for p in positions],
            "cells": np.asarray(cells).reshape(len(positions), len(pattern))}


def table_from_report(rep):
    """A Report of epialleler_amd.rcpp_extract_patterns -> the same structure."""
    if not rep:
        return table_from([], [], [], [], [], [], [], np.zeros((0, 0), np.int32))
    pos = [k for k in rep if k.lstrip("-").isdigit()]
    return table_from(rep["strand"], rep["start"], rep["end"], rep["nbase"], rep["beta"], list(rep["pattern"]),
                      [int(k) for k in pos], np.stack([rep[k] for k in pos]) if pos else np.zeros((0, len(rep["strand"])), np.int32))


def evaluate(tables, expr):
    name = [n for n in tables if n in expr][0]
    t = tables[name]
    e = expr.replace(name, "X")
    npat = len(t["pattern"])
    colnames = ["seqnames", "strand", "start", "end", "nbase", "beta", "pattern"] + [str(p) for p in t["positions"]]
    hi = t["beta"] > 0.5
    if e == "dim(X)":
        return [npat, len(colnames)]
    if e == "length(unique(X$pattern))":
        return [len(set(t["pattern"]))]
    if e == "sum(X$nbase)":
        return [int(t["nbase"].sum())]
    if e == "X[beta>0.5, length(unique(pattern))]":
        return [len({p for p, h in zip(t["pattern"], hi) if h})]
    if e == "c(nrow(X[beta>0.5]), nrow(X))":
        return [int(hi.sum()), npat]
    if e.startswith("match(c("):
        names = [s for s in e[len("match(c("):e.index(")")].replace('"', "").replace(" ", "").split(",")]
        return [colnames.index(s) + 1 if s in colnames else None for s in names]
    if e.startswith("sum(X==") and e.endswith(", na.rm=TRUE)"):
        letter = e[len('sum(X=="')]
        return [int((t["cells"] == LEVELS.index(letter) + 1).sum())]
    if e.startswith("X[, .N, by=.(strand, `"):
        col = e[len("X[, .N, by=.(strand, `"):].split("`")[0]
        with_pattern = ", pattern)]" in e
        codes = t["cells"][t["positions"].index(int(col))]
        groups, order = {}, []
        for i in range(npat):
            k = (int(t["strand"][i]), int(codes[i])) + ((t["pattern"][i],) if with_pattern else ())
            if k not in groups:
                groups[k] = 0
                order.append(k)
            groups[k] += 1
        # [order(strand, col)]: stable, NA last (base::order semantics inside [.data.table)
        order.sort(key=lambda k: (k[0], k[1] == NA, k[1]))
        return [groups[k] for k in order]
    raise AssertionError("unknown expression: " + expr)


@pytest.fixture(scope="module")
def oracle_tables():
    return {n: oracle_patterns(**kw) for n, kw in CALLS.items()}


@pytest.mark.parametrize("k", range(len(GOLD)))
def test_oracle_reproduces_reference_pattern_values(oracle_tables, k):
    assert evaluate(oracle_tables, GOLD[k]["expr"]) == GOLD[k]["value"], GOLD[k]["expr"]


def test_every_reference_expectation_is_covered():
    assert len(GOLD) == 43 and all(any(n in g["expr"] for n in CALLS) for g in GOLD)
