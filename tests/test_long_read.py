"""Long-read (MM/ML) producer, SURVEY 8f row 4: rcpp_read_bam_mm_single (src/rcpp_read_bam.cpp:364-579).
The reference's own tests build their inputs with simulateBam(); here tests/helpers.write_bam() writes the same
records, and every expected table of test_generateCytosineReport.R:262-433 (tests/golden/expected.json "longRead")
is checked (a) through the oracle-side restatement + the CPU oracle and (b) for the C++ producer, byte for byte
against the restatement, also on random MM/ML inputs.  CPU only; the GPU end of it is in test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

import helpers as H
from oracle import bamio
from oracle import oracle as orc

CASES = json.load(open(os.path.join(H.GOLDEN, "expected.json")))["longRead"]


def case_bam(case, path):
    recs = []
    for k in range(max(len(case["flag"]), len(case["Mm"]))):
        recs.append(dict(seq=case["seq"][k % len(case["seq"])], flag=case["flag"][k % len(case["flag"])],
                         pos=case["pos"][k % len(case["pos"])],
                         tags={"Mm": case["Mm"][k % len(case["Mm"])], "Ml": case["Ml"][k % len(case["Ml"])]}))
    return H.write_bam(path, recs)


def eval_check(rep, expr):
    """The R expressions of the long-read section, on a report dict (strand 1/2, context 2/6/7)."""
    st = np.asarray(rep["strand"]); pos = np.asarray(rep["pos"]); ctx = np.asarray(rep["context"])
    me = np.asarray(rep["meth"]); un = np.asarray(rep["unmeth"])
    sname = lambda m: ["+" if v == 1 else "-" for v in st[m]]
    allrows = np.ones(st.size, bool)
    if expr == "cx.report[, .(strand, pos, context, meth, unmeth)]":
        return {"strand": sname(allrows), "pos": pos.tolist(), "context": ctx.tolist(), "meth": me.tolist(), "unmeth": un.tolist()}
    if expr == "cx.report[meth>0, .(strand, pos, context)]":
        m = me > 0
        return {"strand": sname(m), "pos": pos[m].tolist(), "context": ctx[m].tolist()}
    if expr == 'unname(unlist(cx.report[strand=="-", .(sum(meth), sum(unmeth))]))':
        return [int(me[st == 2].sum()), int(un[st == 2].sum())]
    if expr == 'cx.report[strand=="+" & meth>=1, .(pos, context)]':
        m = (st == 1) & (me >= 1)
        return {"pos": pos[m].tolist(), "context": ctx[m].tolist()}
    if expr == "dim(cx.report)":
        return [int(st.size), 6]
    if expr == 'cx.report[context=="CG", .(strand, pos, meth, unmeth)]':
        m = ctx == 7
        return {"strand": sname(m), "pos": pos[m].tolist(), "meth": me[m].tolist(), "unmeth": un[m].tolist()}
    raise AssertionError("unknown expression " + expr)


def expected_value(v):
    if isinstance(v, dict) and "strand" in v and len(v["strand"]) == 1 and len(v.get("pos", [])) > 1:
        v = dict(v, strand=v["strand"] * len(v["pos"]))          # factor("+") recycled by data.table
    return v


def check_case(case, preprocess, cx_report, tmp_path):
    path = case_bam(case, str(tmp_path / "lr.bam"))
    for r in case["reports"]:
        t = preprocess(path, min_prob=r["min_prob"], highest_prob=r["highest_prob"])
        rep = cx_report(t, H.CONTEXT_TO_BASES[r["report_context"]]["ctx_meth"])
        for c in r["checks"]:
            assert eval_check(rep, c["expr"]) == expected_value(c["value"]), (case["Mm"], r["min_prob"], c["expr"])


@pytest.mark.parametrize("k", range(len(CASES)))
def test_oracle_reproduces_reference_long_read_tables(k, tmp_path):
    check_case(CASES[k], lambda p, **kw: bamio.preprocess_bam(p, **kw),
               lambda t, ctx: orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, ctx), tmp_path)


@pytest.fixture(scope="module")
def ea():
    from epialleler_amd import _lib
    _lib.build()
    import epialleler_amd
    return epialleler_amd


def same_templates(b, o):
    for k in ("xm", "off", "rname", "strand", "start"):
        assert np.array_equal(b.host[k], o[k]), k
    assert b.npushed == o["npushed"] and b.nrecs == o["nrecs"]


@pytest.mark.parametrize("k", range(len(CASES)))
def test_producer_matches_restatement_on_reference_cases(ea, k, tmp_path):
    path = case_bam(CASES[k], str(tmp_path / "lr.bam"))
    for r in CASES[k]["reports"]:
        kw = dict(min_prob=r["min_prob"], highest_prob=r["highest_prob"])
        same_templates(ea.preprocessBam(path, **kw), bamio.preprocess_bam(path, **kw))


def random_long_read_records(rng, n):
    recs = []
    for _ in range(n):
        L = int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTACGTACGTNYRM"), L))
        flag = int(rng.choice([0, 16, 0, 16, 256, 4, 1024]))
        order = range(L - 1, -1, -1) if flag & 16 else range(L)
        comp = {"A": "T", "T": "A", "C": "G", "G": "C", "N": "N"}
        entries, ml = [], []
        for _e in range(int(rng.integers(0, 5))):
            base = str(rng.choice(list("CGATN")))
            target = comp[base] if flag & 16 else base
            navail = sum(1 for i in order if base == "N" or seq[i] == target)
            codes = str(rng.choice(["m", "h", "mh", "hm", "27551", "76792", "n", "a"]))
            ncodes = 1 if codes[0].isdigit() else len(codes)
            deltas, used = [], 0
            while used < navail and rng.random() < 0.8:
                d = int(rng.integers(0, 6))
                if used + d + 1 > navail:
                    break
                deltas.append(d)
                used += d + 1
            if rng.random() < 0.05:
                deltas.append(navail + 3)                                     # points beyond the sequence
            entries.append(base + str(rng.choice(["+", "-"])) + codes + str(rng.choice(["", "", ".", "?"])) +
                           "".join(",%d" % d for d in deltas) + ";")
            ml += [int(v) for v in rng.integers(0, 256, size=len(deltas) * ncodes)]
        tags = {}
        if entries or rng.random() < 0.5:
            tags[str(rng.choice(["MM", "Mm"]))] = "".join(entries)
            if rng.random() < 0.9:
                tags[str(rng.choice(["ML", "Ml"]))] = ml
        # a CIGAR with insertions, deletions and soft clips that consumes exactly L query bases
        cigar, left = [], L
        if left > 4 and rng.random() < 0.3:
            s = int(rng.integers(1, 4)); cigar.append((4, s)); left -= s
        while left > 0:
            m = int(rng.integers(1, left + 1)); cigar.append((0, m)); left -= m
            if left > 0 and rng.random() < 0.5:
                i = int(rng.integers(1, min(left, 5) + 1)); cigar.append((1, i)); left -= i
            if left > 0 and rng.random() < 0.5:
                cigar.append((int(rng.choice([2, 3])), int(rng.integers(1, 9))))
        if cigar[-1][0] in (2, 3):
            cigar.pop()
        recs.append(dict(seq=seq, flag=flag, pos=int(rng.integers(1, 5000)), tid=int(rng.integers(0, 2)),
                         mapq=int(rng.integers(0, 61)), qual=bytes(int(q) for q in rng.integers(0, 42, size=L)),
                         cigar=cigar, tags=tags))
    return recs


@pytest.mark.parametrize("seed,kw", [
    (1, {}), (2, dict(min_baseq=20, min_mapq=10)), (3, dict(min_prob=128)), (4, dict(min_prob=100, highest_prob=False)),
    (5, dict(trim=(3, 2), skip_duplicates=True)),
])
def test_producer_matches_restatement_on_random_long_reads(ea, tmp_path, seed, kw):
    rng = np.random.default_rng(seed)
    recs = random_long_read_records(rng, 300)
    recs[0]["tags"].setdefault("MM", "C+m;")                                   # the file must be recognised as long-read
    path = H.write_bam(str(tmp_path / "rnd.bam"), recs, refs=(("chr1", 100000), ("chr2", 100000)))
    b = ea.preprocessBam(path, **kw)
    same_templates(b, bamio.preprocess_bam(path, **kw))
    assert b.n > 100
