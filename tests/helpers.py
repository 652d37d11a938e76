"""Shared test helpers: fixture loading, simulateBam-equivalent template
construction, report comparison.  Test infrastructure only."""
import functools
import json
import os

import numpy as np

from oracle import bamio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# R/internal.R:54-65 (.context.to.bases) -- kept separately from the product's
# copy (epialleler_amd.api.CONTEXT_TO_BASES); test_host_api checks they agree.
CONTEXT_TO_BASES = {
    "CG": dict(ctx_meth="Z", ctx_unmeth="z", ooctx_meth="XH", ooctx_unmeth="xh"),
    "CHG": dict(ctx_meth="X", ctx_unmeth="x", ooctx_meth="ZH", ooctx_unmeth="zh"),
    "CHH": dict(ctx_meth="H", ctx_unmeth="h", ooctx_meth="ZX", ooctx_unmeth="zx"),
    "CxG": dict(ctx_meth="ZX", ctx_unmeth="zx", ooctx_meth="H", ooctx_unmeth="h"),
    "CX": dict(ctx_meth="ZXH", ctx_unmeth="zxh", ooctx_meth="", ooctx_unmeth=""),
}


@functools.lru_cache(maxsize=None)
def expected():
    with open(os.path.join(GOLDEN, "expected.json")) as f:
        return json.load(f)


def expected_values(section, expr_prefix, nth=0):
    """n-th known-answer value in `section` whose R expression starts with `expr_prefix`."""
    hits = [b["value"] for b in expected()[section] if b["expr"].startswith(expr_prefix)]
    return hits[nth]


@functools.lru_cache(maxsize=None)
def load_bam(name, **kw):
    """preprocessBam() on a reference BAM fixture (cached).  kw as hashable items."""
    return bamio.preprocess_bam(os.path.join(GOLDEN, "bam", name), **dict(kw))


def bam(name, **kw):
    return load_bam(name, **{k: v for k, v in sorted(kw.items())})


def ctx_to_idx(ch):
    return ((ord(ch) + 2) >> 2) & 15


def templates_from_xm(xm_strings, starts, strands, rnames=None, seq_code=1):
    """Packed templates a single-end simulateBam() BAM would yield
    (flag 0, cigar <n>M, qual 'F', see R/internal.R:296-398 defaults):
    one byte per base, (nt16<<4)|ctx_to_idx(XM char); rows sorted by (rname,start), stable."""
    n = len(xm_strings)
    rnames = [1] * n if rnames is None else list(rnames)
    order = sorted(range(n), key=lambda i: (rnames[i], starts[i], i))
    chunks = [np.frombuffer(xm_strings[i].encode("latin1"), np.uint8) for i in order]
    xm = (np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)).astype(np.int64)
    packed = ((seq_code << 4) | (((xm + 2) >> 2) & 15)).astype(np.uint8)
    off = np.zeros(n + 1, np.int64)
    np.cumsum([c.size for c in chunks], out=off[1:])
    return {"xm": packed, "off": off,
            "rname": np.asarray([rnames[i] for i in order], np.int32),
            "strand": np.asarray([strands[i] for i in order], np.int32),
            "start": np.asarray([starts[i] for i in order], np.int32)}


def group_sums(rep, value, ctx_code=None):
    """sum(value) by (rname,strand) in (rname,strand) order, like the R tests'
    `[, sum(v), by=.(rname,strand,context)][order(rname,strand,context)]` for one context."""
    m = np.ones(rep["pos"].size, bool) if ctx_code is None else rep["context"] == ctx_code
    key = rep["rname"][m].astype(np.int64) * 4 + rep["strand"][m]
    v = (rep[value][m] if isinstance(value, str) else value[m]).astype(np.float64)
    out = []
    for k in np.unique(key):
        out.append(v[key == k].sum())
    return out


def group_sums_all_ctx(rep, value):
    """by=.(rname,strand,context) ordered by (rname,strand,context factor level = code)."""
    key = (rep["rname"].astype(np.int64) * 4 + rep["strand"]) * 16 + rep["context"]
    v = rep[value].astype(np.float64)
    return [v[key == k].sum() for k in np.unique(key)]


def match_amplicon(b, bed_rows, tolerance=1):
    """src/rcpp_match_target.cpp:16-45; bed_rows = [(rname_idx, start, end)], returns 1-based index or 0 for NA."""
    lens = np.diff(b["off"])
    res = np.zeros(b["start"].size, np.int64)
    for x in range(res.size):
        rs = int(b["start"][x])
        re_ = rs + int(lens[x]) - 1
        for i, (c, s, e) in enumerate(bed_rows):
            if b["rname"][x] == c and (abs(rs - s) <= tolerance or abs(re_ - e) <= tolerance):
                res[x] = i + 1
                break
    return res


def read_bed(name, levels):
    rows = []
    with open(os.path.join(GOLDEN, "bam", name)) as f:
        for ln in f:
            p = ln.split()
            if not p or p[0] in ("chr", "#chr") or not p[1].isdigit():
                continue
            rows.append((levels.index(p[0]) + 1, int(p[1]), int(p[2])))
    return rows


def assert_reports_equal(a, b, float_cols=()):
    assert set(a.keys()) == set(b.keys())
    for k in a:
        assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
        if k in float_cols:
            # bit-exact including NaN positions
            assert np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64)) or \
                np.array_equal(a[k], b[k], equal_nan=True), k
        else:
            assert np.array_equal(a[k], b[k]), k


# ---- a minimal BAM writer for the tests (what the reference's simulateBam() gives its long-read tests) ----------

def write_bam(path, records, refs=(("chrS", 1000),)):
    """records: dicts with seq (str), flag, pos (1-based), optional qname, mapq, cigar [(op, len)], qual (bytes or int),
    tid, tags {name: str (Z) | list of ints (B:C)}.  Defaults follow R/internal.R:296-398: mapq 60, cigar <len>M,
    quality 'F'."""
    import struct
    import zlib
    nt16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    body = bytearray(b"BAM\1")
    text = "@HD\tVN:1.0\tSO:unknown\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    body += struct.pack("<i", len(text)) + text.encode()
    body += struct.pack("<i", len(refs))
    for name, ln in refs:
        body += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    for k, r in enumerate(records):
        seq = r["seq"]
        n = len(seq)
        qname = (r.get("qname") or ("q%06d" % k)).encode() + b"\0"
        cigar = r.get("cigar") or [(0, n)]
        qual = r.get("qual", 37)
        qual = bytes([qual]) * n if isinstance(qual, int) else bytes(qual)
        packed = bytearray((n + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i >> 1] |= nt16[ch] << (4 if (i & 1) == 0 else 0)
        aux = bytearray()
        for tag, val in (r.get("tags") or {}).items():
            if isinstance(val, str):
                aux += tag.encode() + b"Z" + val.encode() + b"\0"
            else:
                aux += tag.encode() + b"BC" + struct.pack("<i", len(val)) + bytes(val)
        core = struct.pack("<iiBBHHHiiii", r.get("tid", 0), r["pos"] - 1, len(qname), r.get("mapq", 60), 4680, len(cigar),
                           r.get("flag", 0), n, -1, -1, 0)
        rec = core + qname + b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in cigar) + bytes(packed) + qual + bytes(aux)
        body += struct.pack("<i", len(rec)) + rec

    def block(data):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(bytes(data)) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(bytes(data)) & 0xFFFFFFFF, len(data)))

    with open(path, "wb") as f:
        for i in range(0, len(body), 60000):
            f.write(block(body[i:i + 60000]))
        f.write(block(b""))
    return path


def dirty_allocator(bam):
    """Leaves a freed block of non-zero int32s of the batch's row count in torch's caching allocator, so that the next
    torch.empty of that size (the `pass` column of cytosine_report_fused) starts out as garbage, not as zeros."""
    import torch
    junk = torch.full((max(bam.n, 1),), 0x5A5A5A5A, dtype=torch.int32, device="cuda:%d" % (bam.device or 0))
    del junk
