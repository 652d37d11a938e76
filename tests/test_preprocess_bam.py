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


# ---- bounded windows: the file is inflated and packed piece by piece; every seam must be invisible --------------------

@pytest.mark.parametrize("name,kw", [
    ("capture.bam", {}), ("capture.bam", dict(nthreads=3, trim=2)), ("amplicon010meth.bam", {}),
    ("dragen-se-unsort-xg-xm.bam", {}), ("dragen-pe-namesort-xg-xm.bam", {}),
])
@pytest.mark.parametrize("window_kib", [1, 7, 64])
def test_windowed_packer_equals_one_pass(ea, name, kw, window_kib):
    """window_kib = 1: every window is a single BGZF block, so records and mate pairs straddle almost every seam."""
    whole = ea.preprocessBam(os.path.join(BAM, name), **kw)
    piecewise = ea.preprocessBam(os.path.join(BAM, name), window_kib=window_kib, **kw)
    for k in ("xm", "off", "rname", "strand", "start"):
        assert np.array_equal(whole.host[k], piecewise.host[k]), k
    assert whole.nrecs == piecewise.nrecs and whole.levels == piecewise.levels


def test_windowed_synthetic_paired_file(ea, tmp_path):
    from epialleler_amd import synth
    path, nrec = synth.write_bam_paired(str(tmp_path / "pe.bam"), 30000, threads=2)
    a = ea.preprocessBam(path, nthreads=4)
    b = ea.preprocessBam(path, nthreads=3, window_kib=200)
    assert a.n == 30000 and a.nrecs == nrec == b.nrecs and a.paired
    for k in ("xm", "off", "rname", "strand", "start"):
        assert np.array_equal(a.host[k], b.host[k]), k


# ---- malformed input: an error message, never an out-of-bounds access ----------------------------------------------------

def _raw_bam(path, records, refs=(("chrS", 100000),), text="@HD\tVN:1.0\tSO:unknown\n"):
    import struct
    import zlib
    body = bytearray(b"BAM\1") + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for nm, ln in refs:
        body += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    for r in records:                                       # bytes, or (declared block_size, bytes)
        body += struct.pack("<i", r[0]) + r[1] if isinstance(r, tuple) else struct.pack("<i", len(r)) + r

    def block(data):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(bytes(data)) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(bytes(data)) & 0xFFFFFFFF, len(data)))
    with open(path, "wb") as f:
        for i in range(0, len(body), 60000):
            f.write(block(body[i:i + 60000]))
        f.write(block(b""))
    return str(path)


def _rec(qname=b"r1\0", flag=0, pos=10, l_seq=8, cigar=((0, 8),), xm=b"z..Z.h.x", xg=b"CT", tid=0, mpos=-1, tlen=0, mapq=60,
         l_seq_field=None, seq_bytes=None):
    import struct
    l_seq_field = l_seq if l_seq_field is None else l_seq_field
    core = struct.pack("<iiBBHHHiiii", tid, pos, len(qname), mapq, 4680, len(cigar), flag, l_seq_field, tid if mpos >= 0 else -1, mpos, tlen)
    cg = b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in cigar)
    seq = bytes([0x12] * ((l_seq + 1) // 2)) if seq_bytes is None else seq_bytes
    aux = b"XMZ" + xm + b"\0" + b"XGZ" + xg + b"\0"
    return core + qname + cg + seq + bytes([40] * l_seq) + aux


def test_well_formed_crafted_records(ea, tmp_path):
    ok = ea.preprocessBam(_raw_bam(tmp_path / "ok.bam", [_rec(qname=b"a\0"), _rec(qname=b"b\0", pos=30)]))
    assert ok.n == 2 and ok.nbytes == 16
    # paired-end mates whose CIGARs end in a deletion: the template is as wide as the last reference position reached
    pe = [_rec(qname=b"p\0", flag=99, pos=100, mpos=104, tlen=8, cigar=((0, 8), (2, 5))),
          _rec(qname=b"p\0", flag=147, pos=104, mpos=100, tlen=-8, cigar=((0, 8), (2, 7)))] * 1
    b = ea.preprocessBam(_raw_bam(tmp_path / "pe.bam", pe))
    assert b.n == 1 and b.nbytes == 4 + 8 + 7 and np.all(b.host["xm"][12:] == 0xFB)


@pytest.mark.parametrize("bad,needle", [
    (dict(l_seq_field=-5), "corrupt BAM record"),
    (dict(qname=b"xy"), "corrupt BAM record"),                                   # no NUL at the end of the name
    (dict(cigar=((0, 12),)), "CIGAR does not match"),                            # consumes more bases than stored
    (dict(cigar=((0, 5),)), "CIGAR does not match"),
    (dict(xm=b"z.."), "XM tag shorter"),
    (dict(tid=7), "reference id out of range"),
    (dict(l_seq_field=1 << 28), "corrupt BAM record"),                           # sizes beyond the block
])
def test_malformed_records_are_rejected(ea, tmp_path, bad, needle):
    recs = [_rec(qname=b"g%d\0" % i, pos=10 + i) for i in range(3)] + [_rec(**bad)]
    with pytest.raises(ValueError) as ei:
        ea.preprocessBam(_raw_bam(tmp_path / "bad.bam", recs), window_kib=1)
    assert needle in str(ei.value)


def test_malformed_pairs_and_truncation(ea, tmp_path):
    pe = [_rec(qname=b"p\0", flag=99, pos=100, mpos=108, tlen=16), _rec(qname=b"p\0", flag=147, pos=50, mpos=100, tlen=-16)]
    with pytest.raises(ValueError) as ei:
        ea.preprocessBam(_raw_bam(tmp_path / "pe.bam", pe))
    assert "starts before its template" in str(ei.value)
    good = _raw_bam(tmp_path / "good.bam", [_rec(qname=b"g%d\0" % i, pos=10 + i) for i in range(50)])
    data = open(good, "rb").read()
    for cut in (len(data) - 40, len(data) // 2, 30):
        p = tmp_path / ("cut%d.bam" % cut)
        p.write_bytes(data[:cut])
        with pytest.raises(ValueError):
            ea.preprocessBam(str(p))
    # a record whose block_size runs past the end of the stream
    with pytest.raises(ValueError) as ei:
        ea.preprocessBam(_raw_bam(tmp_path / "long.bam", [_rec(), (len(_rec()) + 100, _rec())]))
    assert "truncated BAM record" in str(ei.value)


def test_threads_give_the_same_table_on_a_generated_bam(tmp_path):
    """150 k templates (enough for the threaded range sort + pairwise merges and several packing segments): one thread
    and eight threads, whole file and 4 MiB windows, must produce the same sorted table byte for byte."""
    import epialleler_amd as ea
    from epialleler_amd import synth
    path = str(tmp_path / "gen.bam")
    synth.write_bam_paired(path, 150000)
    ref = None
    for nt, win in ((1, 0), (8, 0), (8, 4096)):
        bam = ea.preprocessBam(path, nthreads=nt, window_kib=win)
        h = bam.host
        key = (h["rname"].astype(np.int64) << 32) | h["start"]
        assert np.all(np.diff(key) >= 0)
        cur = {k: np.array(h[k]) for k in ("xm", "off", "rname", "strand", "start")}
        if ref is None:
            ref = cur
            assert bam.n == 150000 and int(cur["off"][-1]) == 150000 * 300
        else:
            for k in ref:
                assert np.array_equal(ref[k], cur[k]), (k, nt, win)


def test_host_columns_outlive_the_processed_bam(ea):
    """The numpy columns of preprocessBam() own the producer's buffers: keeping `bam.host["xm"]`
    after the ProcessedBam is collected must not read freed (or hipHostFree'd) memory."""
    import gc
    path = os.path.join(BAM, "capture.bam")
    want = ea.preprocessBam(path)
    ref = {k: np.array(v) for k, v in want.host.items()}
    xm = ea.preprocessBam(path).host["xm"]               # the ProcessedBam is garbage right away
    cols = ea.preprocessBam(path).host
    gc.collect()
    junk = [ea.preprocessBam(path) for _ in range(3)]    # allocator churn over whatever was freed
    del junk
    gc.collect()
    assert np.array_equal(xm, ref["xm"])
    for k in ref:
        assert np.array_equal(cols[k], ref[k]), k
    sl = xm[100:200]
    del xm
    gc.collect()
    assert np.array_equal(sl, ref["xm"][100:200])


def test_both_inflaters_give_the_same_table(ea, tmp_path):
    """BGZF blocks are inflated by libdeflate when its shared library can be loaded (it ships with the image) and by zlib
    otherwise (EPIHIP_NO_LIBDEFLATE forces zlib): same bytes either way, and a corrupt block is an error either way."""
    import subprocess
    import sys
    path = os.path.join(BAM, "capture.bam")
    code = ("import sys, hashlib; sys.path.insert(0, %r); import epialleler_amd as ea; b = ea.preprocessBam(%r, nthreads=3); "
            "print(hashlib.sha256(b''.join(bytes(memoryview(b.host[k])) for k in ('xm','off','rname','strand','start'))).hexdigest())"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path))
    outs = []
    for env in ({}, {"EPIHIP_NO_LIBDEFLATE": "1"}):
        e = dict(os.environ)
        e.update(env)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=e).decode().strip())
    assert outs[0] == outs[1] and len(outs[0]) == 64
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0xFF                                  # damage the compressed data of some block
    bad = tmp_path / "bad.bam"
    bad.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        ea.preprocessBam(str(bad))


@pytest.mark.parametrize("blk", [777, 5000, 65280])
def test_records_straddling_bgzf_blocks(ea, tmp_path, blk):
    """HTSlib keeps a record inside one BGZF block, and the index takes such blocks as a whole (each walked by the thread
    that inflated it).  A file whose blocks cut records anywhere -- legal BGZF -- must give the same table: the index then
    follows the chain record by record until it meets a block boundary again."""
    import gzip
    import struct
    import zlib
    path = os.path.join(BAM, "capture.bam")
    raw = gzip.open(path, "rb").read()

    def block(data):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    out = tmp_path / "reblocked.bam"
    with open(out, "wb") as f:
        for o in range(0, len(raw), blk):
            f.write(block(raw[o:o + blk]))
        f.write(block(b""))
    want = ea.preprocessBam(path).host
    for nt, win in ((1, 0), (5, 0), (3, 64)):
        got = ea.preprocessBam(str(out), nthreads=nt, window_kib=win).host
        for k in want:
            assert np.array_equal(got[k], want[k]), (k, blk, nt, win)
