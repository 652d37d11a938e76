"""Synthetic packed-template batches generated straight into HBM (bench / tests).

Model and hashing: epialleler_amd/csrc/synth.hip (numpy mirror: tests/synth_np.py).
torch only provides the device buffers the batch adopts.
"""
import ctypes as C

from . import _lib
from .api import ProcessedBam, _stream


def generate_device(n_total, read_len=300, n_chr=4, depth=30, seed=42, row_first=0, n=None,
                    gap_from=0, gap_len=0, device=None):
    """Rows [row_first, row_first+n) of the global sorted synthetic stream, resident on `device`."""
    import torch
    lib = _lib.load()
    if device is None:
        device = torch.cuda.current_device()
    n = n_total - row_first if n is None else n
    dev = "cuda:%d" % device
    nbytes = n * read_len
    cap = (nbytes + 15) // 16 * 16 + 64
    xm = torch.empty(cap, dtype=torch.uint8, device=dev)
    xm[nbytes:] = 0xFB
    off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    rname = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    strand = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    start = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    p = _lib.SynthParams(seed, n_total, row_first, n, read_len, n_chr, depth, gap_from, gap_len)
    with torch.cuda.device(device):
        _lib.check(lib.epi_synth_generate_dev(C.byref(p), C.c_void_p(xm.data_ptr()), C.c_void_p(off.data_ptr()),
                                              C.c_void_p(rname.data_ptr()), C.c_void_p(strand.data_ptr()),
                                              C.c_void_p(start.data_ptr()), _stream(device)))
    levels = tuple("chrS%d" % (i + 1) for i in range(n_chr))
    return ProcessedBam.from_device(xm, nbytes, off, rname, strand, start, levels)


# ---- SURVEY 8d-conformant variant: uniform-random starts, sorted; ragged lengths; gapped templates ---------------
_M64 = (1 << 64) - 1


def _s64(v):
    v &= _M64
    return v - (1 << 64) if v >> 63 else v


def _mix64_py(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _mix64_t(z):
    """splitmix64 finaliser on int64 tensors (two's-complement wrap-around = uint64 arithmetic)."""
    z = z + _s64(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _hash3_t(seed, stream, idx):
    """hash3() of csrc/synth.hip for an int64 tensor of indices."""
    s = _mix64_py((seed + stream * 0xD1B54A32D192ED03) & _M64)
    return _mix64_t(idx ^ _s64(s))


def uniform_layout(n_total, mean_len=300, n_chr=4, depth=30, seed=42, row_first=0, n=None, device="cpu", ragged=True,
                   pileup=None):
    """(rname, start, length) of rows [row_first, row_first+n) of the stream whose starts are uniform on each
    chromosome and then sorted (torch, any device).  Template lengths are uniform in [0.8, 1.2] x mean_len, or exactly
    mean_len with ragged=False (SURVEY 8d's literal config 2: uniform starts, L = 300).  pileup = (first global row,
    rows): those rows all start where the first of them does -- an amplicon-like pile-up inside a WGS-like stream
    (clipped to the chromosome of its first row; sortedness is kept, the rows behind them start further right)."""
    import torch
    n = n_total - row_first if n is None else n
    rpc = (n_total + n_chr - 1) // n_chr
    chr_len = max(rpc * mean_len // depth, 1)
    lo_len, hi_len = (mean_len * 4) // 5, (mean_len * 6) // 5
    rn, st = [], []
    c0, c1 = row_first // rpc, (row_first + n - 1) // rpc if n > 0 else row_first // rpc - 1
    for c in range(c0, c1 + 1):
        a, b = c * rpc, min((c + 1) * rpc, n_total)
        x = torch.arange(a, b, dtype=torch.int64, device=device)
        raw = 1 + _lsr(_hash3_t(seed, 1, x), 1) % chr_len
        srt = torch.sort(raw).values
        if pileup is not None and a <= int(pileup[0]) < b:
            p0, p1 = int(pileup[0]) - a, min(int(pileup[0]) + int(pileup[1]), b) - a
            srt[p0:p1] = srt[p0]
        lo, hi = max(a, row_first) - a, min(b, row_first + n) - a
        st.append(srt[lo:hi].to(torch.int32))
        rn.append(torch.full((hi - lo,), c + 1, dtype=torch.int32, device=device))
    x = torch.arange(row_first, row_first + n, dtype=torch.int64, device=device)
    lens = lo_len + _lsr(_hash3_t(seed, 6, x), 1) % (hi_len - lo_len + 1)
    if not ragged:
        lens = torch.full_like(lens, int(mean_len))
    cat = lambda parts, dt: torch.cat(parts) if parts else torch.empty(0, dtype=dt, device=device)
    return cat(rn, torch.int32), cat(st, torch.int32), lens


def generate_device_uniform(n_total, mean_len=300, n_chr=4, depth=30, seed=42, row_first=0, n=None,
                            gap_every=4, gap_len=50, device=None, ragged=True, pileup=None, tail=None):
    """Rows [row_first, row_first+n) of the uniform-start stream, resident on `device`.  bench: cfg2 / cfg3 / cfg4 / cfg5
    with ragged=False, gap_every=0 (uniform-random starts then sort, fixed L); cfg2u with the defaults
    (ragged lengths, a 0xFB gap between the mates of every fourth template)."""
    import torch
    lib = _lib.load()
    if device is None:
        device = torch.cuda.current_device()
    n = n_total - row_first if n is None else n
    dev = "cuda:%d" % device
    rname, start, lens = uniform_layout(n_total, mean_len, n_chr, depth, seed, row_first, n, dev, ragged, pileup)
    if tail is not None and n:                               # (every, bytes): a paired-end library's insert-size tail -- one template
        every, long_len = int(tail[0]), int(tail[1])         # in `every` (by global row id) is `bytes` long
        first = (-row_first) % every
        lens[torch.arange(first, n, every, device=dev)] = long_len
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=off[1:])
    nbytes = int(off[-1].item()) if n else 0
    cap = (nbytes + 15) // 16 * 16 + 64
    xm = torch.empty(cap, dtype=torch.uint8, device=dev)
    xm[nbytes:] = 0xFB
    strand = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    with torch.cuda.device(device):
        _lib.check(lib.epi_synth_fill_dev(seed, row_first, n, C.c_void_p(off.data_ptr()), C.c_void_p(rname.data_ptr()),
                                          C.c_void_p(start.data_ptr()), nbytes, gap_every, gap_len,
                                          C.c_void_p(xm.data_ptr()), C.c_void_p(strand.data_ptr()), _stream(device)))
    levels = tuple("chrS%d" % (i + 1) for i in range(n_chr))
    return ProcessedBam.from_device(xm, nbytes, off, rname.contiguous(), strand, start.contiguous(), levels)


# ---- a name-sorted paired-end XG/XM BAM on disk (file-to-file bench workload; numpy + zlib only) --------------------
def write_bam_paired(path, n_pairs, n_chr=4, read_len=150, depth=30, seed=42, threads=8, level=1):
    """2 * n_pairs records (mates 99/147 or 83/163, abutting: template = 2 * read_len bases), XM drawn with the
    context frequencies of the synthetic model, XG = CT / GA per template.  All records have the same size, so the
    whole uncompressed stream is one structured numpy array; BGZF blocks are deflated by a thread pool."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    L = int(read_len)
    rng = np.random.default_rng(seed)
    rec = np.dtype([("bs", "<i4"), ("refid", "<i4"), ("pos", "<i4"), ("lrn", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                    ("ncig", "<u2"), ("flag", "<u2"), ("lseq", "<i4"), ("mrefid", "<i4"), ("mpos", "<i4"), ("tlen", "<i4"),
                    ("qname", "u1", 10), ("cigar", "<u4"), ("seq", "u1", (L + 1) // 2), ("qual", "u1", L),
                    ("xmtag", "u1", 3), ("xm", "u1", L), ("xmnul", "u1"), ("xg", "u1", 6)])
    n = 2 * int(n_pairs)
    a = np.zeros(n, rec)
    chr_len = max(int(n_pairs) * 2 * L // (depth * n_chr), 2 * L + 1)
    tid = rng.integers(0, n_chr, n_pairs).astype(np.int32)
    p1 = rng.integers(0, chr_len - 2 * L, n_pairs).astype(np.int32)            # 0-based leftmost position of the template
    fwd = rng.random(n_pairs) < 0.5
    a["bs"] = rec.itemsize - 4
    a["refid"] = np.repeat(tid, 2)
    a["mrefid"] = a["refid"]
    a["pos"][0::2] = p1
    a["pos"][1::2] = p1 + L
    a["mpos"][0::2] = p1 + L
    a["mpos"][1::2] = p1
    a["tlen"][0::2] = 2 * L
    a["tlen"][1::2] = -2 * L
    a["lrn"] = 10
    a["mapq"] = 60
    a["bin"] = 4680
    a["ncig"] = 1
    a["flag"][0::2] = np.where(fwd, 99, 83)
    a["flag"][1::2] = np.where(fwd, 147, 163)
    a["lseq"] = L
    a["cigar"] = (L << 4) | 0
    idx = np.repeat(np.arange(n_pairs, dtype=np.int64), 2)
    q = a["qname"]
    q[:, 0] = ord("q")
    for d in range(8):
        q[:, 8 - d] = 48 + (idx // 10 ** d) % 10
    q[:, 9] = 0
    a["seq"] = np.asarray([0x11, 0x22, 0x44, 0x88, 0x12, 0x48, 0x21, 0x84], np.uint8)[rng.integers(0, 8, (n, (L + 1) // 2))]
    a["qual"] = 37
    a["xmtag"] = np.frombuffer(b"XMZ", np.uint8)
    # one context track per chromosome position (every read sees the same cytosine context at a position, as a
    # genome gives it), methylation per read: CpG 90 % in one read out of ten, 5 % otherwise; other contexts 1 %
    track = np.frombuffer(b".hxzu", np.uint8)[rng.choice(5, size=(n_chr, chr_len), p=[0.76, 0.135, 0.06, 0.035, 0.01])]
    hyper = np.repeat(rng.random(n_pairs) < 0.1, 2)
    xm = a["xm"]
    flat = track.reshape(-1)
    ar = np.arange(L, dtype=np.int64)[None, :]
    for lo in range(0, n, 1 << 17):
        hi = min(lo + (1 << 17), n)
        base = a["refid"][lo:hi].astype(np.int64) * chr_len + a["pos"][lo:hi].astype(np.int64)
        t = flat[base[:, None] + ar]
        u = rng.integers(0, 256, t.shape, dtype=np.uint8)         # thresholds in 1/256: 230 = 0.9, 13 = 0.05, 3 = 0.01
        thr = np.where(t == ord("z"), np.where(hyper[lo:hi, None], np.uint8(230), np.uint8(13)), np.uint8(3))
        meth = (t != ord(".")) & (u < thr)
        xm[lo:hi] = np.where(meth, t - np.uint8(32), t)            # upper case = methylated
    a["xg"][0::2] = np.where(fwd[:, None], np.frombuffer(b"XGZCT\0", np.uint8), np.frombuffer(b"XGZGA\0", np.uint8))
    a["xg"][1::2] = a["xg"][0::2]
    text = ("@HD\tVN:1.0\tSO:queryname\n" + "".join("@SQ\tSN:chrB%d\tLN:%d\n" % (i + 1, chr_len) for i in range(n_chr))).encode()
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", n_chr)
    for i in range(n_chr):
        nm = ("chrB%d" % (i + 1)).encode() + b"\0"
        head += struct.pack("<i", len(nm)) + nm + struct.pack("<i", chr_len)
    raw = memoryview(a).cast("B")

    def block(data):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    step = 65280 // rec.itemsize * rec.itemsize                              # whole records per block
    with open(path, "wb") as f, ThreadPoolExecutor(max(1, int(threads))) as pool:
        f.write(block(head))
        for out in pool.map(lambda o: block(bytes(raw[o:o + step])), range(0, len(raw), step)):
            f.write(out)
        f.write(block(b""))
    return path, n
