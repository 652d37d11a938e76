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


def uniform_layout(n_total, mean_len=300, n_chr=4, depth=30, seed=42, row_first=0, n=None, device="cpu"):
    """(rname, start, length) of rows [row_first, row_first+n) of the stream whose starts are uniform on each
    chromosome and then sorted (torch, any device).  Template lengths are uniform in [0.8, 1.2] x mean_len."""
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
        lo, hi = max(a, row_first) - a, min(b, row_first + n) - a
        st.append(srt[lo:hi].to(torch.int32))
        rn.append(torch.full((hi - lo,), c + 1, dtype=torch.int32, device=device))
    x = torch.arange(row_first, row_first + n, dtype=torch.int64, device=device)
    lens = lo_len + _lsr(_hash3_t(seed, 6, x), 1) % (hi_len - lo_len + 1)
    cat = lambda parts, dt: torch.cat(parts) if parts else torch.empty(0, dtype=dt, device=device)
    return cat(rn, torch.int32), cat(st, torch.int32), lens


def generate_device_uniform(n_total, mean_len=300, n_chr=4, depth=30, seed=42, row_first=0, n=None,
                            gap_every=4, gap_len=50, device=None):
    """Rows [row_first, row_first+n) of the uniform-start stream, resident on `device` (bench workload cfg2u)."""
    import torch
    lib = _lib.load()
    if device is None:
        device = torch.cuda.current_device()
    n = n_total - row_first if n is None else n
    dev = "cuda:%d" % device
    rname, start, lens = uniform_layout(n_total, mean_len, n_chr, depth, seed, row_first, n, dev)
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
