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
