"""What does a tail of long templates cost?  The config-2 stream (10 M templates of 300 bytes, uniform starts) with one template in
`every` stretched to `long_len` bytes: the lane shape of the tile kernels is picked from the batch's LONGEST row."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epialleler_amd as ea
from epialleler_amd import synth, _lib
from epialleler_amd.api import ProcessedBam, _stream

lib = _lib.load()


def make(n, every, long_len, L=300, seed=42):
    dev = "cuda:0"
    rname, start, lens = synth.uniform_layout(n, L, 4, 30, seed, 0, n, dev, False, None)
    if every:
        idx = torch.arange(0, n, every, device=dev)
        lens[idx] = long_len
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=off[1:])
    nbytes = int(off[-1].item())
    xm = torch.empty((nbytes + 15) // 16 * 16 + 64, dtype=torch.uint8, device=dev)
    xm[nbytes:] = 0xFB
    strand = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.epi_synth_fill_dev(seed, 0, n, C.c_void_p(off.data_ptr()), C.c_void_p(rname.data_ptr()), C.c_void_p(start.data_ptr()),
                                      nbytes, 0, 0, C.c_void_p(xm.data_ptr()), C.c_void_p(strand.data_ptr()), _stream(0)))
    return ProcessedBam.from_device(xm, nbytes, off, rname.contiguous(), strand, start.contiguous(), ("a", "b", "c", "d"))


def kernel_ms(fn, name, steps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.epi_prof_reset(); lib.epi_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    lib.epi_prof_enable(0)
    m, c = C.c_double(0), C.c_int64(0)
    lib.epi_prof_get(name, C.byref(m), C.byref(c))
    return dt, m.value / max(c.value, 1), r.nrow


CASES = (("cx fused", 10_000_000, b"cx_tiles", lambda b: ea.generateCytosineReport(b, as_device=True)),
         ("cx plain", 10_000_000, b"cx_tiles", lambda b: ea.generateCytosineReport(b, threshold_reads=False, as_device=True)),
         ("mhl", 10_000_000, b"mhl_tiles", lambda b: ea.generateMhlReport(b, as_device=True)))
for kind, n, name, call in CASES:
    for every, long_len in ((0, 0), (1000, 400), (1000, 600), (1000, 1000), (100000, 1000), (1000, 2500)):
        bam = make(n, every, long_len)
        step, k, nrow = kernel_ms(lambda: call(bam), name)
        print("%s rows=%d, one row in %d of %d bytes: step %.3f ms, kernel %.3f ms, table rows %d" % (kind, n, every, long_len, step, k, nrow), flush=True)
        bam.close()
        del bam
        torch.cuda.empty_cache()
