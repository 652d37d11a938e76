"""What do 16-byte-misaligned chunk loads cost the tile kernels?  Rows of L bytes (L a multiple of 16, so every row starts at a
16-byte-aligned offset) with uniform starts rounded down to multiples of 16 and then shifted by s: the position-aligned chunks the
kernels request are address-aligned for exactly one s (mod 16) and misaligned by the same amount for every row otherwise.
Bytes per row, rows per tile and calls per byte are equal for every s; results are correct (real kernels, product build)."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epialleler_amd as ea
from epialleler_amd import synth, _lib
from epialleler_amd.api import ProcessedBam, _stream

lib = _lib.load()


def make(n, shift, L, seed=42, raw=False):
    dev = "cuda:0"
    rname, start, lens = synth.uniform_layout(n, L, 4, 30, seed, 0, n, dev, False, None)
    if not raw:
        start = ((start.to(torch.int64) // 16) * 16 + shift).to(torch.int32)      # stays sorted (monotone map)
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


CASES = (("cx fused", 10_000_000, 304, b"cx_tiles", lambda b: ea.generateCytosineReport(b, as_device=True)),
         ("cx plain", 10_000_000, 304, b"cx_tiles", lambda b: ea.generateCytosineReport(b, threshold_reads=False, as_device=True)),
         ("cx long", 1_000_000, 10_000, b"cx_tiles", lambda b: ea.generateCytosineReport(b, threshold_reads=False, as_device=True)),
         ("mhl", 20_000_000, 304, b"mhl_tiles", lambda b: ea.generateMhlReport(b, as_device=True)))
for kind, n, L, name, call in CASES:
    for shift in (0, 1, 2, 4, 5, 8, -1):
        bam = make(n, max(shift, 0), L, raw=shift < 0)
        step, k, nrow = kernel_ms(lambda: call(bam), name)
        print("%s rows=%d L=%d shift=%s: step %.3f ms, kernel %.3f ms, table rows %d" % (kind, n, L, "uniform" if shift < 0 else shift, step, k, nrow), flush=True)
        bam.close()
        del bam
        torch.cuda.empty_cache()
