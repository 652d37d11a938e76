"""What do second row visits cost the one-pass lMHL kernel and the CX kernel?  The same rows (50 M / 10 M templates of 256
bytes, uniform starts rounded down to multiples of 256) once so that no row leaves its 1024- / 2048-position tile, once
shifted by 128 positions (a quarter / an eighth of the rows then reach into the next tile).  Everything else -- bytes per
row, rows per tile, calls per byte -- is equal; the difference in kernel time is the cost of the second visits."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epialleler_amd as ea
from epialleler_amd import synth, _lib
from epialleler_amd.api import ProcessedBam, _stream

lib = _lib.load()


def make(n, shift, L=256, seed=42):
    dev = "cuda:0"
    rname, start, lens = synth.uniform_layout(n, L, 4, 30, seed, 0, n, dev, False, None)
    start = ((start.to(torch.int64) // 256) * 256 + shift).to(torch.int32)      # stays sorted (monotone map)
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


for kind, n, name in (("mhl", 50_000_000, b"mhl_tiles"), ("cx", 10_000_000, b"cx_tiles")):
    for shift in (0, 128):
        bam = make(n, shift)
        fn = (lambda: ea.generateMhlReport(bam, as_device=True)) if kind == "mhl" else (lambda: ea.generateCytosineReport(bam, as_device=True))
        step, k, nrow = kernel_ms(fn, name)
        print("%s rows=%d shift=%d: step %.3f ms, kernel %.3f ms, table rows %d" % (kind, n, shift, step, k, nrow), flush=True)
        bam.close()
        del bam
        torch.cuda.empty_cache()
