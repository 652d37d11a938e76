"""cfg2u apart: ragged template lengths (240-360) and gapped mates (a 50-byte 0xFB gap in every `gap_every`-th template), separately and together."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epialleler_amd as ea
from epialleler_amd import synth, _lib
lib = _lib.load()


def kernel_ms(fn, name, steps=8):
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


n = 10_000_000
for ragged, gap_every in ((False, 0), (True, 0), (False, 4), (False, 2), (False, 1), (True, 4), (False, 0)):
    bam = synth.generate_device_uniform(n_total=n, mean_len=300, n_chr=4, seed=42, row_first=0, n=n, device=0, ragged=ragged, gap_every=gap_every)
    for kind, name, call in (("cx fused", b"cx_tiles", lambda b: ea.generateCytosineReport(b, as_device=True)),
                             ("cx plain", b"cx_tiles", lambda b: ea.generateCytosineReport(b, threshold_reads=False, as_device=True)),
                             ("mhl", b"mhl_tiles", lambda b: ea.generateMhlReport(b, as_device=True))):
        step, k, nrow = kernel_ms(lambda: call(bam), name)
        print("ragged=%s gap_every=%d %s: step %.3f ms, kernel %.3f ms, rows %d" % (ragged, gap_every, kind, step, k, nrow), flush=True)
    bam.close(); del bam; torch.cuda.empty_cache()
