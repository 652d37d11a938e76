"""lMHL on ragged templates (240-360 bytes): lane shapes 8 x 48 (one block), 4 x 96 (two blocks: 64 + 32) against each other (EPIHIP_MHLF_SHAPE)."""
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
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    lib.epi_prof_enable(0)
    m, c = C.c_double(0), C.c_int64(0)
    lib.epi_prof_get(name, C.byref(m), C.byref(c))
    return m.value / max(c.value, 1), r.nrow


n = 10_000_000
for ragged, gap_every in ((True, 0), (True, 4)):
    bam = synth.generate_device_uniform(n_total=n, mean_len=300, n_chr=4, seed=42, row_first=0, n=n, device=0, ragged=ragged, gap_every=gap_every)
    for shape in ("", "8,3", "4,4,2", "8,3", "4,4,2", ""):
        if shape:
            os.environ["EPIHIP_MHLF_SHAPE"] = shape
        else:
            os.environ.pop("EPIHIP_MHLF_SHAPE", None)
        lib.epi_options_reload()
        k, nrow = kernel_ms(lambda: ea.generateMhlReport(bam, as_device=True), b"mhl_tiles")
        print("ragged=%s gap_every=%d shape=%s: kernel %.3f ms, rows %d" % (ragged, gap_every, shape or "default", k, nrow), flush=True)
    bam.close(); del bam; torch.cuda.empty_cache()
