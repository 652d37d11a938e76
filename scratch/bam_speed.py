"""Throughput of epi_preprocess_bam on a synthetic single-end XG/XM BAM (host-side producer; CPU only)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import epialleler_amd as ea
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
rng = np.random.default_rng(1)
recs = []
seq = "".join(rng.choice(list("ACGT"), 150))
xm = "".join(rng.choice(list("......hhxzzZ"), 150))
for i in range(n):
    recs.append(dict(seq=seq, flag=0, pos=1 + i * 3, tags={"XM": xm, "XG": "CT"}))
t0 = time.time(); p = H.write_bam("/tmp/big.bam", recs, refs=(("chr1", 10 ** 8),)); t1 = time.time()
print("wrote %d records, %.1f MB in %.1f s" % (n, os.path.getsize(p) / 1e6, t1 - t0))
for th in (1, 4, 8):
    t0 = time.time(); b = ea.preprocessBam(p, nthreads=th); t1 = time.time()
    print("nthreads=%d: %d templates in %.3f s -> %.2f M reads/s" % (th, b.n, t1 - t0, b.n / (t1 - t0) / 1e6))
import ctypes as C
from epialleler_amd import _lib
lib = _lib.load()
for th in (1, 1, 2, 4, 8, 16):
    opt = _lib.BamOptions(0, 0, 0, 1, 1, 1, 0, 0, -1, th, -1, 1, 0)
    t = _lib.Templates()
    t0 = time.time(); rc = lib.epi_preprocess_bam(p.encode(), C.byref(opt), C.byref(t)); t1 = time.time()
    print("C call nthreads=%d: rc %d, %d templates, %.3f s -> %.2f M reads/s" % (th, rc, t.n, t1 - t0, t.n / (t1 - t0) / 1e6), flush=True)
    lib.epi_templates_free(C.byref(t))
