import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import epialleler_amd as ea
from epialleler_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
bam = synth.generate_device(n_total=n, read_len=300)
c = ea.CONTEXT_TO_BASES["CG"]
p1 = ea.rcpp_threshold_reads(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
p2 = ea.rcpp_threshold_reads(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
print("threshold equal:", np.array_equal(p1, p2), p1.sum(), p2.sum())
reps = [ea.rcpp_cx_report(bam, None, "Z") for _ in range(4)]
for i in range(1, 4):
    for k in reps[0]:
        eq = np.array_equal(reps[0][k], reps[i][k])
        if not eq:
            if reps[0][k].shape != reps[i][k].shape:
                print("run", i, k, "shape differs", reps[0][k].shape, reps[i][k].shape)
            else:
                d = np.nonzero(reps[0][k] != reps[i][k])[0]
                print("run", i, k, "ndiff", d.size, "first idx", d[:10], "pos", reps[0]["pos"][d[:10]], "rname", reps[0]["rname"][d[:10]], "vals", reps[0][k][d[:10]], reps[i][k][d[:10]])
        else:
            print("run", i, k, "equal")
