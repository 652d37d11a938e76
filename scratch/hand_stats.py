"""timing build -DEPI_CX_HAND_STATS (prints on stderr): how often does a wavefront step of the fused CX kernel take its pass flags from the start tile?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epialleler_amd as ea
from epialleler_amd import synth
for name, kw in (("cfg2", dict(ragged=False, gap_every=0)), ("cfg2u", dict())):
    n = 10_000_000
    bam = synth.generate_device_uniform(n_total=n, mean_len=300, n_chr=4, seed=42, row_first=0, n=n, device=0, **kw)
    for i in range(3):
        print(name, "report", i, file=sys.stderr, flush=True)
        rep = ea.generateCytosineReport(bam, as_device=True)
    torch.cuda.synchronize()
    bam.close(); del bam; torch.cuda.empty_cache()
