"""Times the un-thresholded CG report (SURVEY 8d's literal config 2) next to the default thresholded one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import epialleler_amd as ea
from epialleler_amd import synth
bam = synth.generate_device(n_total=10_000_000, read_len=300, device=0)
def t(fn, n=20):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("thresholded CG   %.3f ms" % t(lambda: ea.generateCytosineReport(bam, as_device=True)))
print("unthresholded CG %.3f ms" % t(lambda: ea.generateCytosineReport(bam, threshold_reads=False, as_device=True)))
print("unthresholded CX %.3f ms" % t(lambda: ea.generateCytosineReport(bam, threshold_reads=False, report_context="CX", as_device=True)))
