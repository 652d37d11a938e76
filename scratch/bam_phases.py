"""Phase times of epi_preprocess_bam on a generated paired-end BAM (host only; EPIHIP_BAM_TIMING prints them)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EPIHIP_BAM_TIMING"] = "1"
import epialleler_amd as ea
from epialleler_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
path = os.path.join(tempfile.gettempdir(), "phases_%d.bam" % n)
if not os.path.exists(path):
    synth.write_bam_paired(path, n)
for nt in (1, 4, 16, 16):
    t0 = time.perf_counter()
    bam = ea.preprocessBam(path, nthreads=nt)
    dt = time.perf_counter() - t0
    print("threads %d: %.1f ms, %.2f M records/s" % (nt, dt * 1e3, 2 * n / dt / 1e6), flush=True)
