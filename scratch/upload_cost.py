"""What does the position-congruent layout cost an upload?  epi_batch_upload of the config-2 stream (10 M rows, 3 GB) from pinned and
from pageable host memory with EPIHIP_REALIGN = 0 / 16 in one process (epi_options_reload), upload only and upload + first report."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epialleler_amd as ea
from epialleler_amd import synth, _lib

lib = _lib.load()
n = int(os.environ.get("ROWS", "10000000"))
bam = synth.generate_device_uniform(n_total=n, mean_len=300, n_chr=4, seed=42, row_first=0, n=n, device=0, ragged=False, gap_every=0)
host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v) for k, v in bam.dev.items()}
torch.cuda.synchronize()
pageable = {k: v.numpy().copy() for k, v in host.items()}
nbytes = bam.nbytes
bam.close(); del bam
torch.cuda.empty_cache()


def run(realign, pinned, report):
    os.environ["EPIHIP_REALIGN"] = str(realign)
    lib.epi_options_reload()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pinned:
            hb = ea.ProcessedBam.from_pinned(host["xm"], nbytes, host["off"], host["rname"], host["strand"], host["start"], None, device=0)
        else:
            hb = ea.ProcessedBam.from_arrays(pageable["xm"][:nbytes], pageable["off"], pageable["rname"], pageable["strand"], pageable["start"], None, device=0)
        hb.batch()
        if report:
            ea.generateCytosineReport(hb, as_device=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        hb.close()
    return min(ts[1:])


for pinned in (True, False):
    for report in (False, True):
        a = [run(0, pinned, report), run(16, pinned, report), run(0, pinned, report), run(16, pinned, report)]
        print("%s source, %s: back to back %.2f / %.2f ms, congruent %.2f / %.2f ms" % ("pinned" if pinned else "pageable",
              "upload + first report" if report else "upload only", a[0], a[2], a[1], a[3]), flush=True)
