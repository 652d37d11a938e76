import sys, time
sys.path.insert(0, '.')
import torch
import epialleler_amd as ea
from epialleler_amd import synth, distributed as D
bam = synth.generate_device(n_total=10_000_000, read_len=300)
eng = D.HipShardEngine(bam)
def plain(): return ea.generateCytosineReport(bam, as_device=True)
def shard(): return D.sharded_cytosine_report(eng, gather=False)
for name, f in (("plain", plain), ("sharded-path(world=1)", shard), ("plain", plain), ("sharded-path(world=1)", shard)):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): r = f()
    torch.cuda.synchronize()
    print(name, "ms/step %.3f" % ((time.perf_counter() - t0) * 100), "rows", r.nrow)
