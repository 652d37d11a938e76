"""debug: the wgs_tail sharded case with 3 gloo ranks on one GPU; every rank checks its OWN (ungathered) lMHL / CX tables against the oracle of its shard."""
import os, sys, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, iters):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers as H, synth_np
    from oracle import oracle as orc
    import epialleler_amd as ea
    from epialleler_amd import distributed as D
    t = synth_np.with_long_tail(synth_np.generate(n_total=9000, read_len=300, n_chr=3), 61, 1100, first=7)
    n = t["off"].size - 1
    lo, hi = n * rank // world, n * (rank + 1) // world
    off = t["off"][lo:hi + 1]
    sub = dict(xm=t["xm"][int(off[0]):int(off[-1])], off=off - off[0], rname=t["rname"][lo:hi], strand=t["strand"][lo:hi], start=t["start"][lo:hi])
    want = {h: orc.mhl_report(sub["xm"], sub["off"], sub["rname"], sub["strand"], sub["start"], "Zz", h, 0, 0.1) for h in (0, 2)}
    bad = 0
    for it in range(iters):
        shard = ea.ProcessedBam.from_arrays(sub["xm"], sub["off"], sub["rname"], sub["strand"], sub["start"])
        eng = D.HipShardEngine(shard)
        for thr, rctx in ((True, "CG"), (False, "CX")):
            for _ in range(2):
                D.sharded_cytosine_report(eng, threshold_reads=thr, report_context=rctx, gather=False)
        for hmax in (0, 2):
            rep = D.sharded_mhl(eng, max_haplotype_window=hmax, gather=False)
            got = {k: v.cpu().numpy() for k, v in rep.items()}
            for k in want[hmax]:
                g, w = got[k], np.asarray(want[hmax][k])
                if g.shape != w.shape or not np.array_equal(g, w, equal_nan=(g.dtype.kind == "f")):
                    bad += 1
                    m = min(len(g), len(w))
                    i = int(np.argmax(g[:m] != w[:m])) if m else -1
                    print("MISMATCH rank=%d it=%d hmax=%d col=%s shapes %s %s first idx %d got %s want %s | got pos %s want pos %s" % (
                        rank, it, hmax, k, g.shape, w.shape, i, g[i:i + 4], w[i:i + 4], got["pos"][i:i + 4], np.asarray(want[hmax]["pos"])[i:i + 4]), flush=True)
                    break
        shard.close()
        dist.barrier()
    print("rank", rank, "done, mismatches", bad, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(3, port, int(sys.argv[1]) if len(sys.argv) > 1 else 8), nprocs=3, join=True)
