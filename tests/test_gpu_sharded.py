"""The sharded CX report with the real HIP engine: ranks are separate processes that share the one
MI355X of the test box and exchange the shared-tile slab over gloo (RCCL refuses two ranks on one
device, so the nccl backend itself first runs on a multi-GPU node).  The table gathered on rank 0 must
equal the single-GPU table and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

import helpers as H
import synth_np
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(name):
    rng = np.random.default_rng(77)
    if name in ("wgs", "wgs_empty", "amplicon_empty"):
        if name == "amplicon_empty":
            return synth_np.random_templates(rng, 4000, 100, 400, 1, 30)
        return synth_np.generate(n_total=9000, read_len=300, n_chr=3)
    if name == "amplicon":
        return synth_np.random_templates(rng, 4000, 100, 400, 1, 30)
    if name == "wgs_tail":                                  # a tail of long templates: ranks may pick different lane shapes, shared tiles see sliced rows
        return synth_np.with_long_tail(synth_np.generate(n_total=9000, read_len=300, n_chr=3), 61, 1100, first=7)
    return synth_np.random_templates(rng, 2500, 0, 3000, 3, 9000)       # long reads, 3 rnames


def _worker(rank, world, port, name, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import epialleler_amd as ea
    from epialleler_amd import distributed as D
    t = _case(name)
    n = t["off"].size - 1
    lo, hi = n * rank // world, n * (rank + 1) // world
    if name.endswith("_empty"):                             # the last rank holds no rows at all (it still takes part in the exchange)
        w1 = world - 1
        lo, hi = (n * rank // w1, n * (rank + 1) // w1) if rank < w1 else (n, n)
    off = t["off"][lo:hi + 1]
    shard = ea.ProcessedBam.from_arrays(t["xm"][int(off[0]):int(off[-1])], off - off[0], t["rname"][lo:hi],
                                        t["strand"][lo:hi], t["start"][lo:hi])
    eng = D.HipShardEngine(shard)
    for thr, rctx in ((True, "CG"), (False, "CX")):
        for _ in range(2):                                  # (the second call runs on the remembered, verified tile index)
            rep = D.sharded_cytosine_report(eng, threshold_reads=thr, report_context=rctx, gather=True)
        if rank == 0:
            np.savez(os.path.join(outdir, "%s_%s.npz" % (name, rctx)), **{k: v.cpu().numpy() for k, v in rep.items()})
    for hmax in (0, 2):
        rep = D.sharded_mhl(eng, max_haplotype_window=hmax, gather=True)
        if rank == 0:
            np.savez(os.path.join(outdir, "%s_mhl%d.npz" % (name, hmax)), **{k: v.cpu().numpy() for k, v in rep.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "wgs"), (3, "amplicon"), (2, "mixed"), (3, "wgs_empty"), (3, "amplicon_empty"), (3, "wgs_tail")])
def test_sharded_equals_oracle(tmp_path, world, name):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    t = _case(name)
    c = H.CONTEXT_TO_BASES["CG"]
    p = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
    for thr, rctx, letters in ((True, "CG", "Z"), (False, "CX", "ZXH")):
        want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p if thr else None, letters)
        got = dict(np.load(os.path.join(str(tmp_path), "%s_%s.npz" % (name, rctx))))
        H.assert_reports_equal(got, want)
    for hmax in (0, 2):
        want = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", hmax, 0, 0.1)
        got = dict(np.load(os.path.join(str(tmp_path), "%s_mhl%d.npz" % (name, hmax))))
        H.assert_reports_equal(got, want, float_cols=("length", "lmhl"))


def _rccl_worker(rank, port, outdir):
    """One rank on the real RCCL backend: the slab of a few tiles goes through all_reduce (an identity at world
    size 1, but the same tensors, dtypes and stream hand-over as on a multi-GPU node) before the rows are emitted."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import epialleler_amd as ea
    from epialleler_amd import distributed as D
    t = _case("wgs")
    shard = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
    eng = D.HipShardEngine(shard)
    first, last = eng.key_range()
    mine = torch.tensor([first, last], dtype=torch.int64, device=dev)
    allr = [torch.empty(2, dtype=torch.int64, device=dev)]
    dist.all_gather(allr, mine)
    assert tuple(allr[0].tolist()) == (first, last)
    keys = np.array([k for k in range(first, first + 4) if (k >> 32) == (first >> 32) and k <= last], dtype=np.int64)
    owned = np.ones(keys.size, dtype=np.int32)
    c = H.CONTEXT_TO_BASES["CG"]
    p = eng.threshold(c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
    slab = eng.cx_accumulate(p, c["ctx_meth"], keys, owned)
    assert slab.is_cuda and slab.dtype == torch.int32 and slab.numel() > 0
    before = slab.clone()
    dist.all_reduce(slab, op=dist.ReduceOp.SUM)
    assert torch.equal(before, slab)
    cols = eng.cx_finish(c["ctx_meth"])
    names = ("rname", "strand", "pos", "context", "meth", "unmeth")
    np.savez(os.path.join(outdir, "rccl_cx.npz"), **{k: cols[i].cpu().numpy() for i, k in enumerate(names)})
    mfirst, mlast = eng.key_range("mhl")
    mkeys = np.array([k for k in range(mfirst, mfirst + 3) if (k >> 32) == (mfirst >> 32) and k <= mlast], dtype=np.int64)
    cnt_slab, sum_slab = eng.mhl_accumulate("Zz", 0, 0, 0.1, mkeys, np.ones(mkeys.size, dtype=np.int32))
    assert cnt_slab.dtype == torch.int32 and sum_slab.dtype == torch.int64 and sum_slab.is_cuda
    dist.all_reduce(cnt_slab, op=dist.ReduceOp.SUM)
    dist.all_reduce(sum_slab, op=dist.ReduceOp.SUM)
    icols, dcols = eng.mhl_finish()
    mn = ("rname", "strand", "pos", "context", "coverage", "length", "lmhl")
    mc = [icols[i] for i in range(5)] + [dcols[i] for i in range(2)]
    np.savez(os.path.join(outdir, "rccl_mhl.npz"), **{k: v.cpu().numpy() for k, v in zip(mn, mc)})
    # the same through the one-pass kernel's slab layout (1024-position tiles, int32 [4][T] + int64 [6][T] per tile)
    assert eng.mhl_fused_ok("Zz")
    ffirst, flast = eng.key_range("mhlf")
    fkeys = np.array([k for k in range(ffirst, ffirst + 3) if (k >> 32) == (ffirst >> 32) and k <= flast], dtype=np.int64)
    cnt_slab, sum_slab = eng.mhl_accumulate("Zz", 0, 0, 0.1, fkeys, np.ones(fkeys.size, dtype=np.int32), fused=True)
    assert cnt_slab.numel() == fkeys.size * 4 * 1024 and sum_slab.numel() == fkeys.size * 6 * 1024 and int(cnt_slab.abs().sum()) > 0
    dist.all_reduce(cnt_slab, op=dist.ReduceOp.SUM)
    dist.all_reduce(sum_slab, op=dist.ReduceOp.SUM)
    icols, dcols = eng.mhl_finish()
    mc = [icols[i] for i in range(5)] + [dcols[i] for i in range(2)]
    np.savez(os.path.join(outdir, "rccl_mhlf.npz"), **{k: v.cpu().numpy() for k, v in zip(mn, mc)})
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_single_rank_slab_roundtrip(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    t = _case("wgs")
    c = H.CONTEXT_TO_BASES["CG"]
    p = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
    want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, "Z")
    H.assert_reports_equal(dict(np.load(os.path.join(str(tmp_path), "rccl_cx.npz"))), want)
    want = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1)
    H.assert_reports_equal(dict(np.load(os.path.join(str(tmp_path), "rccl_mhl.npz"))), want, float_cols=("length", "lmhl"))
    H.assert_reports_equal(dict(np.load(os.path.join(str(tmp_path), "rccl_mhlf.npz"))), want, float_cols=("length", "lmhl"))


def _native_worker(rank, port, outdir):
    """RCCL behind the C ABI on one rank (csrc/comm.hip): epi_comm_create with a real unique id, 5 forced shared tiles, the
    slab through a real ncclAllReduce inside epi_batch_*_report_sharded; no torch.distributed at all."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    torch.cuda.set_device(0)
    import epialleler_amd as ea
    from epialleler_amd import distributed as D
    for name in ("wgs", "amplicon", "mixed"):
        t = _case(name)
        shard = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
        eng = D.HipShardEngine(shard).attach_comm(test_shared=5)
        for thr, rctx in ((True, "CG"), (False, "CX")):
            for _ in range(2):                              # (the second call reuses the remembered plan)
                rep = D.sharded_cytosine_report(eng, threshold_reads=thr, report_context=rctx, gather=True)
            assert eng.last_exchange_bytes > 0
            np.savez(os.path.join(outdir, "native_%s_%s.npz" % (name, rctx)), **{k: v.cpu().numpy() for k, v in rep.items()})
        for hmax in (0, 2):
            rep = D.sharded_mhl(eng, max_haplotype_window=hmax, gather=True)
            assert eng.last_exchange_bytes > 0
            np.savez(os.path.join(outdir, "native_%s_mhl%d.npz" % (name, hmax)), **{k: v.cpu().numpy() for k, v in rep.items()})
        # the plain calls on the same batch afterwards: the shared-tile state must be gone
        plain = ea.generateCytosineReport(shard, threshold_reads=True)
        np.savez(os.path.join(outdir, "native_%s_plain.npz" % name), **dict(plain))
        eng.close_comm()
        shard.close()


def _native_retry_worker(rank, port, outdir):
    """The deferred synchronisation's fallback: with EPIHIP_CX_SLOT=0 every row goes through the overflow region of the pool,
    which the first report (CG) sizes for itself; the CHH report behind it -- four times the rows, one synchronisation, checks
    at the end -- finds the pool too small and reruns its first half into a scratch slab."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["EPIHIP_CX_SLOT"] = "0"
    import torch
    torch.cuda.set_device(0)
    import epialleler_amd as ea
    from epialleler_amd import distributed as D
    t = synth_np.generate(n_total=60000, read_len=300, n_chr=2)
    shard = ea.ProcessedBam.from_arrays(t["xm"], t["off"], t["rname"], t["strand"], t["start"])
    eng = D.HipShardEngine(shard).attach_comm(test_shared=5)
    for rctx in ("CG", "CG", "CHH", "CHH"):
        rep = D.sharded_cytosine_report(eng, threshold_reads=False, report_context=rctx, gather=False)
        np.savez(os.path.join(outdir, "retry_%s.npz" % rctx), **{k: v.cpu().numpy() for k, v in rep.items()})
    eng.close_comm()
    shard.close()


def test_library_comm_pool_retry(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_native_retry_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    t = synth_np.generate(n_total=60000, read_len=300, n_chr=2)
    for rctx, letters in (("CG", "Z"), ("CHH", "H")):
        want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, letters)
        got = dict(np.load(os.path.join(str(tmp_path), "retry_%s.npz" % rctx)))
        assert got["pos"].size > (65536 if rctx == "CHH" else 1000)
        H.assert_reports_equal(got, want)


def test_library_comm_single_rank(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_native_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    c = H.CONTEXT_TO_BASES["CG"]
    for name in ("wgs", "amplicon", "mixed"):
        t = _case(name)
        p = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
        for thr, rctx, letters in ((True, "CG", "Z"), (False, "CX", "ZXH")):
            want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p if thr else None, letters)
            H.assert_reports_equal(dict(np.load(os.path.join(str(tmp_path), "native_%s_%s.npz" % (name, rctx)))), want)
        H.assert_reports_equal(dict(np.load(os.path.join(str(tmp_path), "native_%s_plain.npz" % name))),
                               orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, "Z"))
        for hmax in (0, 2):
            want = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", hmax, 0, 0.1)
            H.assert_reports_equal(dict(np.load(os.path.join(str(tmp_path), "native_%s_mhl%d.npz" % (name, hmax)))), want,
                                   float_cols=("length", "lmhl"))


def _nccl_worker(rank, world, port, outdir):
    """Real RCCL ranks, one GPU each (needs >= 2 devices): sharded CX and lMHL tables gathered on rank 0."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import epialleler_amd as ea
    from epialleler_amd import distributed as D
    for name in ("wgs", "amplicon"):
        t = _case(name)
        n = t["off"].size - 1
        lo, hi = n * rank // world, n * (rank + 1) // world
        off = t["off"][lo:hi + 1]
        shard = ea.ProcessedBam.from_arrays(t["xm"][int(off[0]):int(off[-1])], off - off[0], t["rname"][lo:hi],
                                            t["strand"][lo:hi], t["start"][lo:hi], device=rank)
        eng = D.HipShardEngine(shard).attach_comm()          # RCCL behind the C ABI: epi_batch_*_report_sharded
        rep = D.sharded_cytosine_report(eng, threshold_reads=True, report_context="CG", gather=True)
        m = D.sharded_mhl(eng, gather=True)
        assert eng.last_exchange_bytes > 0
        eng.close_comm()
        eng2 = D.HipShardEngine(shard)                      # ... and the same exchange through torch.distributed
        rep2 = D.sharded_cytosine_report(eng2, threshold_reads=True, report_context="CG", gather=True)
        if rank == 0:
            assert all(torch.equal(rep[k], rep2[k]) for k in rep)
            np.savez(os.path.join(outdir, "nccl_%s_cx.npz" % name), **{k: v.cpu().numpy() for k, v in rep.items()})
            np.savez(os.path.join(outdir, "nccl_%s_mhl.npz" % name), **{k: v.cpu().numpy() for k, v in m.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_two_gpus(tmp_path):
    """The nccl (RCCL over xGMI) path with more than one rank; skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs at least two GPUs")
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 4)
    mp.spawn(_nccl_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    c = H.CONTEXT_TO_BASES["CG"]
    for name in ("wgs", "amplicon"):
        t = _case(name)
        p = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
        want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, "Z")
        H.assert_reports_equal(dict(np.load(os.path.join(str(tmp_path), "nccl_%s_cx.npz" % name))), want)
        wm = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1)
        H.assert_reports_equal(dict(np.load(os.path.join(str(tmp_path), "nccl_%s_mhl.npz" % name))), wm, float_cols=("length", "lmhl"))
