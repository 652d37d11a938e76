"""Row-range sharded reports over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Rank k holds a contiguous range of the globally (rname,start)-sorted rows.
Per-read work (thresholding, beta) needs no exchange.  For the CX table every
rank cuts positions on the same absolute tile grid; only tiles reachable from
the rows of more than one rank ("shared" tiles: the halo at each cut, or
everything for ultra-deep amplicon data) need their raw counters summed across
ranks.  One exchange step:

  1. all_gather of each rank's (first,last) tile key              -- 16 B per rank
  2. every rank derives the same sorted list of shared tile keys and the owner
     of each (lowest rank whose range contains it)
  3. local accumulation; shared tiles go to a dense int32 slab [nshared][16][T]
  4. all_reduce(sum) of the slab                                   -- RCCL, KBs..MBs
  5. the owner applies the majority rule to its shared tiles; rows of a rank
     stay in (rname,pos,strand) order and rank order is genomic order
  6. optionally (gather=True) rows are sent to rank 0, which concatenates them in
     rank order -- the one table generateCytosineReport() returns
     (R/generateCytosineReport.R:200-207).  With gather=False every rank keeps its
     rows in HBM; the logical table is their concatenation in rank order (a report
     file is then written as one part per rank).

Steps 1-5 exist twice.  The product path is INSIDE the library (csrc/comm.hip): an engine with a communicator
attached (HipShardEngine.attach_comm: epi_comm_create = ncclCommInitRank, the id broadcast over torch.distributed) makes
one C call per report -- epi_batch_cytosine_report_sharded / epi_batch_mhl_report_sharded -- in which RCCL is called
directly on the report's stream; this module then only adds step 6.  The Python rendering of the same steps below
(torch.distributed collectives around the two-step C entry points) serves engines without a communicator: the gloo
rehearsals in which several ranks share one GPU (RCCL refuses that), and the CPU tests, where the engine is abstract
(tests/fake_engine.py supplies a numpy engine; the product engine is HipShardEngine, HIP only).
"""
import ctypes as C

import numpy as np

from . import _lib
from .api import CONTEXT_TO_BASES, ProcessedBam, Report, _ptr_array, _stream


def shared_tile_keys(ranges):
    """ranges: [(first_key, last_key)] per rank (first > last = rank has no rows).
    Returns (keys sorted ascending, owner rank per key).  A key is shared when it lies in the
    ranges of at least two ranks.  Ranges are intervals of ((rname << 32) | tile); two ranks can
    only overlap inside one rname (rows are globally sorted), so the overlap is a run of
    consecutive tile indices."""
    keys = {}
    nr = len(ranges)
    for i in range(nr):
        fi, li = ranges[i]
        if fi > li:
            continue
        for j in range(i + 1, nr):
            fj, lj = ranges[j]
            if fj > lj:
                continue
            lo, hi = max(fi, fj), min(li, lj)
            if lo > hi:
                continue
            if (lo >> 32) != (hi >> 32):
                raise ValueError("rank ranges overlap across reference sequences: shards are not contiguous "
                                 "ranges of a globally sorted row stream")
            for k in range(lo, hi + 1):
                if k not in keys:
                    keys[k] = i          # i is the lowest rank containing k (outer loop ascending)
                else:
                    keys[k] = min(keys[k], i)
    ks = sorted(keys)
    return np.asarray(ks, np.int64), np.asarray([keys[k] for k in ks], np.int32)


class HipShardEngine:
    """One rank's shard on its MI355X (product engine; no CPU path)."""

    def __init__(self, bam):
        import torch
        self.torch = torch
        self.bam = bam if isinstance(bam, ProcessedBam) else ProcessedBam.from_arrays(**bam)
        self.h = self.bam.batch()
        self.lib = _lib.load()
        self.device = torch.device("cuda", self.bam.device)
        self._slab = None
        self._range = {}
        self.last_exchange_bytes = 0          # bytes this rank handed to the last report's all-reduce(s)
        self.comm = None                      # epi_comm: RCCL behind the C ABI (attach_comm)

    def attach_comm(self, group=None, test_shared=0):
        """One RCCL communicator for this engine, created by the library (epi_comm_create = ncclCommInitRank; collective
        over `group`).  Rank 0's id (epi_comm_unique_id) travels as a 128-byte tensor over torch.distributed -- the only
        thing torch does for the sharded reports from here on.  test_shared > 0 (world size 1 only): that many tiles in
        the middle of the shard are treated as shared, so that slab, all-reduce and the owners' emit run on one GPU."""
        import torch.distributed as dist
        torch = self.torch
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        idbuf = (C.c_ubyte * 128)()
        if rank == 0:
            _lib.check(self.lib.epi_comm_unique_id(idbuf))
        if world > 1:
            backend = dist.get_backend(group)
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=self.device if backend == "nccl" else "cpu")
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            idbuf = (C.c_ubyte * 128)(*t.cpu().tolist())
        from .api import _engine
        h = C.c_void_p()
        _lib.check(self.lib.epi_comm_create(_engine(self.bam.device), idbuf, rank, world, C.byref(h)))
        self.comm = h
        if test_shared:
            self.lib.epi_comm_set_test_shared(h, int(test_shared))
        return self

    def close_comm(self):
        if self.comm is not None:
            self.lib.epi_comm_free(self.comm)
            self.comm = None

    def native_cx(self, pass_, ctx, threshold=None):
        """epi_batch_cytosine_report_sharded: steps 1-5 in one C call; returns this rank's [6, nrow] columns."""
        torch = self.torch
        nrow = C.c_int64(0)
        thr = threshold or (None, None, None, None, 0, 0.0, 0.0)
        enc = lambda v: _lib.enc(v) if v is not None else None
        _lib.check(self.lib.epi_batch_cytosine_report_sharded(
            self.h, self.comm, enc(thr[0]), enc(thr[1]), enc(thr[2]), enc(thr[3]), int(thr[4]), float(thr[5]), float(thr[6]),
            C.c_void_p(pass_.data_ptr()) if (pass_ is not None and threshold is None and self.bam.n) else None,
            _lib.enc(ctx), None, _stream(self.bam.device), C.byref(nrow)))
        n = nrow.value
        cols = torch.empty((6, n), dtype=torch.int32, device=self.device)
        if n:
            _lib.check(self.lib.epi_batch_cx_fetch_dev(self.h, _ptr_array([cols[i] for i in range(6)]), _stream(self.bam.device)))
        self.last_exchange_bytes = int(self.lib.epi_comm_last_exchange_bytes(self.comm))
        return cols

    def native_mhl(self, ctx, hmax, hmin, max_oo):
        """epi_batch_mhl_report_sharded; returns this rank's ([5, nrow] int32, [2, nrow] float64) columns."""
        torch = self.torch
        nrow = C.c_int64(0)
        _lib.check(self.lib.epi_batch_mhl_report_sharded(self.h, self.comm, _lib.enc(ctx), int(hmax), int(hmin), float(max_oo),
                                                         _stream(self.bam.device), C.byref(nrow)))
        n = nrow.value
        icols = torch.empty((5, n), dtype=torch.int32, device=self.device)
        dcols = torch.empty((2, n), dtype=torch.float64, device=self.device)
        if n:
            _lib.check(self.lib.epi_batch_mhl_fetch_dev(self.h, _ptr_array([icols[i] for i in range(5)]),
                                                        _ptr_array([dcols[i] for i in range(2)]), _stream(self.bam.device)))
        self.last_exchange_bytes = int(self.lib.epi_comm_last_exchange_bytes(self.comm))
        return icols, dcols

    def tile_positions(self, ctx="Z"):
        """Positions per CX tile for this report context string (one reported context: 2048, else 1024)."""
        return self.lib.epi_cx_tile_positions(_lib.enc(ctx))

    def mhl_fused_ok(self, ctx):
        """Can THIS rank's rows take the one-pass lMHL kernel for `ctx`?  (All ranks must take the same path: the
        answers are combined in _exchange_ranges.)"""
        ok = C.c_int32(0)
        _lib.check(self.lib.epi_batch_mhl_fused_ok(self.h, _lib.enc(ctx), _stream(self.bam.device), C.byref(ok)))
        return bool(ok.value)

    def mhl_tile_positions(self, fused):
        return self.lib.epi_mhl_fused_tile_positions() if fused else self.lib.epi_mhl_tile_positions()

    def key_range(self, kind="cx", ctx="Z"):
        # a property of the resident (immutable) shard and the tile grid: computed once per tile size
        T = self.tile_positions(ctx) if kind == "cx" else self.mhl_tile_positions(kind == "mhlf")
        if (kind, T) not in self._range:
            a, b = C.c_int64(0), C.c_int64(-1)
            _lib.check(self.lib.epi_batch_tile_key_range_for(self.h, T, _stream(self.bam.device), C.byref(a), C.byref(b)))
            self._range[(kind, T)] = (a.value, b.value)
        return self._range[(kind, T)]

    def threshold(self, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n, min_frac, max_oo):
        from .api import rcpp_threshold_reads
        return rcpp_threshold_reads(self.bam, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n, min_frac,
                                    max_oo, as_device=True)

    def _attach_cx_slab(self, keys, owned, ctx):
        torch = self.torch
        T = self.tile_positions(ctx)
        want = max(keys.size, 1) * 16 * T
        if self._slab is None or self._slab.numel() != want:
            self._slab = torch.zeros(want, dtype=torch.int32, device=self.device)
        else:
            self._slab.zero_()
        keys = np.ascontiguousarray(keys, np.int64)
        owned = np.ascontiguousarray(owned, np.int32)
        _lib.check(self.lib.epi_batch_cx_set_shared(
            self.h, C.c_void_p(keys.ctypes.data) if keys.size else None,
            C.c_void_p(owned.ctypes.data) if keys.size else None, int(keys.size),
            C.c_void_p(self._slab.data_ptr()) if keys.size else None))
        self._nshared = int(keys.size)
        return self._slab

    def cx_accumulate(self, pass_, ctx, keys, owned):
        self._attach_cx_slab(keys, owned, ctx)
        self._nshared = int(keys.size)
        nrow = C.c_int64(0)
        _lib.check(self.lib.epi_batch_cx_report_dev(
            self.h, C.c_void_p(pass_.data_ptr()) if pass_ is not None and self.bam.n else None,
            _lib.enc(ctx), _stream(self.bam.device), C.byref(nrow)))
        self._nrow = nrow.value
        return self._slab

    def cx_accumulate_fused(self, thr, ctx, keys, owned):
        """cx_accumulate with the thresholding (thr = the seven rcpp_threshold_reads arguments) done inside the tile kernel."""
        slab = self._attach_cx_slab(keys, owned, ctx)
        nrow = C.c_int64(0)
        _lib.check(self.lib.epi_batch_cytosine_report_dev(
            self.h, _lib.enc(thr[0]), _lib.enc(thr[1]), _lib.enc(thr[2]), _lib.enc(thr[3]), int(thr[4]), float(thr[5]),
            float(thr[6]), _lib.enc(ctx), None, _stream(self.bam.device), C.byref(nrow)))
        self._nrow = nrow.value
        return slab

    def cx_finish(self, ctx):
        torch = self.torch
        nrow = C.c_int64(self._nrow)
        if self._nshared:
            _lib.check(self.lib.epi_batch_cx_finish_shared(self.h, _lib.enc(ctx), _stream(self.bam.device), C.byref(nrow)))
        n = nrow.value
        cols = torch.empty((6, n), dtype=torch.int32, device=self.device)     # row i = column i, contiguous
        if n:
            _lib.check(self.lib.epi_batch_cx_fetch_dev(self.h, _ptr_array([cols[i] for i in range(6)]),
                                                       _stream(self.bam.device)))
        # detach the shared-tile state so that later single-GPU calls on this batch emit every tile
        _lib.check(self.lib.epi_batch_cx_set_shared(self.h, None, None, 0, None))
        return cols


    # ---- lMHL -----------------------------------------------------------------------------------
    def mhl_accumulate(self, ctx, hmax, hmin, max_oo, keys, owned, fused=False):
        torch = self.torch
        n = max(keys.size, 1)
        if fused:        # the one-pass kernel's slabs: int32 [n][4][T] and int64 [n][6][T] (include/epihip.h)
            T = self.lib.epi_mhl_fused_tile_positions()
            want = (n * 4 * T, n * 6 * T)
        else:
            T, NS = self.lib.epi_mhl_tile_positions(), self.lib.epi_mhl_slab_sums()
            want = (n * 16 * T, n * NS)
        if getattr(self, "_mcnt", None) is None or (self._mcnt.numel(), self._msum.numel()) != want:
            self._mcnt = torch.zeros(want[0], dtype=torch.int32, device=self.device)
            self._msum = torch.zeros(want[1], dtype=torch.int64, device=self.device)
        else:
            self._mcnt.zero_()
            self._msum.zero_()
        keys = np.ascontiguousarray(keys, np.int64)
        owned = np.ascontiguousarray(owned, np.int32)
        set_shared = self.lib.epi_batch_mhl_set_shared_fused if fused else self.lib.epi_batch_mhl_set_shared
        _lib.check(set_shared(
            self.h, C.c_void_p(keys.ctypes.data) if keys.size else None,
            C.c_void_p(owned.ctypes.data) if keys.size else None, int(keys.size),
            C.c_void_p(self._mcnt.data_ptr()) if keys.size else None,
            C.c_void_p(self._msum.data_ptr()) if keys.size else None))
        self._nshared = int(keys.size)
        nrow = C.c_int64(0)
        _lib.check(self.lib.epi_batch_mhl_report_dev(self.h, _lib.enc(ctx), int(hmax), int(hmin), float(max_oo),
                                                     _stream(self.bam.device), C.byref(nrow)))
        self._nrow = nrow.value
        return self._mcnt, self._msum

    def mhl_finish(self):
        torch = self.torch
        nrow = C.c_int64(self._nrow)
        if self._nshared:
            _lib.check(self.lib.epi_batch_mhl_finish_shared(self.h, _stream(self.bam.device), C.byref(nrow)))
        n = nrow.value
        icols = torch.empty((5, n), dtype=torch.int32, device=self.device)
        dcols = torch.empty((2, n), dtype=torch.float64, device=self.device)
        if n:
            _lib.check(self.lib.epi_batch_mhl_fetch_dev(self.h, _ptr_array([icols[i] for i in range(5)]),
                                                        _ptr_array([dcols[i] for i in range(2)]), _stream(self.bam.device)))
        _lib.check(self.lib.epi_batch_mhl_set_shared(self.h, None, None, 0, None, None))
        return icols, dcols


def _exchange_ranges(engine, kind, group, ctx="Z"):
    """all_gather of every rank's (first,last) tile key -> (ranges, world, rank).  The shards and the tile grid do not
    change, so the exchange is done once per (engine, tile grid, group) and remembered on the engine."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    memo = engine.__dict__.setdefault("_all_ranges", {})
    T = engine.tile_positions(ctx) if kind == "cx" else 0
    kind = (kind, T if kind == "cx" else ctx)
    if (kind, id(group), world) in memo:
        return memo[(kind, id(group), world)], world, rank
    if kind[0] == "cx":
        first, last = engine.key_range("cx", ctx)
        mine = [first, last]
    else:
        # lMHL: both tile grids' ranges and whether this rank's rows allow the one-pass kernel; the ranks take it only
        # if all of them can (one all_gather decides path and shared tiles alike)
        ok = int(engine.mhl_fused_ok(ctx)) if hasattr(engine, "mhl_fused_ok") else 0
        f2, l2 = engine.key_range("mhl")
        f1, l1 = engine.key_range("mhlf") if ok else (0, -1)
        mine = [f2, l2, f1, l1, ok]
    if world > 1:
        t_mine = torch.tensor(mine, dtype=torch.int64, device=engine.device)
        allr = [torch.empty(len(mine), dtype=torch.int64, device=engine.device) for _ in range(world)]
        dist.all_gather(allr, t_mine, group=group)
        rows = [[int(v) for v in t.cpu().tolist()] for t in allr]
    else:
        rows = [mine]
    if kind[0] == "cx":
        ranges = [tuple(r) for r in rows]
    else:
        fused = all(r[4] for r in rows)
        ranges = ([(r[2], r[3]) for r in rows] if fused else [(r[0], r[1]) for r in rows], fused)
    memo[(kind, id(group), world)] = ranges
    return ranges, world, rank


def _gather_rows(tensors, group, world, rank, dev):
    """Concatenates [k, n_r] tensors of all ranks on rank 0, in rank order (None elsewhere).  Point-to-point messages of
    the gloo backend read and write the tensor's memory from the host, outside any stream: device tensors go through host
    copies there (`.cpu()` waits for the kernels that fill them -- sending the device tensor raced with the table's gather
    kernel); RCCL (backend nccl) sends from the device, in stream order."""
    import torch
    import torch.distributed as dist
    host = dist.get_backend(group) != "nccl"
    cnt = torch.tensor([tensors[0].shape[1]], dtype=torch.int64, device=dev)
    cnts = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    cnts = [int(c.item()) for c in cnts]
    if rank != 0:
        if tensors[0].shape[1]:
            for t in tensors:
                dist.send(t.contiguous().cpu() if host else t.contiguous(), dst=0, group=group)
        return None
    outs = [torch.empty((t.shape[0], sum(cnts)), dtype=t.dtype, device=dev) for t in tensors]
    for o, t in zip(outs, tensors):
        o[:, :cnts[0]] = t
    pos = cnts[0]
    for r in range(1, world):
        if cnts[r]:
            for o in outs:
                buf = torch.empty((o.shape[0], cnts[r]), dtype=o.dtype, device="cpu" if host else dev)
                dist.recv(buf, src=r, group=group)
                o[:, pos:pos + cnts[r]] = buf.to(dev) if host else buf
            pos += cnts[r]
    return outs


def sharded_mhl_report(engine, ctx, hmax, hmin, max_ooctx_meth_frac, group=None, gather=True, levels=None):
    """rcpp_mhl_report over row-range shards (same exchange as the CX table, on 512-position tiles; the
    64-bit sums travel as int64 and add with wrap-around, which is what unsigned addition does)."""
    import torch.distributed as dist
    if getattr(engine, "comm", None) is not None:          # RCCL behind the C ABI: steps 1-5 in one call
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        icols, dcols = engine.native_mhl(ctx, hmax, hmin, max_ooctx_meth_frac)
        names = ("rname", "strand", "pos", "context", "coverage", "length", "lmhl")
        if gather and world > 1:
            got = _gather_rows([icols, dcols], group, world, rank, engine.device)
            if got is None:
                return None
            icols, dcols = got
        return Report(dict(zip(names, [icols[i] for i in range(5)] + [dcols[i] for i in range(2)])), levels)
    (ranges, fused), world, rank = _exchange_ranges(engine, "mhl", group, ctx)
    keys, owner = shared_tile_keys(ranges)
    owned = (owner == rank).astype(np.int32)
    if fused:
        cnt_slab, sum_slab = engine.mhl_accumulate(ctx, hmax, hmin, max_ooctx_meth_frac, keys, owned, fused=True)
    else:
        cnt_slab, sum_slab = engine.mhl_accumulate(ctx, hmax, hmin, max_ooctx_meth_frac, keys, owned)
    engine.last_exchange_bytes = (int(cnt_slab.numel() * cnt_slab.element_size() + sum_slab.numel() * sum_slab.element_size())
                                  if (world > 1 and keys.size) else 0)
    if world > 1 and keys.size:
        dist.all_reduce(cnt_slab, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(sum_slab, op=dist.ReduceOp.SUM, group=group)
    icols, dcols = engine.mhl_finish()
    names = ("rname", "strand", "pos", "context", "coverage", "length", "lmhl")
    if gather and world > 1:
        got = _gather_rows([icols, dcols], group, world, rank, engine.device)
        if got is None:
            return None
        icols, dcols = got
    cols = [icols[i] for i in range(5)] + [dcols[i] for i in range(2)]
    return Report(dict(zip(names, cols)), levels)


def sharded_mhl(engine, haplotype_context="CG", max_haplotype_window=0, min_haplotype_length=0,
                max_outofcontext_beta=0.1, group=None, gather=True, levels=None):
    """generateMhlReport() over shards (R/generateMhlReport.R:170-197)."""
    c = CONTEXT_TO_BASES[haplotype_context]
    return sharded_mhl_report(engine, c["ctx_meth"] + c["ctx_unmeth"], max_haplotype_window, min_haplotype_length,
                              max_outofcontext_beta, group, gather, levels)


def sharded_cx_report(engine, pass_, ctx, group=None, gather=True, levels=None, threshold=None):
    """rcpp_cx_report over row-range shards.  Returns the full Report on rank 0 (None elsewhere)
    when gather=True, else this rank's rows (rank order = table order).  threshold = the seven
    rcpp_threshold_reads arguments: thresholding is then done inside the tile kernel (pass_ is ignored)."""
    import torch
    import torch.distributed as dist
    names = ("rname", "strand", "pos", "context", "meth", "unmeth")
    if getattr(engine, "comm", None) is not None:          # RCCL behind the C ABI: steps 1-5 in one call
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        cols = engine.native_cx(pass_, ctx, threshold)
        if gather and world > 1:
            got = _gather_rows([cols], group, world, rank, engine.device)
            if got is None:
                return None
            cols = got[0]
        return Report({k: cols[i] for i, k in enumerate(names)}, levels)
    ranges, world, rank = _exchange_ranges(engine, "cx", group, ctx)
    dev = engine.device
    keys, owner = shared_tile_keys(ranges)
    owned = (owner == rank).astype(np.int32)
    if threshold is not None and hasattr(engine, "cx_accumulate_fused"):
        slab = engine.cx_accumulate_fused(threshold, ctx, keys, owned)
    else:
        if threshold is not None:
            pass_ = engine.threshold(*threshold)
        slab = engine.cx_accumulate(pass_, ctx, keys, owned)
    engine.last_exchange_bytes = int(slab.numel() * slab.element_size()) if (world > 1 and keys.size) else 0
    if world > 1 and keys.size:
        dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=group)      # the one data-path collective
    cols = engine.cx_finish(ctx)                                      # [6, nrow_local] int32
    names = ("rname", "strand", "pos", "context", "meth", "unmeth")
    if not gather or world == 1:
        return Report({k: cols[i] for i, k in enumerate(names)}, levels)
    got = _gather_rows([cols], group, world, rank, dev)               # rows -> rank 0, concatenated in rank order
    if got is None:
        return None
    return Report({k: got[0][i] for i, k in enumerate(names)}, levels)


def sharded_cytosine_report(engine, threshold_reads=True, threshold_context="CG", min_context_sites=2,
                            min_context_beta=0.5, max_outofcontext_beta=0.1, report_context=None,
                            group=None, gather=True, levels=None):
    """generateCytosineReport() over shards (R/generateCytosineReport.R:164-208)."""
    report_context = report_context or threshold_context
    thr = None
    if threshold_reads:
        c = CONTEXT_TO_BASES[threshold_context]
        thr = (c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], min_context_sites, min_context_beta,
               max_outofcontext_beta)
    return sharded_cx_report(engine, None, CONTEXT_TO_BASES[report_context]["ctx_meth"], group, gather, levels, threshold=thr)
