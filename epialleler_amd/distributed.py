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

The engine behind a shard is abstract (`ShardEngine`) so that the exchange logic
runs unchanged on CPU tensors under gloo in tests (tests/fake_engine.py supplies
a numpy engine there; the product engine is HipShardEngine, HIP only).
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
        self._range = None

    def tile_positions(self):
        return self.lib.epi_tile_positions()

    def key_range(self):
        # a property of the resident (immutable) shard and the tile grid: computed once
        if self._range is None:
            a, b = C.c_int64(0), C.c_int64(-1)
            _lib.check(self.lib.epi_batch_tile_key_range(self.h, _stream(self.bam.device), C.byref(a), C.byref(b)))
            self._range = (a.value, b.value)
        return self._range

    def threshold(self, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n, min_frac, max_oo):
        from .api import rcpp_threshold_reads
        return rcpp_threshold_reads(self.bam, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n, min_frac,
                                    max_oo, as_device=True)

    def cx_accumulate(self, pass_, ctx, keys, owned):
        torch = self.torch
        T = self.tile_positions()
        self._slab = torch.zeros(max(keys.size, 1) * 16 * T, dtype=torch.int32, device=self.device)
        keys = np.ascontiguousarray(keys, np.int64)
        owned = np.ascontiguousarray(owned, np.int32)
        _lib.check(self.lib.epi_batch_cx_set_shared(
            self.h, C.c_void_p(keys.ctypes.data) if keys.size else None,
            C.c_void_p(owned.ctypes.data) if keys.size else None, int(keys.size),
            C.c_void_p(self._slab.data_ptr()) if keys.size else None))
        self._nshared = int(keys.size)
        nrow = C.c_int64(0)
        _lib.check(self.lib.epi_batch_cx_report_dev(
            self.h, C.c_void_p(pass_.data_ptr()) if pass_ is not None and self.bam.n else None,
            _lib.enc(ctx), _stream(self.bam.device), C.byref(nrow)))
        self._nrow = nrow.value
        return self._slab

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


def sharded_cx_report(engine, pass_, ctx, group=None, gather=True, levels=None):
    """rcpp_cx_report over row-range shards.  Returns the full Report on rank 0 (None elsewhere)
    when gather=True, else this rank's rows (rank order = table order)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = engine.device
    first, last = engine.key_range()
    if world > 1:
        mine = torch.tensor([first, last], dtype=torch.int64, device=dev)
        allr = [torch.empty(2, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(allr, mine, group=group)
        ranges = [tuple(int(v) for v in t.cpu().tolist()) for t in allr]
    else:
        ranges = [(first, last)]
    keys, owner = shared_tile_keys(ranges)
    owned = (owner == rank).astype(np.int32)
    slab = engine.cx_accumulate(pass_, ctx, keys, owned)
    if world > 1 and keys.size:
        dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=group)      # the one data-path collective
    cols = engine.cx_finish(ctx)                                      # [6, nrow_local] int32
    names = ("rname", "strand", "pos", "context", "meth", "unmeth")
    if not gather or world == 1:
        return Report({k: cols[i] for i, k in enumerate(names)}, levels)
    # rows -> rank 0, concatenated in rank order
    cnt = torch.tensor([cols.shape[1]], dtype=torch.int64, device=dev)
    cnts = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    cnts = [int(c.item()) for c in cnts]
    if rank == 0:
        total = sum(cnts)
        out = torch.empty((6, total), dtype=torch.int32, device=dev)
        out[:, :cnts[0]] = cols
        o = cnts[0]
        for r in range(1, world):
            if cnts[r]:
                buf = torch.empty((6, cnts[r]), dtype=torch.int32, device=dev)
                dist.recv(buf, src=r, group=group)
                out[:, o:o + cnts[r]] = buf
                o += cnts[r]
        return Report({k: out[i] for i, k in enumerate(names)}, levels)
    if cols.shape[1]:
        dist.send(cols.contiguous(), dst=0, group=group)
    return None


def sharded_cytosine_report(engine, threshold_reads=True, threshold_context="CG", min_context_sites=2,
                            min_context_beta=0.5, max_outofcontext_beta=0.1, report_context=None,
                            group=None, gather=True, levels=None):
    """generateCytosineReport() over shards (R/generateCytosineReport.R:164-208)."""
    report_context = report_context or threshold_context
    pass_ = None
    if threshold_reads:
        c = CONTEXT_TO_BASES[threshold_context]
        pass_ = engine.threshold(c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"],
                                 min_context_sites, min_context_beta, max_outofcontext_beta)
    return sharded_cx_report(engine, pass_, CONTEXT_TO_BASES[report_context]["ctx_meth"], group, gather, levels)
