#!/usr/bin/env python3
"""Headline benchmark: Mreads/s of generateCytosineReport() on synthetic PE150
templates resident in HBM (BASELINE.json metric; config 2 at N=1, weak-scaled by
row-range shards at N>1).  One process per GPU over RCCL (backend "nccl") for the
one shared-tile all-reduce and the row gather.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8                      # starts its own 8 ranks (torch.distributed.run children)

A "step" is one full generateCytosineReport(bam) on the resident batch: thresholding,
the tile index, the LDS-histogram tile kernel with the majority rule, and the ordered
gather into the six output columns (which stay in HBM, as the input does).  A "read"
is one template row (a merged read pair).

The workload is SURVEY 8d's stream literally: starts uniform-random on the genome, sorted (on the device), L = 300.
Besides the contract line's `value` (inputs resident in HBM) the N=1 line carries
  streamed   the same report with the batch starting in pinned host memory (upload + report), SURVEY 8d
  d2h        the report table copied to pinned host memory
  host_out   the report as a host binding (the R shim) gets it: table in host memory, one-call and two-step forms
  selfcheck  N=1: the last timed report of the timed batch against the CPU oracle on three row windows (exit non-zero on a difference)
  tile_hint_off  the step with EPIHIP_TILE_HINT=0 (tile index counted and scanned by every call)
  layout_off     the step on the same rows adopted back to back (no position-congruent copy; config.batch_ms is what the copy costs once)
  n1_same_stream  the step at N=1 on the 3-chromosome stream the N>1 runs use (the base for a 1 -> N ratio)
  sharded_1rank  the same step through epi_batch_cytosine_report_sharded (RCCL inside the library) on one rank: 14 forced shared
             tiles + a real ncclAllReduce
  cfg2g / cfg2u / cfg2p / cfg2t   the stream's effects apart: round 1/2's jittered-grid starts / ragged lengths and gapped
             mates / one 20 000-row pile-up in the stream / one template in 1000 stretched to 1 kb (the lane shapes follow the
             bulk of the rows, not the longest one)
  strong_cfg3  BASELINE config 3 (100 M templates in total, split over the N GPUs)
and every N>1 run first checks, on a reduced-size stream whose cuts lie inside chromosomes, that the sharded tables equal
the single-GPU tables and that the shared-tile exchange really carried bytes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Streams (epialleler_amd/synth.py, csrc/synth.hip; one context track per chromosome position, depth 30):
#   "uniform"  SURVEY 8d's literal wording: starts uniform-random on the chromosome, then sorted (on the device); every
#              template exactly L bytes.  The headline stream since round 3.
#   "grid"     round 1/2's headline stream: start_j = 1 + floor(j G / N) + jitter (sorted by construction)
#   "ragged"   uniform starts, template lengths 0.8-1.2 L, a 50-byte 0xFB gap between the mates of every fourth template
#   pileup     (uniform) + `pileup` consecutive rows that all start at one position: an amplicon hot spot in WGS-like data
WORKLOADS = {
    # name: (rows per GPU, template bytes, report)
    "cfg2": dict(rows=10_000_000, read_len=300, kind="cx", threshold=True, report_context="CG", stream="uniform",
                 desc="simulateBam-like 10M PE150 templates (L=300 fixed, uniform-random starts sorted on device, depth 30), "
                      "generateCytosineReport defaults (threshold.reads=TRUE, CG)"),
    "cfg2g": dict(rows=10_000_000, read_len=300, kind="cx", threshold=True, report_context="CG", stream="grid",
                  desc="as cfg2 on round 1/2's stream: jittered grid starts instead of uniform-random ones"),
    "cfg2u": dict(rows=10_000_000, read_len=300, kind="cx", threshold=True, report_context="CG", stream="ragged",
                  desc="as cfg2 with template lengths 240-360 and a 50-byte 0xFB gap in every fourth template"),
    "cfg2p": dict(rows=10_000_000, read_len=300, kind="cx", threshold=True, report_context="CG", stream="uniform", pileup=20_000,
                  desc="as cfg2 with one 20 000-row pile-up (all rows starting at one position) in the middle of the stream"),
    "cfg2t": dict(rows=10_000_000, read_len=300, kind="cx", threshold=True, report_context="CG", stream="uniform", tail=(1000, 1000),
                  desc="as cfg2 with one template in 1000 stretched to 1000 bytes (an insert-size tail: the batch's longest row is 1 kb)"),
    "cfg2n": dict(rows=10_000_000, read_len=300, kind="cx", threshold=False, report_context="CG", stream="uniform",
                  desc="10M PE150 templates, generateCytosineReport(threshold.reads=FALSE): SURVEY 8d's un-thresholded config 2"),
    "cfg2cx": dict(rows=10_000_000, read_len=300, kind="cx", threshold=False, report_context="CX", stream="uniform",
                   desc="10M PE150 templates, generateCytosineReport(threshold.reads=FALSE, report.context='CX')"),
    "cfg3": dict(rows=100_000_000, strong=True, read_len=300, kind="cx", threshold=True, report_context="CG", stream="uniform",
                 desc="100M PE150 templates in total (split over the GPUs), generateCytosineReport defaults"),
    "cfg4": dict(rows=50_000_000, read_len=300, kind="mhl", stream="uniform", desc="50M PE150 templates, generateMhlReport defaults"),
    "cfg4d": dict(rows=50_000_000, read_len=300, kind="mhl", stream="uniform", pileup=20_000,
                  desc="as cfg4 with one 20 000-row pile-up in the middle of the stream"),
    "cfg5": dict(rows=5_000_000, read_len=10000, kind="cx", threshold=False, report_context="CG", stream="uniform",
                 desc="5M long-read (10 kb) templates, generateCytosineReport(threshold.reads=FALSE)"),
    # the only throughput the reference publishes (vignettes/epialleleR.Rmd:172-176): BAM on disk -> CX report on disk
    "file": dict(rows=500_000, read_len=300, kind="file",
                 desc="name-sorted paired-end XG/XM BAM on disk (2 x 150 bp mates per template) -> preprocessBam -> "
                      "generateCytosineReport defaults -> TSV report on disk"),
}


def _parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--read-len", type=int, default=0, help="override the template length of the workload (experiments)")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="rows timed on the CPU oracle (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for rehearsals")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--gather", action="store_true", help="N>1: also send every rank's rows to rank 0 inside the timed step")
    ap.add_argument("--check", action="store_true", help="N>1: rank 0 checks the FULL-size gathered table against a single-GPU "
                                                         "run of the whole stream (a reduced-size check always runs)")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the self-check (N=1: the timed report against the oracle on "
                                                                "row windows; N>1: reduced-size sharded == single-GPU tables)")
    ap.add_argument("--selfcheck-rows", type=int, default=600_000, help="N>1: total rows of the reduced-size check")
    ap.add_argument("--no-extras", action="store_true", help="only the contract fields (no streamed / d2h / cfg2u / strong_cfg3)")
    ap.add_argument("--strong-steps", type=int, default=3)
    ap.add_argument("--host-threads", type=int, default=0, help="file workload: BGZF inflate / packing / report writer threads (0 = min(cores, 16))")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: this process becomes a pure
    launcher.  It has not touched HIP (torch is not even imported here) and never execs: the N ranks are fresh child
    processes started by torch.distributed.run; their output is relayed and the exit code is the children's."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


class Ctx:
    """What every measurement below needs: the rank layout, torch, the library."""

    def __init__(self, args):
        import numpy as np
        import torch
        import torch.distributed as dist
        self.np, self.torch, self.dist, self.args = np, torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, self.world))
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(args.backend)
        import epialleler_amd as ea
        from epialleler_amd import _lib, distributed as D, synth
        self.ea, self.D, self.synth, self.lib = ea, D, synth, _lib.load()

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.world == 1:
            return dt
        t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v):
        if self.world == 1:
            return int(v)
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t)
        return int(t.item())


def attach_native(cx, eng):
    """N > 1 on RCCL: the shard's communicator is created by the library (epi_comm_create = ncclCommInitRank; csrc/comm.hip), so
    that the report's exchange is one C call per rank.  Every rank must end up on the same path: whether the creation worked is
    agreed on over torch.distributed, and if it failed anywhere every rank uses the torch.distributed rendering of the same
    steps instead (the line's `sharding` field says which one ran)."""
    if eng is None or cx.args.backend != "nccl" or cx.args.share_gpu:
        return False
    ok = 1
    try:
        eng.attach_comm()
    except Exception as e:                                   # pragma: no cover (needs a broken RCCL set-up)
        ok = 0
        print("rank %d: epi_comm_create failed (%s); falling back to torch.distributed collectives" % (cx.rank, e), file=sys.stderr, flush=True)
    t = cx.torch.tensor([ok], dtype=cx.torch.int64, device=cx.dev)
    cx.dist.all_reduce(t, op=cx.dist.ReduceOp.MIN)
    if int(t.item()) != 1:
        eng.close_comm()
        return False
    return True


def n_chr_for(world):
    """Chromosomes of the synthetic genome.  SURVEY 8d: 4.  With 2 or 4 ranks (and 4 of the 7 cuts at 8) equal row ranges
    of a 4-chromosome stream are cut exactly at chromosome boundaries: no tile is shared and the exchange step never
    runs.  N > 1 therefore uses 3 chromosomes (same rows, bytes, depth per GPU): every cut lies inside a chromosome."""
    return 4 if world == 1 else 3


def make_batch(cx, wl, rows, L, n_total, seed=42, n_chr=None):
    kw = dict(n_total=n_total, n_chr=n_chr or n_chr_for(cx.world), seed=seed, row_first=cx.rank * rows, n=rows, device=cx.local)
    stream = wl.get("stream", "uniform")
    if stream == "grid":
        return cx.synth.generate_device(read_len=L, **kw)
    if wl.get("pileup"):
        kw["pileup"] = (n_total // 2 + 1234, int(wl["pileup"]))
    if wl.get("tail"):
        kw["tail"] = wl["tail"]
    if stream == "ragged":
        return cx.synth.generate_device_uniform(mean_len=L, **kw)
    return cx.synth.generate_device_uniform(mean_len=L, ragged=False, gap_every=0, **kw)


def make_step(cx, wl, bam, eng, gather):
    ea, D = cx.ea, cx.D
    if wl["kind"] == "mhl":
        if cx.world == 1:
            return lambda: ea.generateMhlReport(bam, as_device=True)
        return lambda: D.sharded_mhl(eng, gather=gather, levels=bam.levels)
    if cx.world == 1:
        return lambda: ea.generateCytosineReport(bam, threshold_reads=wl["threshold"], report_context=wl["report_context"],
                                                 as_device=True)
    return lambda: D.sharded_cytosine_report(eng, threshold_reads=wl["threshold"], report_context=wl["report_context"],
                                             gather=gather, levels=bam.levels)


def timed_run(cx, wl, rows, L, steps, warmup, gather=False, keep=False, n_chr=None):
    """W untimed + exactly K timed steps of `wl` on `rows` rows per rank, barrier + synchronize on both sides,
    max over ranks.  Returns a dict (and the batch / last report when keep=True)."""
    import ctypes as C
    n_total = rows * cx.world
    bam = make_batch(cx, wl, rows, L, n_total, n_chr=n_chr)
    cx.torch.cuda.synchronize()
    t0 = time.perf_counter()
    bam.batch()                                            # adopt + row statistics + the engine's own row layout (epi_batch_realign): once per batch
    cx.torch.cuda.synchronize()
    batch_ms = (time.perf_counter() - t0) * 1e3
    eng = cx.D.HipShardEngine(bam) if cx.world > 1 else None
    native = attach_native(cx, eng)                        # RCCL behind the C ABI (one communicator per rank, created by the library)
    step = make_step(cx, wl, bam, eng, gather)
    # setup, not steps: the first calls size the engine's row pool / record space for this workload and the caching
    # allocator's blocks for the two output tables that are alive at a time (rep = step() frees the previous one late)
    rep = step()
    cx.torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = step()
    cx.torch.cuda.synchronize()
    # ... and the GPU's clocks: a process that has queued a few milliseconds of work is timed 3-4 % slower than one that has been
    # busy for a quarter of a second (profiles/r04_layout.txt, section 7), so the step runs untimed for ~0.3 s first -- the same
    # count on every rank (the sharded step has collectives) -- whatever --warmup says
    n_ramp = int(0.3 / max(cx.max_over_ranks(time.perf_counter() - t0), 1e-4))
    n_ramp = max(5, min(n_ramp, 400))
    for _ in range(n_ramp):
        rep = step()
    for _ in range(warmup):
        rep = step()
    cx.barrier()
    cx.lib.epi_prof_reset()
    cx.lib.epi_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        rep = step()
    cx.barrier()
    dt = time.perf_counter() - t0
    cx.lib.epi_prof_enable(0)
    dt = cx.max_over_ranks(dt)
    kernels = {}
    for nm in (b"threshold", b"cx_tiles", b"cx_deep", b"cx_heavy", b"mhl_rows", b"mhl_tiles", b"mhl_deep", b"mhl_heavy", b"tile_index", b"gather"):
        m2, c2 = C.c_double(0), C.c_int64(0)
        cx.lib.epi_prof_get(nm, C.byref(m2), C.byref(c2))
        if c2.value:
            kernels[nm.decode()] = round(m2.value / c2.value, 4)
    nrow_local = rep.nrow if rep is not None else 0
    out = dict(dt=dt, ms_per_step=dt / steps * 1e3, n_total=n_total, rows=rows, L=L, kernels=kernels, native_comm=bool(native),
               batch_ms=batch_ms, layout=int(cx.lib.epi_batch_layout(bam.batch())), ramp_steps=n_ramp,
               nrow_local=nrow_local, nbytes_local=bam.nbytes,
               exchange_bytes=getattr(eng, "last_exchange_bytes", 0) if eng is not None else 0)
    if keep:
        out["bam"], out["rep"], out["eng"] = bam, rep, eng
    else:
        if eng is not None:
            eng.close_comm()
        del rep, step, eng
        bam.close()
    return out


def selfcheck(cx, n_total):
    """N>1: the sharded tables of a reduced-size stream, gathered on rank 0, must equal the single-GPU tables of the
    same stream (CX with thresholding, and lMHL), AND the exchange step must have run: the stream's cuts lie inside
    chromosomes (n_chr_for), so every cut has shared tiles and the slab all-reduce carries bytes.  Every rank raises on a
    mismatch or on an exchange of zero bytes."""
    torch, ea, D = cx.torch, cx.ea, cx.D
    rows = (n_total + cx.world - 1) // cx.world
    n_total = rows * cx.world
    wl = WORKLOADS["cfg2"]
    bam = make_batch(cx, wl, rows, 300, n_total, seed=5)
    eng = D.HipShardEngine(bam)
    attach_native(cx, eng)
    got_cx = D.sharded_cytosine_report(eng, threshold_reads=True, report_context="CG", gather=True, levels=bam.levels)
    xb_cx = cx.sum_over_ranks(eng.last_exchange_bytes)
    got_mhl = D.sharded_mhl(eng, gather=True, levels=bam.levels)
    xb_mhl = cx.sum_over_ranks(eng.last_exchange_bytes)
    ok = 1
    nrow = 0
    if cx.rank == 0:
        whole = cx.synth.generate_device_uniform(n_total=n_total, mean_len=300, n_chr=n_chr_for(cx.world), seed=5, device=cx.local,
                                                 ragged=False, gap_every=0)
        ref_cx = ea.generateCytosineReport(whole, threshold_reads=True, report_context="CG", as_device=True)
        ref_mhl = ea.generateMhlReport(whole, as_device=True)
        ok = int(all(bool(torch.equal(ref_cx[k], got_cx[k])) for k in ref_cx) and
                 all(bool(torch.equal(ref_mhl[k].view(torch.int64) if ref_mhl[k].dtype == torch.float64 else ref_mhl[k],
                                      got_mhl[k].view(torch.int64) if got_mhl[k].dtype == torch.float64 else got_mhl[k]))
                     for k in ref_mhl))
        nrow = ref_cx.nrow
        whole.close()
    t = torch.tensor([ok], dtype=torch.int64, device=cx.dev)
    cx.dist.broadcast(t, src=0)
    eng.close_comm()
    bam.close()
    if int(t.item()) != 1:
        raise SystemExit("selfcheck FAILED: sharded table differs from the single-GPU table (%d rows over %d ranks)" % (n_total, cx.world))
    if xb_cx <= 0 or xb_mhl <= 0:
        raise SystemExit("selfcheck FAILED: the exchange step did not run (all-reduce bytes CX %d, lMHL %d over %d ranks): the "
                         "check would not have covered the shared-tile path" % (xb_cx, xb_mhl, cx.world))
    return {"rows_total": n_total, "cx_rows": nrow, "ok": True, "all_reduce_bytes": {"cx": int(xb_cx), "mhl": int(xb_mhl)},
            "what": "sharded CX (thresholded) and lMHL tables, gathered on rank 0 over %s, bit-equal to the single-GPU tables of "
                    "the same stream; %d cuts inside chromosomes, shared tiles all-reduced" % (cx.args.backend, cx.world - 1)}


def n1_selfcheck(cx, wl, res, L):
    """N=1: the LAST TIMED report of the timed batch against the CPU oracle (the checker, never the thing measured) on
    three row windows -- first, middle and last rows of the batch -- cut where no read outside the window can reach
    (oracle/windows.py; rule: src/rcpp_cx_report.cpp:108-131, src/rcpp_mhl_report.cpp:138-198), plus strict
    (rname, pos, strand) order of the whole table.  Integer columns bit-exact, float64 columns bitwise.  Untimed; raises
    SystemExit on a mismatch."""
    from oracle import windows as W
    rows = res["rows"]
    if wl.get("stream") == "ragged":
        L = L * 6 // 5                                     # template lengths 0.8-1.2 L: the trimmed margin is the LONGEST read
    if wl.get("tail"):
        L = max(L, int(wl["tail"][1]))
    wrows = 20000 if L <= 1000 else max(300, 20000 * 300 // L)
    letters = {"CG": "Z", "CHG": "X", "CHH": "H", "CxG": "ZX", "CX": "ZXH"}[wl.get("report_context", "CG")]
    fn = W.oracle_for(wl["kind"], wl.get("threshold", False), letters)
    t0 = time.perf_counter()
    try:
        r = W.check_windows(res["rep"], res["bam"].dev, rows, fn, float_cols=("length", "lmhl") if wl["kind"] == "mhl" else (),
                            L=L, wrows=wrows, starts=(0, rows // 2, rows - wrows))
    except AssertionError as e:
        raise SystemExit("selfcheck FAILED: the timed report differs from the oracle: %s" % e)
    r["seconds"] = round(time.perf_counter() - t0, 2)
    r["what"] = ("the last timed report of the timed batch == the CPU oracle (oracle/epi_oracle.c) on %d-row windows at the "
                 "first, middle and last rows (one read length trimmed at cut ends), all columns bit-exact; whole table in "
                 "strict (rname, pos, strand) order" % wrows)
    return r


def tile_hint_off(cx, wl, res, steps):
    """N=1 extra: the same step with EPIHIP_TILE_HINT=0 -- the tile index counted, scanned and asked for by every call
    instead of rebuilt from the block offsets remembered from the first call on the batch."""
    bam = res["bam"]
    os.environ["EPIHIP_TILE_HINT"] = "0"
    cx.lib.epi_options_reload()
    try:
        step = make_step(cx, wl, bam, None, False)
        for _ in range(2):
            step()
        cx.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        cx.torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        del os.environ["EPIHIP_TILE_HINT"]
        cx.lib.epi_options_reload()
    return {"ms_per_step": round(ms, 4), "vs_hinted": round(ms / res["ms_per_step"], 3),
            "what": "EPIHIP_TILE_HINT=0: count pass + scan + host round trip for the tile count in every step"}


def layout_off(cx, wl, res, steps):
    """N=1 extra: the same step on the same rows adopted WITHOUT the engine's position-congruent copy (rows back to back as the
    generator wrote them: the position-aligned chunks of the tile kernels then have any byte alignment)."""
    bam = res["bam"]
    d = bam.dev
    raw = cx.ea.ProcessedBam.from_device(d["xm"], bam.nbytes, d["off"], d["rname"], d["strand"], d["start"], bam.levels, realign=False)
    try:
        import ctypes as C
        step = make_step(cx, wl, raw, None, False)
        for _ in range(3):
            step()
        cx.torch.cuda.synchronize()
        cx.lib.epi_prof_reset()
        cx.lib.epi_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        cx.torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        cx.lib.epi_prof_enable(0)
        m2, c2 = C.c_double(0), C.c_int64(0)
        cx.lib.epi_prof_get(b"mhl_tiles" if wl["kind"] == "mhl" else b"cx_tiles", C.byref(m2), C.byref(c2))
    finally:
        raw.close()
    return {"ms_per_step": round(ms, 4), "vs_congruent": round(ms / res["ms_per_step"], 3), "kernel_ms": round(m2.value / max(c2.value, 1), 4),
            "what": "rows back to back (epi_batch_adopt without epi_batch_realign): chunk loads at any byte alignment, and the tile "
                    "table rebuilt (verified) by every step because the columns stay the caller's"}


def streamed_and_d2h(cx, wl, res):
    """N=1: the same report when the batch starts in pinned host memory (what preprocessBam hands over): upload
    (hipMemcpyAsync straight from the pinned columns) + report; and the finished table copied to pinned host memory."""
    torch, ea = cx.torch, cx.ea
    bam = res["bam"]
    host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v) for k, v in bam.dev.items()}
    torch.cuda.synchronize()
    step_kw = dict(threshold_reads=wl["threshold"], report_context=wl["report_context"], as_device=True)
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hb = ea.ProcessedBam.from_pinned(host["xm"], bam.nbytes, host["off"], host["rname"], host["strand"], host["start"],
                                         bam.levels, device=cx.local)
        rep = ea.generateCytosineReport(hb, **step_kw)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        hb.close()
    t_stream = min(times[1:])
    rep = res["rep"]
    pinned = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in rep.items()}
    d2h = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, v in rep.items():
            pinned[k].copy_(v, non_blocking=True)
        torch.cuda.synchronize()
        d2h.append(time.perf_counter() - t0)
    table_bytes = sum(v.numel() * v.element_size() for v in rep.values())
    in_bytes = sum(v.numel() * v.element_size() for v in host.values())
    return ({"value": round(res["n_total"] / t_stream / 1e6, 2), "unit": "Mreads/s", "ms": round(t_stream * 1e3, 3),
             "host_bytes": int(in_bytes), "GBps_pcie": round(in_bytes / t_stream / 1e9, 2),
             "what": "pinned host SoA -> hipMemcpyAsync -> HBM -> one report (upload + report, best of 2 after a warm-up)"},
            {"ms": round(min(d2h) * 1e3, 3), "bytes": int(table_bytes), "GBps": round(table_bytes / min(d2h) / 1e9, 2),
             "what": "the report table (6 int32 columns) copied to pinned host memory"})


def sharded_one_rank(cx, wl, res, steps):
    """N=1: what the sharded entry point costs per step on top of the plain call -- the same workload through
    epi_batch_cytosine_report_sharded (csrc/comm.hip: RCCL behind the C ABI) on a world-size-1 communicator with 14 forced
    shared tiles (two per cut of an 8-GPU run): slab zeroing, tile table with slots, slab dump, a REAL ncclAllReduce of the
    slab on the report's stream, the owners' emit, both host synchronisations.  No torch.distributed involved."""
    torch, D = cx.torch, cx.D
    bam = res["bam"]
    eng = D.HipShardEngine(bam).attach_comm(test_shared=14)
    try:
        step = lambda: D.sharded_cytosine_report(eng, threshold_reads=wl["threshold"], report_context=wl["report_context"], gather=False)
        rep = step()
        ref = res["rep"]
        same = bool(all(torch.equal(rep[k], ref[k]) for k in ref))
        xbytes = int(eng.last_exchange_bytes)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        eng.close_comm()
    if not same:
        raise SystemExit("sharded_1rank: the table through the shared-tile path differs from the plain call's")
    if xbytes <= 0:
        raise SystemExit("sharded_1rank: the all-reduce carried no bytes")
    return {"ms_per_step": round(ms, 4), "plain_ms_per_step": round(res["ms_per_step"], 4),
            "overhead_ms": round(ms - res["ms_per_step"], 4), "shared_tiles": 14, "all_reduce_bytes": xbytes,
            "backend": "RCCL called by the library (epi_comm, world size 1)", "table_equal_to_plain": True,
            "what": "the same workload through epi_batch_cytosine_report_sharded on one rank: 14 forced shared tiles dumped to the "
                    "slab, ncclAllReduce(sum) of the slab on the report's stream, owners' emit, ordered columns"}


def host_out(cx, wl, res, steps=5):
    """N=1: what a host binding (the R shim) sees -- the report with the table in HOST memory.  `table`: the one-call entry
    point (epi_batch_cytosine_report / epi_batch_cx_report: library-owned malloc'ed columns).  `into`: the two-step form the
    shim uses (epi_batch_*_begin, then epi_batch_cx_fetch_host straight into caller-owned columns), with fresh destination
    arrays per call (what R does: new vectors, page faults included) and with reused ones."""
    import ctypes as C
    import numpy as np
    lib, torch, _lib = cx.lib, cx.torch, cx.ea._lib
    bam = res["bam"]
    b = bam.batch()
    c = cx.ea.CONTEXT_TO_BASES["CG"]
    ctx = _lib.enc(cx.ea.CONTEXT_TO_BASES[wl["report_context"]]["ctx_meth"])
    thr = [_lib.enc(c[k]) for k in ("ctx_meth", "ctx_unmeth", "ooctx_meth", "ooctx_unmeth")]

    def table():
        t = _lib.CxTable()
        if wl["threshold"]:
            _lib.check(lib.epi_batch_cytosine_report(b, thr[0], thr[1], thr[2], thr[3], 2, 0.5, 0.1, ctx, None, C.byref(t)))
        else:
            _lib.check(lib.epi_batch_cx_report(b, None, ctx, C.byref(t)))
        n = t.nrow
        lib.epi_cx_table_free(C.byref(t))
        return n

    keep = {}

    def into(fresh):
        nrow = C.c_int64(0)
        if wl["threshold"]:
            _lib.check(lib.epi_batch_cytosine_report_begin(b, thr[0], thr[1], thr[2], thr[3], 2, 0.5, 0.1, ctx, None, C.byref(nrow)))
        else:
            _lib.check(lib.epi_batch_cx_report_begin(b, None, ctx, C.byref(nrow)))
        n = nrow.value
        if fresh or keep.get("n") != n:
            keep["cols"] = [np.empty(n, np.int32) for _ in range(6)]
            keep["n"] = n
        arr = (C.c_void_p * 6)(*[a.ctypes.data for a in keep["cols"]])
        _lib.check(lib.epi_batch_cx_fetch_host(b, arr, None))
        return n

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            n = fn()
        return (time.perf_counter() - t0) / steps * 1e3, n

    ms_table, n1 = timeit(table)
    ms_fresh, n2 = timeit(lambda: into(True))
    ms_reuse, n3 = timeit(lambda: into(False))
    rep = res["rep"]
    ok = n1 == n2 == n3 == rep.nrow and all(np.array_equal(keep["cols"][i], rep[k].cpu().numpy())
                                            for i, k in enumerate(("rname", "strand", "pos", "context", "meth", "unmeth")))
    if not ok:
        raise SystemExit("host_out: the host table differs from the device table")
    return {"ms_table": round(ms_table, 3), "ms_into_fresh": round(ms_fresh, 3), "ms_into_reused": round(ms_reuse, 3),
            "rows": int(n1), "bytes": int(n1) * 24,
            "what": "report with the table in host memory: library-owned table (one call) / caller-owned columns allocated per "
                    "call / caller-owned columns reused; pageable destinations through the engine's pinned staging pipeline"}


def file_workload(cx, args):
    """BAM on disk -> report on disk, the reference's only published throughput (250-400 thousand reads/s on one
    core of an i7-7700, HTSlib decode included).  One step = preprocessBam (host: BGZF inflate + template packing into
    pinned SoA) + upload + generateCytosineReport defaults (fused kernel) + table to host + threaded TSV writer."""
    import tempfile
    ea, torch = cx.ea, cx.torch
    wl = WORKLOADS["file"]
    pairs = args.rows or wl["rows"]
    nth = args.host_threads or max(1, min(os.cpu_count() or 1, 16))
    tmp = tempfile.mkdtemp(prefix="epihip_bench_")
    path, nrec = cx.synth.write_bam_paired(os.path.join(tmp, "in.bam"), pairs, threads=nth)
    out_path = os.path.join(tmp, "report.tsv")
    parts = []

    def step():
        t0 = time.perf_counter()
        bam = ea.preprocessBam(path, nthreads=nth)
        t1 = time.perf_counter()
        bam.batch(cx.local)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rep = ea.generateCytosineReport(bam)
        t3 = time.perf_counter()
        ea.writeReport(rep, out_path, nthreads=nth)
        t4 = time.perf_counter()
        parts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        n, nrow = bam.n, rep.nrow
        bam.close()
        return n, nrow

    for _ in range(max(1, args.warmup)):
        step()
    parts.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_templ, nrow = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    avg = [sum(p[i] for p in parts) / len(parts) * 1e3 for i in range(4)]
    out = {
        "metric": "Mreads/s BAM file -> CX report file (generateCytosineReport; read = BAM record)",
        "value": round(nrec * args.steps / dt / 1e6, 4), "unit": "Mreads/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "file: %s" % wl["desc"], "bam_records": nrec, "templates": n_templ, "bam_bytes": os.path.getsize(path),
                   "report_rows": nrow, "report_bytes": os.path.getsize(out_path), "host_threads": nth,
                   "ms": {"preprocessBam": round(avg[0], 2), "upload": round(avg[1], 2), "report_incl_table_to_host": round(avg[2], 2),
                          "writeReport": round(avg[3], 2)},
                   "templates_per_s_M": round(n_templ * args.steps / dt / 1e6, 4),
                   "BAM_MB_per_s": round(os.path.getsize(path) * args.steps / dt / 1e6, 1)},
        "published_reference": "250-400 thousand reads/s (30-50 MB/s of BAM), one core of an Intel Core i7-7700, HTSlib decode "
                               "included (vignettes/epialleleR.Rmd:172-176); different hardware, so vs_baseline stays null",
        "roofline": None, "cpu_baseline": None,
    }
    for f in (path, out_path):
        try:
            os.remove(f)
        except OSError:
            pass
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    print(json.dumps(out), flush=True)


def main():
    args = _parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _launch_ranks(args)
    cx = Ctx(args)
    if args.workload == "file":
        if cx.world > 1:
            raise SystemExit("the file workload is a single-GPU measurement")
        return file_workload(cx, args)
    np, torch = cx.np, cx.torch
    world, rank = cx.world, cx.rank
    wl = WORKLOADS[args.workload]
    rows = args.rows or wl["rows"]
    if wl.get("strong"):
        rows = (rows + world - 1) // world                      # fixed total: BASELINE config 3 is a strong-scaling case
    L = args.read_len or wl["read_len"]

    check = None
    if world > 1 and not args.no_selfcheck:
        check = selfcheck(cx, args.selfcheck_rows)
    ranks_seen = cx.sum_over_ranks(1)

    gathered = world > 1 and (args.gather or args.check)
    res = timed_run(cx, wl, rows, L, args.steps, args.warmup, gather=gathered, keep=True)
    bam, rep = res["bam"], res["rep"]
    n_total = res["n_total"]

    if world == 1 and not args.no_selfcheck:
        check = n1_selfcheck(cx, wl, res, L)

    if args.check and world > 1 and rank == 0 and wl["kind"] == "cx":
        # the whole stream on rank 0 alone, same generator arguments (n_chr_for(world): the genome of the sharded run)
        whole = make_batch(type("Whole", (), dict(world=world, rank=0, local=cx.local, synth=cx.synth))(), wl, n_total, L, n_total)
        ref = cx.ea.generateCytosineReport(whole, threshold_reads=wl["threshold"], report_context=wl["report_context"], as_device=True)
        ok = all(bool(torch.equal(ref[k], rep[k])) for k in ref)
        print("CHECK sharded == single-GPU table: %s (%d rows)" % (ok, ref.nrow), flush=True)
        whole.close()
        if not ok:
            raise SystemExit("sharded result differs from the single-GPU result")
    nrow_local = res["nrow_local"]
    nrow_out = nrow_local if gathered or world == 1 else cx.sum_over_ranks(nrow_local)
    if gathered and world > 1:
        nrow_out = cx.sum_over_ranks(nrow_local if rank == 0 else 0)

    out = None
    if rank == 0:
        kname = "mhl_tiles" if wl["kind"] == "mhl" else "cx_tiles"
        rows_this_rank = nrow_local if not gathered else nrow_out // world   # rows rank 0's kernel emitted
        row_bytes = 36 if wl["kind"] == "mhl" else 24
        # L (xm) + 8 (off) + 12 (rname,strand,start) + 4 (pass) per read, + 24/36 B per output row.  The step
        # is charged all of it (step_frac); the tile kernel only what IT writes per output row (its pool rows: key, meth,
        # unmeth = 12 B for CX; key, coverage and three 64-bit sums = 32 B for lMHL) -- the table columns are written by
        # the gather kernel, whose time is in kernel_ms_all.
        in_bytes = res["nbytes_local"] + rows * (8 + 12 + 4)
        step_bytes = in_bytes + row_bytes * rows_this_rank
        alg_bytes = in_bytes + (32 if wl["kind"] == "mhl" else 12) * rows_this_rank
        kms = res["kernels"].get(kname, 0.0)
        achieved = alg_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        traffic, tsrc = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile)).get(args.workload, {})
                if tj.get("rows_per_gpu", rows) == rows and not args.read_len:
                    traffic = tj.get(kname)
                    tsrc = "static: %s, rocprofv3 --pmc passes of this workload at this size (%s)" % (
                        os.path.relpath(tfile, ROOT), tj.get("source", "profiles/"))
            except Exception:
                traffic = None
        out = {
            "metric": "Mreads/s generateCytosineReport (150 bp PE)" if wl["kind"] == "cx" and L == 300 else
                      "Mreads/s %s" % ("generateMhlReport" if wl["kind"] == "mhl" else "generateCytosineReport (long reads)"),
            "value": round(n_total * args.steps / res["dt"] / 1e6, 3),
            "unit": "Mreads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(res["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "strong" if wl.get("strong") else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "rows_per_gpu": rows, "template_bytes": L,
                       "stream": wl.get("stream", "uniform"), "n_chr": n_chr_for(world),
                       "read_unit": "template row (merged pair); mates/s = 2x", "output_rows": int(nrow_out),
                       "inputs": "resident in HBM",
                       "layout": ("rows at offsets congruent to their start position modulo %d in the engine's own copy of xm, made once per "
                                  "batch by epi_batch_realign (batch_ms below; layout_off: the step without it)" % res["layout"]) if res.get("layout")
                                 else "rows back to back as adopted",
                       "batch_ms": round(res["batch_ms"], 3),
                       "setup": "%d untimed steps (~0.3 s) before the --warmup steps, so that the timed steps see sustained clocks" % res["ramp_steps"],
                       "tile_index": ("the batch owns its columns (epi_batch_realign / epi_batch_upload): the tile table of the first report is kept, steps only reset its counters "
                                      "(tile_hint_off: EPIHIP_TILE_HINT=0, counted, scanned and filled by every step)" if res.get("layout") else
                                      "rebuilt by every step from all rows; block offsets remembered from the first call on the batch and verified block by block "
                                      "(EPIHIP_TILE_HINT=0: counted and scanned every step)"), "sharding": ("row ranges; shared tiles all-reduced (%s); output rows %s"
                                    % ("RCCL, called by the library: epi_batch_*_report_sharded" if res.get("native_comm")
                                       else args.backend + " collectives of torch.distributed around the two-step C entry points"
                                       + (", a rehearsal without RCCL" if args.backend != "nccl" else ""),
                                       "gathered to rank 0" if gathered else "stay sharded in rank order")) if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": tsrc,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": round(kms, 4),
                         "kernel_ms_all": res["kernels"], "step_algorithmic_bytes": int(step_bytes),
                         "step_frac": round(step_bytes / (res["ms_per_step"] * 1e-3) / 1e9 / 8000.0, 5)},
            "ranks_seen": ranks_seen, "backend": args.backend if world > 1 else None,
            "all_reduce_bytes_per_step": int(res["exchange_bytes"]), "selfcheck": check,
        }
    extras = not args.no_extras and not args.rows and not args.read_len
    if world == 1 and extras and out is not None:
        out["tile_hint_off"] = tile_hint_off(cx, wl, res, max(3, args.steps // 2))
        if res.get("layout"):
            out["layout_off"] = layout_off(cx, wl, res, max(3, args.steps // 2))
    if world == 1 and extras and wl["kind"] == "cx" and L <= 1000 and out is not None:
        out["streamed"], out["d2h"] = streamed_and_d2h(cx, wl, res)
        out["host_out"] = host_out(cx, wl, res)
        out["sharded_1rank"] = sharded_one_rank(cx, wl, res, max(5, args.steps))
    if world == 1 and args.cpu_sample > 0 and out is not None:
        out["cpu_baseline"] = cpu_baseline(bam, wl, min(args.cpu_sample, rows), np)
    elif out is not None:
        out["cpu_baseline"] = None                   # --cpu-sample 0 / N > 1: not timed in this run
    del rep
    if res.get("eng") is not None:
        res["eng"].close_comm()
    res.pop("rep", None); res.pop("eng", None)
    bam.close()
    del bam
    res.pop("bam", None)
    torch.cuda.empty_cache()

    if extras and args.workload == "cfg2":
        if world == 1:
            k2 = max(3, args.steps // 2)
            for name in ("cfg2g", "cfg2u", "cfg2p", "cfg2t"):     # the effects apart: start distribution / ragged + gaps / a pile-up / a tail of long templates
                u = timed_run(cx, WORKLOADS[name], rows, L, k2, 1)
                if out is not None:
                    out[name] = {"value": round(u["n_total"] * k2 / u["dt"] / 1e6, 3), "unit": "Mreads/s",
                                 "ms_per_step": round(u["ms_per_step"], 4), "vs_cfg2": round(u["ms_per_step"] / res["ms_per_step"], 3),
                                 "kernel_ms_all": u["kernels"], "what": WORKLOADS[name]["desc"]}
            # the comparator for N > 1 runs: N = 1 on THEIR stream (3 chromosomes, n_chr_for), so that a 1 -> N ratio
            # divides numbers from the same genome layout
            u = timed_run(cx, wl, rows, L, k2, 1, n_chr=n_chr_for(2))
            if out is not None:
                out["n1_same_stream"] = {"value": round(u["n_total"] * k2 / u["dt"] / 1e6, 3), "unit": "Mreads/s", "n_chr": n_chr_for(2),
                                         "ms_per_step": round(u["ms_per_step"], 4), "kernel_ms_all": u["kernels"],
                                         "what": "this workload at N = 1 on the stream the N > 1 runs use (3 chromosomes: every "
                                                 "cut of equal row ranges lies inside a chromosome); the base for a 1 -> N ratio"}
        w3 = WORKLOADS["cfg3"]
        r3 = (w3["rows"] + world - 1) // world
        s3 = timed_run(cx, w3, r3, w3["read_len"], args.strong_steps, 1)
        if out is not None:
            out["strong_cfg3"] = {"value": round(s3["n_total"] * args.strong_steps / s3["dt"] / 1e6, 3), "unit": "Mreads/s",
                                  "rows_total": s3["n_total"], "rows_per_gpu": r3, "n_gpus": world, "steps": args.strong_steps,
                                  "ms_per_step": round(s3["ms_per_step"], 4), "scaling": "strong",
                                  "all_reduce_bytes_per_step": int(s3["exchange_bytes"]), "kernel_ms_all": s3["kernels"],
                                  "what": w3["desc"]}
    if out is not None:
        print(json.dumps(out), flush=True)
    if world > 1:
        cx.dist.barrier()
        cx.dist.destroy_process_group()


def _oracle_report(args):
    """One oracle pass (module-level so that a process pool can run it)."""
    xm, off, rname, strand, start, kind, threshold, letters = args
    from oracle import oracle as orc
    t0 = time.perf_counter()
    if kind == "mhl":
        orc.mhl_report(xm, off, rname, strand, start, "Zz", 0, 0, 0.1)
    else:
        p = orc.threshold_reads(xm, off, "Z", "z", "XH", "xh", 2, 0.5, 0.1) if threshold else None
        orc.cx_report(xm, off, rname, strand, start, p, letters)
    return time.perf_counter() - t0


def cpu_baseline(bam, wl, sample, np):
    """The CPU restatement of the reference algorithm (oracle/epi_oracle.c, kind "port": same per-base ordered-map
    emplace and flush rule as src/rcpp_cx_report.cpp), one thread (the reference is single-threaded), on the first
    `sample` rows of the same synthetic stream; plus, labelled as not a reference feature, P independent oracle
    processes on disjoint row ranges of the stream ("all host cores")."""
    d = bam.dev
    letters = {"CG": "Z", "CX": "ZXH"}.get(wl.get("report_context", "CG"), "Z")

    def rows_of(lo, hi):
        off = d["off"][lo:hi + 1].cpu().numpy()
        xm = d["xm"][int(off[0]):int(off[-1])].cpu().numpy()
        return (xm, off - off[0], d["rname"][lo:hi].cpu().numpy(), d["strand"][lo:hi].cpu().numpy(),
                d["start"][lo:hi].cpu().numpy(), wl["kind"], wl.get("threshold", False), letters)

    dt = _oracle_report(rows_of(0, sample))
    out = {"value": round(sample / dt / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
           "sample": "first %d rows of the same synthetic stream, %.1f s of CPU work" % (sample, dt)}
    try:
        from concurrent.futures import ThreadPoolExecutor      # the oracle is a C library without global state and
        P = max(1, min(os.cpu_count() or 1, 16))               # ctypes drops the GIL: threads = independent oracle runs
        per = max(1, min(sample // 2, bam.n // P))
        parts = [rows_of(i * per, (i + 1) * per) for i in range(P)]
        cpu_model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        t0 = time.perf_counter()
        with ThreadPoolExecutor(P) as pool:
            list(pool.map(_oracle_report, parts))
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": round(P * per / wall / 1e6, 4), "unit": "Mreads/s", "cores": P, "cpu": cpu_model,
                            "sample": "%d concurrent oracle runs x %d rows (disjoint row ranges of the stream), %.1f s wall" % (P, per, wall),
                            "note": "not a reference feature: epialleleR runs single-threaded (vignettes/epialleleR.Rmd:160-165)"}
    except Exception as e:                                   # the single-thread figure is the baseline; this one is extra
        out["all_cores"] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    main()
