#!/usr/bin/env python3
"""Headline benchmark: Mreads/s of generateCytosineReport() on synthetic PE150
templates resident in HBM (BASELINE.json metric; config 2 at N=1, weak-scaled by
row-range shards at N>1).  One process per GPU; N>1 is launched by
torch.distributed.run and uses RCCL (backend "nccl") for the one shared-tile
all-reduce and the row gather.

    python bench.py --gpus 1 --steps 5 --warmup 2

A "step" is one full generateCytosineReport(bam) on the resident batch: the
thresholding kernel, the tile index, the LDS-histogram tile kernel with the
majority rule, and the ordered gather into the six output columns (which stay in
HBM, as the input does).  A "read" is one template row (a merged read pair).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (rows per GPU, template bytes, report)
    "cfg2": dict(rows=10_000_000, read_len=300, kind="cx", threshold=True, report_context="CG",
                 desc="simulateBam-like 10M PE150 templates (L=300, depth 30, 4 chr), generateCytosineReport defaults "
                      "(threshold.reads=TRUE, CG)"),
    "cfg2cx": dict(rows=10_000_000, read_len=300, kind="cx", threshold=False, report_context="CX",
                   desc="10M PE150 templates, generateCytosineReport(threshold.reads=FALSE, report.context='CX')"),
    "cfg3": dict(rows=100_000_000, strong=True, read_len=300, kind="cx", threshold=True, report_context="CG",
                 desc="100M PE150 templates in total (split over the GPUs), generateCytosineReport defaults"),
    "cfg4": dict(rows=50_000_000, read_len=300, kind="mhl", desc="50M PE150 templates, generateMhlReport defaults"),
    "cfg5": dict(rows=5_000_000, read_len=10000, kind="cx", threshold=False, report_context="CG",
                 desc="5M long-read (10 kb) templates, generateCytosineReport(threshold.reads=FALSE)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--read-len", type=int, default=0, help="override the template length of the workload (experiments)")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=5_000_000, help="rows timed on the CPU oracle (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for rehearsals")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--gather", action="store_true", help="N>1: also send every rank's rows to rank 0 inside the timed step")
    ap.add_argument("--check", action="store_true", help="rank 0 checks the gathered table against a single-GPU run of the whole stream")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import epialleler_amd as ea
    from epialleler_amd import _lib, distributed as D, synth
    lib = _lib.load()

    wl = WORKLOADS[args.workload]
    rows = args.rows or wl["rows"]
    if wl.get("strong"):
        rows = (rows + world - 1) // world                      # fixed total: BASELINE config 3 is a strong-scaling case
    L = args.read_len or wl["read_len"]
    n_total = rows * world
    bam = synth.generate_device(n_total=n_total, read_len=L, row_first=rank * rows, n=rows, device=local)
    torch.cuda.synchronize()

    eng = D.HipShardEngine(bam) if world > 1 else None

    def step():
        if wl["kind"] == "mhl":
            return ea.generateMhlReport(bam, as_device=True)
        if world == 1:
            return ea.generateCytosineReport(bam, threshold_reads=wl["threshold"], report_context=wl["report_context"],
                                             as_device=True)
        return D.sharded_cytosine_report(eng, threshold_reads=wl["threshold"], report_context=wl["report_context"],
                                         gather=args.gather or args.check, levels=bam.levels)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not steps: the first calls size the engine's row pool / record space for this workload and the caching
    # allocator's blocks for the two output tables that are alive at a time (rep = step() frees the previous one late)
    rep = step()
    rep = step()
    for _ in range(args.warmup):
        rep = step()
    barrier()
    lib.epi_prof_reset()
    lib.epi_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rep = step()
    barrier()
    dt = time.perf_counter() - t0
    lib.epi_prof_enable(0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # dominant kernel, timed with HIP events on the stream it is launched on (library hook)
    import ctypes as C
    kname = b"mhl_tiles" if wl["kind"] == "mhl" else b"cx_tiles"
    ms, cnt = C.c_double(0), C.c_int64(0)
    lib.epi_prof_get(kname, C.byref(ms), C.byref(cnt))
    kernels = {}
    for nm in (b"threshold", b"cx_tiles", b"mhl_rows", b"mhl_tiles"):
        m2, c2 = C.c_double(0), C.c_int64(0)
        lib.epi_prof_get(nm, C.byref(m2), C.byref(c2))
        if c2.value:
            kernels[nm.decode()] = round(m2.value / c2.value, 4)

    if args.check and world > 1 and rank == 0 and wl["kind"] == "cx":
        whole = synth.generate_device(n_total=n_total, read_len=L, device=local)
        ref = ea.generateCytosineReport(whole, threshold_reads=wl["threshold"], report_context=wl["report_context"], as_device=True)
        ok = all(bool(torch.equal(ref[k], rep[k])) for k in ref)
        print("CHECK sharded == single-GPU table: %s (%d rows)" % (ok, ref.nrow), flush=True)
        whole.close()
        if not ok:
            raise SystemExit("sharded result differs from the single-GPU result")
    gathered = world > 1 and (args.gather or args.check)
    nrow_local = rep.nrow if rep is not None else 0
    nrow_out = nrow_local
    if world > 1 and not gathered:
        tot = torch.tensor([nrow_local], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        nrow_out = int(tot.item())
    if rank == 0:
        rows_this_rank = nrow_local if not gathered else nrow_out // world   # rows rank 0's kernel emitted
        row_bytes = 36 if wl["kind"] == "mhl" else 24
        # L (xm) + 8 (off) + 12 (rname,strand,start) + 4 (pass) per read, + 24/36 B per output row
        alg_bytes = rows * (L + 8 + 12 + 4) + row_bytes * rows_this_rank
        kms = ms.value / max(cnt.value, 1)
        achieved = alg_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(args.workload, {}).get(kname.decode())
            except Exception:
                traffic = None
        out = {
            "metric": "Mreads/s generateCytosineReport (150 bp PE)" if wl["kind"] == "cx" and L == 300 else
                      "Mreads/s %s" % ("generateMhlReport" if wl["kind"] == "mhl" else "generateCytosineReport (long reads)"),
            "value": round(n_total * args.steps / dt / 1e6, 3),
            "unit": "Mreads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if wl.get("strong") else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "rows_per_gpu": rows, "template_bytes": L,
                       "read_unit": "template row (merged pair); mates/s = 2x", "output_rows": int(nrow_out),
                       "inputs": "resident in HBM", "sharding": ("row ranges; shared tiles all-reduced (RCCL); output rows %s"
                                    % ("gathered to rank 0" if gathered else "stay sharded in rank order")) if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": kname.decode(), "achieved": round(achieved, 2), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": round(kms, 4),
                         "kernel_ms_all": kernels},
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(bam, wl, min(args.cpu_sample, rows), np)
        else:
            out["cpu_baseline"] = None                   # --cpu-sample 0 / N > 1: not timed in this run
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(bam, wl, sample, np):
    """The CPU restatement of the reference algorithm (oracle/epi_oracle.c, kind "port": same per-base ordered-map
    emplace and flush rule as src/rcpp_cx_report.cpp), one thread (the reference is single-threaded), on the first
    `sample` rows of the same synthetic stream."""
    from oracle import oracle as orc
    d = bam.dev
    off = d["off"][:sample + 1].cpu().numpy()
    xm = d["xm"][:int(off[-1])].cpu().numpy()
    rname, strand, start = (d[k][:sample].cpu().numpy() for k in ("rname", "strand", "start"))
    c = {"CG": ("Z", "z", "XH", "xh"), "CX": ("ZXH", "zxh", "", "")}
    t0 = time.perf_counter()
    if wl["kind"] == "mhl":
        orc.mhl_report(xm, off, rname, strand, start, "Zz", 0, 0, 0.1)
    else:
        p = None
        if wl["threshold"]:
            p = orc.threshold_reads(xm, off, *c["CG"], 2, 0.5, 0.1)
        letters = c[wl["report_context"]][0]
        orc.cx_report(xm, off, rname, strand, start, p, letters)
    dt = time.perf_counter() - t0
    return {"value": round(sample / dt / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": "first %d rows of the same synthetic stream, %.1f s of CPU work" % (sample, dt)}


if __name__ == "__main__":
    main()
