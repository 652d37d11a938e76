"""world_size 2..4 gloo runs (CPU) of the product's sharding / shared-tile exchange
logic (epialleler_amd/distributed.py) with a numpy engine as test double; the
concatenated result must equal the oracle on the unsharded input."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
import synth_np
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(t, lo, hi):
    off = t["off"][lo:hi + 1]
    return {"xm": t["xm"][int(off[0]):int(off[-1])], "off": off - off[0], "rname": t["rname"][lo:hi],
            "strand": t["strand"][lo:hi], "start": t["start"][lo:hi]}


def _worker(rank, world, port, case, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from epialleler_amd import distributed as D
    from fake_engine import NumpyShardEngine
    t = _cases()[case]
    n = t["off"].size - 1
    cuts = [n * r // world for r in range(world + 1)]
    if case == "uneven":
        cuts = [0] + [min(n, 5 + 3 * r) for r in range(1, world)] + [n]
    eng = NumpyShardEngine(_shard(t, cuts[rank], cuts[rank + 1]))
    for thr, rctx in ((True, "CG"), (False, "CX")):
        rep = D.sharded_cytosine_report(eng, threshold_reads=thr, report_context=rctx)
        if rank == 0:
            np.savez(os.path.join(outdir, "%s_%d_%s.npz" % (case, world, rctx)), **{k: v.numpy() for k, v in rep.items()})
        else:
            assert rep is None
    dist.destroy_process_group()


def _cases():
    rng = np.random.default_rng(101)
    wgs = synth_np.generate(n_total=1200, read_len=300, n_chr=2)
    amp = synth_np.random_templates(rng, 900, 100, 400, 1, 30)           # every rank shares the same tiles
    mixed = synth_np.random_templates(rng, 700, 0, 2500, 3, 6000)        # long reads: multi-tile halos, 3 rnames
    return {"wgs": wgs, "amplicon": amp, "mixed": mixed, "uneven": mixed}


@pytest.mark.parametrize("world,case", [(2, "wgs"), (2, "amplicon"), (2, "mixed"), (3, "mixed"), (4, "amplicon"), (3, "uneven")])
def test_sharded_report_equals_oracle(tmp_path, world, case):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    t = _cases()[case]
    c = H.CONTEXT_TO_BASES["CG"]
    p = orc.threshold_reads(t["xm"], t["off"], c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"], 2, 0.5, 0.1)
    for thr, rctx, letters in ((True, "CG", "Z"), (False, "CX", "ZXH")):
        want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p if thr else None, letters)
        got = dict(np.load(os.path.join(str(tmp_path), "%s_%d_%s.npz" % (case, world, rctx))))
        H.assert_reports_equal(got, want)


def test_shared_tile_keys_logic():
    from epialleler_amd.distributed import shared_tile_keys
    K = lambda r, t: (r << 32) | t
    keys, owner = shared_tile_keys([(K(1, 10), K(1, 20)), (K(1, 12), K(1, 14)), (K(1, 14), K(1, 30)), (0, -1)])
    assert keys.tolist() == [K(1, t) for t in range(12, 21)] and owner.tolist() == [0] * 9
    keys, owner = shared_tile_keys([(K(1, 5), K(1, 6)), (K(1, 6), K(2, 3)), (K(2, 3), K(2, 9))])
    assert keys.tolist() == [K(1, 6), K(2, 3)] and owner.tolist() == [0, 1]
    keys, owner = shared_tile_keys([(K(1, 5), K(1, 6)), (K(1, 7), K(1, 9))])
    assert keys.size == 0
    with pytest.raises(ValueError):
        shared_tile_keys([(K(1, 5), K(2, 6)), (K(1, 7), K(2, 9))])


def test_library_shared_tile_keys_match_the_python_rendering():
    """csrc/comm.hip derives the shared tiles of a sharded report itself (epi_shared_tile_keys is that step on its own): the
    same keys and owners as distributed.shared_tile_keys on random rank layouts -- contiguous shards with halos, ranks
    without rows, amplicon-like total overlap, cuts at reference-sequence boundaries."""
    import ctypes as C
    import numpy as np
    from epialleler_amd import _lib
    from epialleler_amd.distributed import shared_tile_keys
    lib = _lib.load()
    rng = np.random.default_rng(5)
    cases = []
    for _ in range(300):
        world = int(rng.integers(1, 9))
        ranges, pos, rname = [], int(rng.integers(0, 50)), int(rng.integers(1, 4))
        for r in range(world):
            if rng.random() < 0.15:
                ranges.append((0, -1))                                  # a rank without rows
                continue
            if rng.random() < 0.2:
                rname += 1; pos = int(rng.integers(0, 50))              # the shard starts on the next reference sequence
            first = (rname << 32) | max(pos - int(rng.integers(0, 3)), 0)   # halo: reaches back into the previous rank's tiles
            last = (rname << 32) | (pos + int(rng.integers(0, 40)))
            ranges.append((first, last))
            pos = (last & 0xFFFFFFFF) + int(rng.integers(0, 2))
        cases.append(ranges)
    cases.append([((1 << 32) | 5, (1 << 32) | 9)] * 4)                  # every rank reaches every tile
    for ranges in cases:
        want_k, want_o = shared_tile_keys(ranges)
        flat = np.asarray(ranges, np.int64).reshape(-1)
        n = C.c_int32(0)
        _lib.check(lib.epi_shared_tile_keys(C.c_void_p(flat.ctypes.data), len(ranges), None, None, 0, C.byref(n)))
        assert n.value == want_k.size
        keys, owner = np.zeros(max(n.value, 1), np.int64), np.zeros(max(n.value, 1), np.int32)
        _lib.check(lib.epi_shared_tile_keys(C.c_void_p(flat.ctypes.data), len(ranges), C.c_void_p(keys.ctypes.data),
                                            C.c_void_p(owner.ctypes.data), int(keys.size), C.byref(n)))
        assert np.array_equal(keys[:n.value], want_k) and np.array_equal(owner[:n.value], want_o)
