"""numpy stand-in for epialleler_amd.distributed.HipShardEngine -- a TEST DOUBLE so
that the row-range sharding / shared-tile exchange logic (product code in
epialleler_amd/distributed.py) runs under gloo on CPU.  Same interface, CPU
tensors.  Not a product path."""
import numpy as np
import torch

from oracle import oracle as orc

T = 1024
BIAS = 1 << 31
# nibble -> slot, as in epialleler_amd/csrc/common.hpp / cx_report.hip
SLOT = np.asarray([1, 1, 2, 1, 1, 1, 4, 6, 1, 1, 3, 8, 0, 1, 5, 7], np.int64)


class NumpyShardEngine:
    def __init__(self, t):
        self.t = t
        self.device = torch.device("cpu")
        self.n = t["off"].size - 1

    def tile_positions(self, ctx="Z"):
        return T

    def key_range(self, kind="cx", ctx="Z"):
        if self.n == 0:
            return 0, -1
        t = self.t
        lmax = max(int(np.diff(t["off"]).max()), 1)
        first = (int(t["rname"][0]) << 32) | ((int(t["start"][0]) + BIAS) // T)
        last = (int(t["rname"][-1]) << 32) | ((int(t["start"][-1]) + lmax - 1 + BIAS) // T)
        return first, last

    def threshold(self, cm, cu, om, ou, min_n, min_frac, max_oo):
        return torch.from_numpy(orc.threshold_reads(self.t["xm"], self.t["off"], cm, cu, om, ou, min_n, min_frac, max_oo))

    def cx_accumulate(self, pass_, ctx, keys, owned):
        t = self.t
        lens = np.diff(t["off"])
        row = np.repeat(np.arange(self.n), lens)
        idx = np.arange(t["xm"].size) - t["off"][:-1][row]
        code = (t["xm"] & 15).astype(np.int64)
        if pass_ is not None:
            p = pass_.numpy() if hasattr(pass_, "numpy") else np.asarray(pass_)
            code = code | np.where(p[row] == 0, 8, 0)
        slot = SLOT[code]
        keep = slot != 8
        inc = np.where(code == 9, 2, 1)[keep]
        pos = (t["start"].astype(np.int64)[row] + idx)[keep]
        rn = t["rname"].astype(np.int64)[row][keep]
        plane = ((t["strand"].astype(np.int64)[row] - 1) * 8 + slot)[keep]
        key = (rn << 32) | ((pos + BIAS) // T)
        self.keys, self.owned = np.asarray(keys, np.int64), np.asarray(owned, np.int32)
        slab = np.zeros((max(self.keys.size, 1), 16, T), np.int32)
        si = np.searchsorted(self.keys, key)
        si_c = np.minimum(si, max(self.keys.size - 1, 0))
        shared = (self.keys[si_c] == key) if self.keys.size else np.zeros(key.size, bool)
        np.add.at(slab, (si_c[shared], plane[shared], ((pos + BIAS) % T)[shared]), inc[shared].astype(np.int32))
        # local (exclusive) counters, dense per (rname,pos)
        lk = (rn[~shared] << 33) | (pos[~shared] + BIAS)
        uk, inv = np.unique(lk, return_inverse=True)
        cnt = np.zeros((uk.size, 16), np.int64)
        np.add.at(cnt, (inv, plane[~shared]), inc[~shared])
        self.local = (uk >> 33, (uk & ((1 << 33) - 1)) - BIAS, cnt)
        self.slab = torch.from_numpy(slab.reshape(-1))
        return self.slab

    def cx_finish(self, ctx):
        ctx_ok = {((ord(ch) + 2) >> 2) & 15 for ch in ctx}
        rn, pos, cnt = self.local
        slab = self.slab.numpy().reshape(-1, 16, T)
        for s in np.nonzero(self.owned)[0]:
            k = int(self.keys[s])
            pp = ((k & 0xFFFFFFFF) * T - BIAS) + np.arange(T)
            rn = np.concatenate([rn, np.full(T, k >> 32)])
            pos = np.concatenate([pos, pp])
            cnt = np.concatenate([cnt, slab[s].T.astype(np.int64)])
        order = np.lexsort((pos, rn))
        rn, pos, cnt = rn[order], pos[order], cnt[order]
        rows = []
        for s in (0, 1):
            c = cnt[:, s * 8:(s + 1) * 8]
            cov = c.sum(1)
            half = cov // 2
            nH, nX, nZ = c[:, 2] + c[:, 3], c[:, 4] + c[:, 5], c[:, 6] + c[:, 7]
            k = np.where(cov == 0, 0, np.where(c[:, 0] > half, 0, np.where(nH > half, 2, np.where(nX > half, 6, np.where(nZ > half, 7, 0)))))
            meth = np.where(k == 2, c[:, 2], np.where(k == 6, c[:, 4], c[:, 6]))
            unmeth = np.where(k == 2, c[:, 3], np.where(k == 6, c[:, 5], c[:, 7]))
            ok = np.isin(k, list(ctx_ok)) & (k != 0)
            rows.append(np.stack([rn[ok], np.full(ok.sum(), s + 1), pos[ok], k[ok], meth[ok], unmeth[ok]]))
        allr = np.concatenate(rows, axis=1) if rows else np.zeros((6, 0), np.int64)
        o = np.lexsort((allr[1], allr[2], allr[0]))
        return torch.from_numpy(allr[:, o].astype(np.int32))
