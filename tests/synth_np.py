"""numpy mirror of the device generator epialleler_amd/csrc/synth.hip (same
counter-based hashes, same model) -- test infrastructure.  Used for CPU-side
seeded inputs and to check that the device generator produces exactly these
bytes."""
import numpy as np

U64 = np.uint64
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(z):
    z = (z + U64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)) & _M
    return z ^ (z >> U64(31))


def hash3(seed, stream, idx):
    with np.errstate(over="ignore"):
        s = mix64(np.asarray(U64(seed) + U64(stream) * U64(0xD1B54A32D192ED03), dtype=U64))
        return mix64(s ^ np.asarray(idx, dtype=U64))


def generate(n_total, read_len=300, n_chr=4, depth=30, seed=42, row_first=0, n=None, gap_from=0, gap_len=0):
    """Rows [row_first, row_first+n) of the global sorted stream.  Returns SoA dict."""
    n = n_total - row_first if n is None else n
    L = read_len
    rows_per_chr = (n_total + n_chr - 1) // n_chr
    chr_len = max(rows_per_chr * L // depth, 1)
    stride = max(chr_len // rows_per_chr, 1)
    with np.errstate(over="ignore"):
        x = np.arange(row_first, row_first + n, dtype=np.int64)
        c = x // rows_per_chr
        j = x - c * rows_per_chr
        start = (1 + (j * chr_len) // rows_per_chr + (hash3(seed, 1, x.astype(U64)) % U64(stride)).astype(np.int64)).astype(np.int32)
        strand = (1 + (hash3(seed, 2, x.astype(U64)) & U64(1)).astype(np.int64)).astype(np.int32)
        hyper = (hash3(seed, 3, x.astype(U64)) % U64(10)) == U64(0)
        i = np.arange(L, dtype=np.int64)[None, :]
        pos = (start.astype(np.int64)[:, None] + i).astype(U64)
        # per-chromosome stream 16+c
        sd = mix64((U64(seed) + (U64(16) + c.astype(U64)) * U64(0xD1B54A32D192ED03)) & _M)
        t = (mix64(sd[:, None] ^ pos) % U64(1000)).astype(np.int64)
        code = np.full((n, L), 12, np.int64)
        code[(t >= 760) & (t < 895)] = 10
        code[(t >= 895) & (t < 955)] = 14
        code[(t >= 955) & (t < 990)] = 15
        code[t >= 990] = 13
        cyt = t >= 760
        v = hash3(seed, 4, (x.astype(U64)[:, None] << U64(16)) + i.astype(U64))
        noise = cyt & ((v % U64(100)) == U64(0))
        nxt = np.where(code == 10, 14, np.where(code == 14, 15, 10))
        code = np.where(noise, nxt, code)
        thr = np.where(code == 15, np.where(hyper[:, None], 900, 50), 10)
        meth = cyt & (((v >> U64(20)) % U64(1000)).astype(np.int64) < thr)
        code = np.where(meth, code - 8, code)
        byte = (0x10 | code).astype(np.uint8)
        if gap_len > 0:
            byte[:, gap_from:gap_from + gap_len] = 0xFB
    off = np.arange(n + 1, dtype=np.int64) * L
    return {"xm": byte.reshape(-1), "off": off, "rname": (c + 1).astype(np.int32), "strand": strand, "start": start}


def random_templates(rng, n, min_len=0, max_len=400, n_rname=3, span=5000, p_garbage=0.0, alphabet=None):
    """Ragged random templates (not the bench model): arbitrary lengths, any XM letters, optional raw
    garbage bytes; sorted by (rname,start)."""
    lens = rng.integers(min_len, max_len + 1, size=n)
    rname = np.sort(rng.integers(1, n_rname + 1, size=n)).astype(np.int32)
    start = rng.integers(1, span, size=n).astype(np.int32)
    order = np.lexsort((start, rname))
    rname, start = rname[order], start[order]
    strand = rng.integers(1, 3, size=n).astype(np.int32)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    nb = int(off[-1])
    if alphabet is None:
        alphabet = ".......hhxzZHXuU+-"
    letters = np.frombuffer(alphabet.encode("latin1"), np.uint8)
    ch = letters[rng.integers(0, letters.size, size=nb)].astype(np.int64)
    xm = (((rng.integers(0, 16, size=nb) << 4) | (((ch + 2) >> 2) & 15))).astype(np.uint8)
    if p_garbage > 0:
        g = rng.random(nb) < p_garbage
        xm[g] = rng.integers(0, 256, size=int(g.sum())).astype(np.uint8)
    return {"xm": xm, "off": off, "rname": rname, "strand": strand, "start": start}


def generate_uniform(n_total, mean_len=300, n_chr=4, depth=30, seed=42, row_first=0, n=None, gap_every=4, gap_len=50, ragged=True, pileup=None):
    """numpy mirror of epialleler_amd.synth.generate_device_uniform (layout by the same torch code on the CPU, bytes by
    the same hashes as epi_synth_fill_dev)."""
    from epialleler_amd import synth
    n = n_total - row_first if n is None else n
    rname, start, lens = (t.numpy() for t in synth.uniform_layout(n_total, mean_len, n_chr, depth, seed, row_first, n, "cpu", ragged, pileup))
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    nb = int(off[-1])
    with np.errstate(over="ignore"):
        row = np.repeat(np.arange(n, dtype=np.int64), lens)
        i = np.arange(nb, dtype=np.int64) - off[row]
        x = (row + row_first).astype(U64)
        c = (rname[row] - 1).astype(U64)
        pos = (start[row].astype(np.int64) + i).astype(U64)
        sd = mix64((U64(seed) + (U64(16) + c) * U64(0xD1B54A32D192ED03)) & _M)
        t = (mix64(sd ^ pos) % U64(1000)).astype(np.int64)
        code = np.full(nb, 12, np.int64)
        code[(t >= 760) & (t < 895)] = 10
        code[(t >= 895) & (t < 955)] = 14
        code[(t >= 955) & (t < 990)] = 15
        code[t >= 990] = 13
        cyt = t >= 760
        v = hash3(seed, 4, (x << U64(16)) + i.astype(U64))
        noise = cyt & ((v % U64(100)) == U64(0))
        nxt = np.where(code == 10, 14, np.where(code == 14, 15, 10))
        code = np.where(noise, nxt, code)
        hyper = (hash3(seed, 3, x) % U64(10)) == U64(0)
        thr = np.where(code == 15, np.where(hyper, 900, 50), 10)
        meth = cyt & (((v >> U64(20)) % U64(1000)).astype(np.int64) < thr)
        code = np.where(meth, code - 8, code)
        byte = (0x10 | code).astype(np.uint8)
        if gap_every > 0 and gap_len > 0:
            gapped = (hash3(seed, 5, x) % U64(gap_every)) == U64(0)
            g0 = lens[row] // 2 - gap_len // 2
            byte[gapped & (i >= g0) & (i < g0 + gap_len)] = 0xFB
        xs = np.arange(row_first, row_first + n, dtype=np.int64).astype(U64)
        strand = (1 + (hash3(seed, 2, xs) & U64(1)).astype(np.int64)).astype(np.int32)
    return {"xm": byte, "off": off, "rname": rname.astype(np.int32), "strand": strand, "start": start.astype(np.int32)}


def with_long_tail(t, every, long_len, first=0):
    """Every `every`-th row (from `first`) stretched to `long_len` bytes by repeating its own bytes: the bulk of the rows keeps its
    length, the batch's longest row becomes long_len -- the shape of a paired-end library's insert-size tail."""
    n = len(t["start"])
    lens = np.diff(t["off"]).astype(np.int64)
    rows = [t["xm"][t["off"][x]:t["off"][x + 1]] for x in range(n)]
    for x in range(first, n, every):
        if lens[x] > 0:
            rows[x] = np.tile(rows[x], long_len // int(lens[x]) + 1)[:long_len]
    out = dict(t)
    out["off"] = np.concatenate(([0], np.cumsum([len(r) for r in rows]))).astype(np.int64)
    out["xm"] = np.concatenate(rows).astype(np.uint8) if n else t["xm"]
    return out
