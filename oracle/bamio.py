"""BAM fixture decoding + restatement of the reference's short-read producer.

TEST INFRASTRUCTURE ONLY (see oracle/epi_oracle.c header).  Pure Python/numpy:
BGZF is a series of gzip members, so `gzip` yields the raw BAM stream; the BAM
record layout is fixed (SAM spec §4.2).  The packer follows
  src/rcpp_read_bam.cpp:19-192  (rcpp_read_bam_paired)
  src/rcpp_read_bam.cpp:199-343 (rcpp_read_bam_single)
and the R wrappers R/internal.R:75-128 (.checkBam), 154-199 (.readBam),
R/preprocessBam.R:197-237 (defaults).
"""
import gzip
import struct

import numpy as np

FILLER = 0xFB  # (N, '-'): "no information", rcpp_read_bam.cpp:58


def ctx_to_idx(c):
    """src/epialleleR.h:28 -- works on ints or uint8 arrays."""
    return ((np.asarray(c, dtype=np.int64) + 2) >> 2) & 15


class BamRecord:
    __slots__ = ("tid", "pos", "mapq", "flag", "mtid", "mpos", "isize", "qname",
                 "cigar", "seq", "qual", "tags")


def _parse_aux(buf):
    tags = {}
    p = 0
    n = len(buf)
    size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    while p + 3 <= n:
        tag = buf[p:p + 2].decode("latin1")
        typ = chr(buf[p + 2])
        p += 3
        if typ in ("Z", "H"):
            e = buf.index(b"\0", p)
            tags[tag] = (typ, bytes(buf[p:e]))
            p = e + 1
        elif typ == "B":
            sub = chr(buf[p])
            cnt = struct.unpack_from("<i", buf, p + 1)[0]
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
            tags[tag] = (typ, (sub, struct.unpack_from("<%d%s" % (cnt, fmt), buf, p + 5)))
            p += 5 + cnt * size[sub]
        else:
            tags[tag] = (typ, bytes(buf[p:p + size[typ]]))
            p += size[typ]
    return tags


def read_bam_records(path):
    """Returns (target_names, [BamRecord...])."""
    with gzip.open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"BAM\1", "not a BAM file"
    l_text = struct.unpack_from("<i", data, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    names = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]
        names.append(data[p + 4:p + 4 + l_name - 1].decode("latin1"))
        p += 4 + l_name + 4
    recs = []
    n = len(data)
    while p + 4 <= n:
        block_size = struct.unpack_from("<i", data, p)[0]
        q = p + 4
        (tid, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, mtid, mpos, isize) = \
            struct.unpack_from("<iiBBHHHiiii", data, q)
        q += 32
        r = BamRecord()
        r.tid, r.pos, r.mapq, r.flag, r.mtid, r.mpos, r.isize = tid, pos, mapq, flag, mtid, mpos, isize
        r.qname = data[q:q + l_read_name - 1]
        q += l_read_name
        r.cigar = np.frombuffer(data, dtype="<u4", count=n_cigar, offset=q)
        q += 4 * n_cigar
        r.seq = np.frombuffer(data, dtype=np.uint8, count=(l_seq + 1) // 2, offset=q)
        q += (l_seq + 1) // 2
        r.qual = np.frombuffer(data, dtype=np.uint8, count=l_seq, offset=q)
        q += l_seq
        r.tags = _parse_aux(data[q:p + 4 + block_size])
        recs.append(r)
        p += 4 + block_size
    return names, recs


def check_bam(recs):
    """src/rcpp_check_bam.cpp:40-50 + R/internal.R:82-86 over the first 1024 records."""
    nrecs = npaired = ntempls = 0
    prev = None
    for r in recs[:1024]:
        nrecs += 1
        if r.flag & 0x2:
            npaired += 1
        if prev is not None and prev == r.qname:
            ntempls += 1
        prev = r.qname
    paired = npaired > nrecs / 2
    sorted_ = (ntempls > 0) and (ntempls >= nrecs // 2 or ntempls >= npaired // 2)
    return {"nrecs": nrecs, "npaired": npaired, "ntempls": ntempls, "paired": paired, "sorted": sorted_}


def _seqi_shifted(seq, idx):
    """src/epialleleR.h:32 bam_seqi_shifted, vectorised over query indices."""
    b = seq[idx >> 1].astype(np.uint16)
    return ((b << ((idx & 1) << 2)) & 0xF0).astype(np.uint8)


def _pack_paired(recs, min_mapq, min__baseq, skip_flags, trim5, trim3):
    """rcpp_read_bam_paired, src/rcpp_read_bam.cpp:19-192."""
    min_baseq = (min__baseq - (1 if min__baseq > 0 else 0)) & 0xFF      # :30, stored as uint8 (:57)
    max_w = 8192
    tq = np.full(max_w, min_baseq, np.uint8)                             # :57
    ts = np.full(max_w, FILLER, np.uint8)                                # :58
    rname, strand, start, seqs = [], [], [], []
    t_qname = None
    t_rname = t_start = t_strand = t_width = 0

    def push():                                                          # :61-69
        nonlocal tq, ts
        rname.append(t_rname + 1)
        strand.append(t_strand)
        start.append(t_start + trim5 + 1)
        seqs.append(ts[trim5:trim5 + max(t_width - (trim5 + trim3), 0)].copy())
        tq[:t_width] = min_baseq
        ts[:t_width] = FILLER

    nrecs = 0
    for r in recs:
        nrecs += 1
        if (r.flag & skip_flags) or not (r.flag & 0x2) or r.mapq < min_mapq:   # :76-78
            continue
        xg = r.tags.get("XG")
        xmt = r.tags.get("XM")
        if xg is None or xmt is None:                                    # :80-82
            continue
        if t_qname != r.qname:                                           # :85
            if t_strand != 0:
                push()                                                   # :87
            t_qname = r.qname
            t_rname = r.tid
            t_start = min(r.pos, r.mpos)                                 # :92-93
            t_width = abs(r.isize)                                       # :94
            t_strand = 2 - (1 if xg[1][:1] == b"C" else 0)               # :95
            if t_width > max_w:                                          # :98-105
                max_w = t_width
                tq = np.concatenate([tq, np.full(max_w - tq.size, min_baseq, np.uint8)])
                ts = np.concatenate([ts, np.full(max_w - ts.size, FILLER, np.uint8)])
        xm = np.frombuffer(xmt[1], dtype=np.uint8)
        qpos = 0
        dpos = r.pos - t_start                                           # :118
        for c in r.cigar:                                                # :119-150
            op = int(c) & 0xF
            ln = int(c) >> 4
            if op in (0, 7, 8):
                need = dpos + ln
                if need > tq.size:   # the reference would write past its buffer; grow instead
                    tq = np.concatenate([tq, np.full(need - tq.size, min_baseq, np.uint8)])
                    ts = np.concatenate([ts, np.full(need - ts.size, FILLER, np.uint8)])
                qi = np.arange(qpos, qpos + ln)
                better = r.qual[qi] > tq[dpos:dpos + ln]                 # :127 strictly higher
                val = _seqi_shifted(r.seq, qi) | ctx_to_idx(xm[qi]).astype(np.uint8)   # :129
                sl_q = tq[dpos:dpos + ln]
                sl_s = ts[dpos:dpos + ln]
                sl_q[better] = r.qual[qi][better]
                sl_s[better] = val[better]
                qpos += ln
                dpos += ln
            elif op in (1, 4):
                qpos += ln
            elif op in (2, 3):
                dpos += ln
            elif op in (5, 6, 9):
                pass
            else:
                raise ValueError("Unknown CIGAR operation")
        if t_width < dpos:                                               # :151
            t_width = dpos
    push()                                                               # :155
    return rname, strand, start, seqs, nrecs


def _pack_single(recs, min_mapq, min_baseq, skip_flags, trim5, trim3):
    """rcpp_read_bam_single, src/rcpp_read_bam.cpp:199-343."""
    rname, strand, start, seqs = [], [], [], []
    nrecs = 0
    for r in recs:
        nrecs += 1
        if (r.flag & skip_flags) or r.mapq < min_mapq:                   # :240-241
            continue
        xg = r.tags.get("XG")
        xmt = r.tags.get("XM")
        if xg is None or xmt is None:                                    # :243-245
            continue
        xm = np.frombuffer(xmt[1], dtype=np.uint8)
        width = sum((int(c) >> 4) for c in r.cigar if (int(c) & 0xF) in (0, 2, 3, 7, 8))   # bam_cigar2rlen :255
        buf = np.full(width, FILLER, np.uint8)                           # :265
        qpos = dpos = 0
        for c in r.cigar:                                                # :270-300
            op = int(c) & 0xF
            ln = int(c) >> 4
            if op in (0, 7, 8):
                qi = np.arange(qpos, qpos + ln)
                ok = r.qual[qi] >= min_baseq                             # :278
                val = _seqi_shifted(r.seq, qi) | ctx_to_idx(xm[qi]).astype(np.uint8)
                sl = buf[dpos:dpos + ln]
                sl[ok] = val[ok]
                qpos += ln
                dpos += ln
            elif op in (1, 4):
                qpos += ln
            elif op in (2, 3):
                dpos += ln
            elif op in (5, 6, 9):
                pass
            else:
                raise ValueError("Unknown CIGAR operation")
        rname.append(r.tid + 1)                                          # :303
        strand.append(1 if xg[1][:1] == b"C" else 2)                     # :304
        start.append(r.pos + trim5 + 1)                                  # :305
        seqs.append(buf[trim5:trim5 + max(dpos - (trim5 + trim3), 0)].copy())   # :306
    return rname, strand, start, seqs, nrecs



# ---- long-read (MM/ML) records: rcpp_read_bam_mm_single, src/rcpp_read_bam.cpp:364-579 --------------------------
# HTSlib (bam_parse_basemod / bam_next_basemod) is a dependency outside the reference tree; the tag rules restated
# here are those of the SAM tags specification, section 1.7 (see csrc/bam_pack.cpp for the summary).

NT16_STR = "=ACMGRSVTWYHKDBN"
_NT16 = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "N": 15}
_COMP = {1: 8, 8: 1, 2: 4, 4: 2}


def parse_basemods(codes4, flag, mm, ml):
    """[(query position, modification code, strand 0/1, probability or -1)] in MM order.
    codes4: 4-bit base codes of SEQ as stored; a ChEBI number n is reported as -n."""
    hits = []
    rev = bool(flag & 16)
    L = len(codes4)
    k = 0
    for entry in mm.split(";"):
        if not entry:
            continue
        base = _NT16.get(entry[0])
        if base is None or len(entry) < 3 or entry[1] not in "+-":
            break
        strand = 1 if entry[1] == "-" else 0
        head, _, tail = entry[2:].partition(",")
        head = head.rstrip(".?")
        if not head:
            break
        codes = [-int(head)] if head[0].isdigit() else [ord(ch) for ch in head]
        target = _COMP.get(base, base) if rev else base
        order = range(L - 1, -1, -1) if rev else range(L)
        typed = [i for i in order if target == 15 or codes4[i] == target]     # bases of the canonical type, as sequenced
        at = 0
        bad = False
        for d in (tail.split(",") if tail else []):
            at += int(d)
            if at >= len(typed):
                bad = True
                break
            for c in codes:
                hits.append((typed[at], c, strand, (ml[k] if k < len(ml) else -1) if ml is not None else -1))
                k += 1
            at += 1
        if bad:
            break
    return hits


def _tri_ok(c):
    return c in (1, 3, 4, 6, 7)


def _ctx_forward(b0, b1, b2):
    if b0 != 3 or not _tri_ok(b1) or not _tri_ok(b2):
        return "."
    return "z" if b1 == 7 else ("x" if b2 == 7 else "h")


def _ctx_reverse(b0, b1, b2):
    if b2 != 7 or not _tri_ok(b0) or not _tri_ok(b1):
        return "."
    return "z" if b1 == 3 else ("x" if b0 == 3 else "h")


def _pack_mm_single(recs, min_mapq, min_baseq, min_prob, highest_prob, skip_flags, trim5, trim3):
    rname, strand, start, seqs = [], [], [], []
    nrecs = 0
    for r in recs:
        nrecs += 1
        if (r.flag & skip_flags) or r.mapq < min_mapq:                   # :423-424
            continue
        rstrand = 1 if (r.flag & 16) else 0                              # :426
        L = r.qual.size
        idx = np.arange(L)
        codes4 = ((r.seq[idx >> 1] >> ((~idx & 1) << 2)) & 0xF).astype(np.int64) if L else np.zeros(0, np.int64)
        seq = "NN" + "".join(NT16_STR[c] for c in codes4) + "NN"        # :457-461
        lo = [ord(ch) & 7 for ch in seq]
        xm = [[_ctx_forward(lo[i + 2], lo[i + 3], lo[i + 4]) for i in range(L)],      # :464-467
              [_ctx_reverse(lo[i], lo[i + 1], lo[i + 2]) for i in range(L)]]
        has_mods = [False, False]
        mmt = r.tags.get("MM") or r.tags.get("Mm")
        if mmt is not None and mmt[0] == "Z":
            mlt = r.tags.get("ML") or r.tags.get("Ml")
            ml = list(mlt[1][1]) if (mlt is not None and mlt[0] == "B" and mlt[1][0] in "Cc") else None
            hits = parse_basemods(list(codes4), r.flag, mmt[1].decode("latin1"), ml)
            bypos = {}
            for h in hits:
                bypos.setdefault(h[0], []).append(h)
            for pos in sorted(bypos):                                    # :469-491
                ismeth, prob, other = [0, 0], [-2, -2], [-2, -2]
                for (_p, code, st, q) in bypos[pos]:
                    if code == ord("m") or code == -27551:
                        ismeth[st] = 1
                        prob[st] = q
                    elif other[st] < q:
                        other[st] = q
                for st in (0, 1):
                    cs = abs(rstrand - st)
                    if ismeth[st] and prob[st] >= min_prob and (not highest_prob or prob[st] > other[st]) and xm[cs][pos] > "A":
                        xm[cs][pos] = xm[cs][pos].upper()
                        has_mods[cs] = True
        width = sum((int(c) >> 4) for c in r.cigar if (int(c) & 0xF) in (0, 2, 3, 7, 8))
        rs = [np.full(width, FILLER, np.uint8), np.full(width, FILLER, np.uint8)]
        qpos = dpos = 0
        for c in r.cigar:                                                # :494-531
            op, ln = int(c) & 0xF, int(c) >> 4
            if op in (0, 7, 8):
                for j in range(ln):
                    if r.qual[qpos + j] >= min_baseq:
                        hi = int(codes4[qpos + j]) << 4
                        for st in (0, 1):
                            rs[st][dpos + j] = hi | int(ctx_to_idx(ord(xm[st][qpos + j])))
                qpos += ln
                dpos += ln
            elif op in (1, 4):
                qpos += ln
            elif op in (2, 3):
                dpos += ln
            elif op in (5, 6, 9):
                pass
            else:
                raise ValueError("Unknown CIGAR operation")
        has_mods[rstrand] = True                                         # :534
        for st in (0, 1):
            if has_mods[st]:
                rname.append(r.tid + 1)
                strand.append(st + 1)
                start.append(r.pos + trim5 + 1)
                seqs.append(rs[st][trim5:trim5 + max(dpos - (trim5 + trim3), 0)].copy())
    return rname, strand, start, seqs, nrecs


def preprocess_bam(path, paired=None, min_mapq=0, min_baseq=0, skip_duplicates=False,
                   skip_secondary=True, skip_qcfail=True, skip_supplementary=True, trim=0, min_prob=-1, highest_prob=True):
    """preprocessBam() for short-read XG/XM BAMs (R/preprocessBam.R:197-237).

    Returns a dict with sorted SoA columns: xm (uint8, concatenated in row
    order), off (int64, n+1), rname/strand/start (int32), levels (target names).
    """
    names, recs = read_bam_records(path)
    chk = check_bam(recs)
    if chk["nrecs"] == 0:
        raise ValueError("Empty file provided! Exiting")
    if paired is not None and bool(paired) != chk["paired"]:
        raise ValueError("Expected endness is different from detected! Exiting")
    if chk["paired"] and not chk["sorted"]:
        raise ValueError("BAM file seems to be paired-end but not sorted by name!")
    trim5, trim3 = (trim, trim) if np.isscalar(trim) else (trim[0], trim[1])
    skip_flags = 4                                                       # R/internal.R:173-177
    skip_flags += 256 if skip_secondary else 0
    skip_flags += 512 if skip_qcfail else 0
    skip_flags += 1024 if skip_duplicates else 0
    skip_flags += 2048 if skip_supplementary else 0
    long_read = any(("MM" in r.tags or "Mm" in r.tags) for r in recs[:1024])   # R/internal.R:104
    if long_read:
        rname, strand, start, seqs, nrecs = _pack_mm_single(recs, min_mapq, min_baseq, min_prob, highest_prob, skip_flags,
                                                            trim5, trim3)
    elif chk["paired"]:
        skip_flags += 8
        rname, strand, start, seqs, nrecs = _pack_paired(recs, min_mapq, min_baseq, skip_flags, trim5, trim3)
    else:
        rname, strand, start, seqs, nrecs = _pack_single(recs, min_mapq, min_baseq, skip_flags, trim5, trim3)
    rname = np.asarray(rname, np.int32)
    strand = np.asarray(strand, np.int32)
    start = np.asarray(start, np.int32)
    # templid := 0..N-1 ; setorder(rname, start) -- stable (R/internal.R:193-195)
    order = np.lexsort((np.arange(rname.size), start, rname))
    lens = np.asarray([seqs[i].size for i in order], np.int64)
    off = np.zeros(order.size + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    xm = np.concatenate([seqs[i] for i in order]) if order.size else np.zeros(0, np.uint8)
    return {"xm": xm, "off": off, "rname": rname[order], "strand": strand[order],
            "start": start[order], "levels": names, "nrecs": nrecs, "npushed": int(order.size)}
