"""BED-assisted reports: the R-level callers that reuse thresholding and per-read beta
(SURVEY 8f row 3).  Mirrors R/generateBedReport.R:219-273 (+ generateAmpliconReport /
generateCaptureReport aliases), R/generateBedEcdf.R:122-153 and the helpers
.readBed / .matchTarget / .getBedReport / .getBedEcdf (R/internal.R:205-222, 463-478, 529-604).
Matching runs on the GPU (epi_batch_match_target_dev); the per-region tallies are what data.table's
dcast/merge do in the reference and are done with torch.bincount on the device vectors.
"""
import ctypes as C

import numpy as np

from . import _lib
from .api import (CONTEXT_TO_BASES, Report, _CTX_CHOICES, _as_bam, _match_arg, _stream, preprocessBam,
                  rcpp_extract_patterns, rcpp_get_xm_beta, rcpp_threshold_reads, writeReport)

NA_INTEGER = -2 ** 31


class Bed:
    """A BED table as GRanges would hold it: 1-based closed ranges + the extra columns."""

    def __init__(self, chrom, start, end, extra=None):
        self.chrom = list(chrom)
        self.start = np.asarray(start, np.int64)
        self.end = np.asarray(end, np.int64)
        self.extra = dict(extra or {})

    def __len__(self):
        return len(self.chrom)

    def names(self):
        """as.character(GRanges): "chr:start-end"."""
        return ["%s:%d-%d" % (c, s, e) for c, s, e in zip(self.chrom, self.start, self.end)]


def readBed(bed_file, zero_based_bed=False):
    """R/internal.R:205-222: tab-separated, blank lines skipped, first three columns chr/start/end,
    header detected as data.table::fread does (a first line whose start/end are not numeric)."""
    rows = []
    with open(bed_file) as f:
        for ln in f:
            ln = ln.rstrip("\n\r")
            if ln.strip():
                rows.append(ln.split("\t"))
    header = None
    if rows and not (rows[0][1].lstrip("-").isdigit() and rows[0][2].lstrip("-").isdigit()):
        header, rows = rows[0], rows[1:]
    ncol = max(len(r) for r in rows) if rows else 3
    names = list(header) if header else ["V%d" % (i + 1) for i in range(ncol)]
    names[:3] = ["chr", "start", "end"]
    start = np.asarray([int(r[1]) for r in rows], np.int64) + (1 if zero_based_bed else 0)
    end = np.asarray([int(r[2]) for r in rows], np.int64)
    extra = {names[i]: [r[i] if i < len(r) else "" for r in rows] for i in range(3, ncol)}
    return Bed([r[0] for r in rows], start, end, extra)


def _match_target(bam, bed, bed_type, match_tolerance, match_min_overlap):
    """.matchTarget (R/internal.R:463-478): BED seqnames become factor codes of the BAM's rname levels."""
    import torch
    lib = _lib.load()
    b = bam.batch()
    dev = "cuda:%d" % bam.device
    levels = list(bam.levels) if bam.levels is not None else []
    idx = {c: i + 1 for i, c in enumerate(levels)}
    chrom = np.asarray([idx.get(c, NA_INTEGER) for c in bed.chrom], np.int32)   # NA never equals an rname
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
    d_chr, d_s, d_e = t(chrom), t(bed.start), t(bed.end)
    out = torch.empty(max(bam.n, 1), dtype=torch.int32, device=dev)
    _lib.check(lib.epi_batch_match_target_dev(
        b, C.c_void_p(d_chr.data_ptr()), C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), len(bed),
        1 if bed_type == "capture" else 0, int(match_min_overlap if bed_type == "capture" else match_tolerance),
        C.c_void_p(out.data_ptr()), _stream(bam.device)))
    return out[:bam.n]


def generateBedReport(bam, bed, report_file=None, zero_based_bed=False, bed_type=None, match_tolerance=1,
                      match_min_overlap=1, threshold_reads=True, threshold_context=None, min_context_sites=2,
                      min_context_beta=0.5, max_outofcontext_beta=0.1, gzip=False, verbose=False, **preprocess_args):
    """R/generateBedReport.R:219-273.  Returns the BED rows (plus a last NA row for unmatched reads, when there
    are any) with columns seqnames,start,end,width,strand,<extra>,nreads+,nreads-,VEF; NaN where R has NA."""
    import torch
    bed_type = _match_arg(bed_type, ("amplicon", "capture"), "bed.type")
    threshold_context = _match_arg(threshold_context, _CTX_CHOICES, "threshold.context")
    if not isinstance(bed, Bed):
        bed = readBed(bed, zero_based_bed)
    bam = _as_bam(preprocessBam(bam, **preprocess_args))
    bam.batch()
    dev = "cuda:%d" % bam.device
    if threshold_reads:
        c = CONTEXT_TO_BASES[threshold_context]
        pass_ = rcpp_threshold_reads(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"],
                                     min_context_sites, min_context_beta, max_outofcontext_beta, as_device=True)
    else:
        pass_ = torch.ones(bam.n, dtype=torch.int32, device=dev)
    match = _match_target(bam, bed, bed_type, match_tolerance, match_min_overlap)
    strand = bam.dev["strand"] if bam.dev is not None else torch.from_numpy(bam.host["strand"]).to(dev)
    # .getBedReport (R/internal.R:529-561): reads per (bedmatch, strand, pass); NA matches get slot nbed
    nbed = len(bed)
    slot = torch.where(match < 0, torch.full_like(match, nbed), match - 1).to(torch.int64)
    key = slot * 4 + (strand.to(torch.int64) - 1) * 2 + (pass_ != 0).to(torch.int64)
    key = key[(strand >= 1) & (strand <= 2)]                # (strand 0: the reference's placeholder template of an empty paired-end
                                                            # file, src/rcpp_read_bam.cpp:155 -- an NA strand, counted on neither)
    cnt = torch.bincount(key, minlength=(nbed + 1) * 4).reshape(nbed + 1, 2, 2).cpu().numpy().astype(np.float64)
    present = cnt.sum(axis=(1, 2)) > 0                      # regions with no read at all are NA after the merge
    rows = list(range(nbed)) + ([nbed] if present[nbed] else [])
    npl, nmi = cnt[:, 0, :].sum(1), cnt[:, 1, :].sum(1)
    vef = np.divide(cnt[:, :, 1].sum(1), npl + nmi, out=np.full(nbed + 1, np.nan), where=(npl + nmi) > 0)
    na = lambda a: np.where(present, a, np.nan)[rows]
    cols = {"seqnames": np.asarray(bed.chrom + [None], object)[rows],
            "start": np.append(bed.start.astype(np.float64), np.nan)[rows],
            "end": np.append(bed.end.astype(np.float64), np.nan)[rows],
            "width": np.append((bed.end - bed.start + 1).astype(np.float64), np.nan)[rows],
            "strand": np.asarray(["*"] * nbed + [None], object)[rows]}
    for k, v in bed.extra.items():
        cols[k] = np.asarray(list(v) + [None], object)[rows]
    cols["nreads+"] = na(npl)
    cols["nreads-"] = na(nmi)
    cols["VEF"] = na(vef) if threshold_reads else np.full(len(rows), np.nan)          # :267
    rep = Report(cols)
    rep.levels = {}
    if report_file is None:
        return rep
    writeReport(rep, report_file, gzip)
    return None


def generateAmpliconReport(bam, bed, **kw):
    """R/generateBedReport.R:187-199."""
    kw.pop("bed_type", None)
    return generateBedReport(bam, bed, bed_type="amplicon", **kw)


def generateCaptureReport(bam, bed, **kw):
    """R/generateBedReport.R:203-215."""
    kw.pop("bed_type", None)
    return generateBedReport(bam, bed, bed_type="capture", **kw)


class Ecdf:
    """stats::ecdf of a numeric vector: right-continuous step function, callable on scalars or arrays."""

    def __init__(self, x):
        self.x = np.sort(np.asarray(x, np.float64))

    def __call__(self, q):
        n = self.x.size
        r = np.searchsorted(self.x, np.asarray(q, np.float64), side="right") / n if n else np.full(np.shape(q), np.nan)
        return float(r) if np.ndim(q) == 0 else r


def generateBedEcdf(bam, bed, bed_type=None, bed_rows=(1,), zero_based_bed=False, match_tolerance=1,
                    match_min_overlap=1, ecdf_context=None, verbose=False, **preprocess_args):
    """R/generateBedEcdf.R:122-153 + .getBedEcdf (R/internal.R:568-604).  bed_rows: 1-based BED rows, None = all
    (including the NA group of unmatched reads, keyed None).  Returns {region name: {"context": Ecdf,
    "out.of.context": Ecdf}} in the reference's order."""
    bed_type = _match_arg(bed_type, ("amplicon", "capture"), "bed.type")
    ecdf_context = _match_arg(ecdf_context, _CTX_CHOICES, "ecdf.context")
    if not isinstance(bed, Bed):
        bed = readBed(bed, zero_based_bed)
    bam = _as_bam(preprocessBam(bam, **preprocess_args))
    c = CONTEXT_TO_BASES[ecdf_context]
    match = _match_target(bam, bed, bed_type, match_tolerance, match_min_overlap).cpu().numpy()
    ctx_beta = rcpp_get_xm_beta(bam, c["ctx_meth"], c["ctx_unmeth"])
    oo_beta = rcpp_get_xm_beta(bam, c["ooctx_meth"], c["ooctx_unmeth"])
    all_rows = sorted(set(int(m) for m in np.unique(match[match > 0]))) + ([None] if (match < 0).any() else [])
    rows = all_rows if bed_rows is None else [r for r in bed_rows if r in all_rows]      # intersect(bed.rows, all.bed.rows)
    names = bed.names()
    out = {}
    for r in rows:
        sel = (match < 0) if r is None else (match == r)
        out[None if r is None else names[r - 1]] = {"context": Ecdf(ctx_beta[sel]), "out.of.context": Ecdf(oo_beta[sel])}
    return out


def extractPatterns(bam, bed, bed_row=1, zero_based_bed=False, match_min_overlap=1, extract_context=None,
                    min_context_freq=0.01, clip_patterns=False, strand_offset=None, highlight_positions=(), verbose=False,
                    **preprocess_args):
    """R/extractPatterns.R:107-143 + .getPatterns (R/internal.R:683-714).  bed: a path, a Bed, or a string
    "chr:start-end"; bed_row is 1-based.  The result carries the BED row it belongs to in `.bed`."""
    extract_context = _match_arg(extract_context, _CTX_CHOICES, "extract.context")
    if strand_offset is None:
        strand_offset = {"CG": 1, "CHG": 2, "CHH": 0, "CxG": 0, "CX": 0}[extract_context]
    if isinstance(bed, str) and ":" in bed and "-" in bed.rsplit(":", 1)[1] and not __import__("os").path.exists(bed):
        chrom, rng = bed.rsplit(":", 1)                                        # as("chr:start-end", "GRanges")
        a, b_ = rng.split("-")
        bed = Bed([chrom], [int(a)], [int(b_)])
    elif not isinstance(bed, Bed):
        bed = readBed(bed, zero_based_bed)
    bam = _as_bam(preprocessBam(bam, **preprocess_args))
    row = int(np.atleast_1d(bed_row)[0]) - 1
    if row < 0 or row >= len(bed):
        return Report({}, bam.levels)                                          # data.table()[bed.row] of a missing row: no target
    levels = list(bam.levels) if bam.levels is not None else []
    chrom = bed.chrom[row]
    seq = levels.index(chrom) + 1 if chrom in levels else NA_INTEGER           # factor(seqnames, levels=levels(rname))
    start, end = int(bed.start[row]), int(bed.end[row])
    hl = sorted({int(p) for p in np.atleast_1d(np.asarray(highlight_positions, np.int64)) if start <= int(p) <= end})
    c = CONTEXT_TO_BASES[extract_context]
    rep = rcpp_extract_patterns(bam, seq, start, end, match_min_overlap, c["ctx_meth"] + c["ctx_unmeth"], min_context_freq,
                                clip_patterns, int(strand_offset), hl)
    rep.bed = bed.names()[row]
    return rep
