"""Host-side mirror of the reference's R interface for the hot path.

Same names, argument meaning, defaults and error behaviour as
  R/generateCytosineReport.R:164-208, R/generateMhlReport.R:170-197,
  R/preprocessBam.R:197-237, R/internal.R:54-65 / 439-456 / 486-522
and of the Rcpp exports they call (R/RcppExports.R: rcpp_threshold_reads,
rcpp_get_xm_beta, rcpp_cx_report, rcpp_mhl_report).  All compute goes through
the C ABI of libepihip.so (include/epihip.h); torch is used only for device
memory and streams.  There is no CPU path here.
"""
import ctypes as C
import gzip as _gzip

import numpy as np

from . import _lib

# R/internal.R:54-65 (.context.to.bases), verbatim
CONTEXT_TO_BASES = {
    "CG": dict(ctx_meth="Z", ctx_unmeth="z", ooctx_meth="XH", ooctx_unmeth="xh"),
    "CHG": dict(ctx_meth="X", ctx_unmeth="x", ooctx_meth="ZH", ooctx_unmeth="zh"),
    "CHH": dict(ctx_meth="H", ctx_unmeth="h", ooctx_meth="ZX", ooctx_unmeth="zx"),
    "CxG": dict(ctx_meth="ZX", ctx_unmeth="zx", ooctx_meth="H", ooctx_unmeth="h"),
    "CX": dict(ctx_meth="ZXH", ctx_unmeth="zxh", ooctx_meth="", ooctx_unmeth=""),
}
STRAND_LEVELS = ("+", "-")                                            # src/rcpp_read_bam.cpp:175
CONTEXT_LEVELS = ("NA1", "CHH", "NA3", "NA4", "NA5", "CHG", "CG")     # src/rcpp_cx_report.cpp:150-152

_engines = {}


def _torch():
    import torch
    return torch


def _engine(device=None):
    """One epi_engine per GPU, created on first use.  Fails loudly without a GPU."""
    lib = _lib.load()
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _engines:
        h = C.c_void_p()
        _lib.check(lib.epi_engine_create(int(device), C.byref(h)))
        _engines[device] = h
    return _engines[device]


def _stream(device):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Report(dict):
    """A report table: dict of equal-length numpy (or torch) columns plus factor levels,
    the analogue of the data.table the reference returns."""

    def __init__(self, cols, rname_levels=None):
        super().__init__(cols)
        self.levels = {"rname": tuple(rname_levels) if rname_levels is not None else None,
                       "strand": STRAND_LEVELS, "context": CONTEXT_LEVELS}

    @property
    def nrow(self):
        return int(next(iter(self.values())).shape[0]) if self else 0


class ProcessedBam:
    """What preprocessBam() returns in the reference: templates sorted by (rname,start)
    (R/internal.R:193-195) -- here as SoA columns (packed SEQXM bytes + offsets instead of a
    vector of strings behind seqxm_xptr), lazily made resident in HBM."""

    def __init__(self, n, nbytes, levels=None):
        self.n = int(n)
        self.nbytes = int(nbytes)
        self.levels = tuple(levels) if levels is not None else None
        self.host = None        # dict of numpy arrays or None
        self.dev = None         # dict of torch tensors (kept alive for an adopted batch) or None
        self.device = None
        self._batch = None
        self.realign = True     # adopted device columns: let the engine lay the rows out its own way (from_device)
        self._keep = None       # owner of the host buffers the numpy columns are views of (producer output, pinned tensors)

    @classmethod
    def from_arrays(cls, xm, off, rname, strand, start, levels=None, keepalive=None, device=None):
        off = np.ascontiguousarray(off, dtype=np.int64)
        n = off.size - 1
        if n < 0:
            raise ValueError("off must have n+1 entries")
        xm = np.ascontiguousarray(xm, dtype=np.uint8)
        cols = [np.ascontiguousarray(a, dtype=np.int32) for a in (rname, strand, start)]
        for a in cols:
            if a.size != n:
                raise ValueError("column length does not match off")
        if n and int(off[-1]) != xm.size:
            raise ValueError("off[n] does not match len(xm)")
        self = cls(n, int(off[-1]) if off.size else 0, levels)
        self.host = dict(xm=xm, off=off, rname=cols[0], strand=cols[1], start=cols[2])
        self._keep = keepalive
        self.device = device
        return self

    @classmethod
    def from_pinned(cls, xm, nbytes, off, rname, strand, start, levels=None, device=None):
        """Pinned host tensors (torch, CPU): the columns are used in place -- epi_batch_upload recognises page-locked
        sources and DMAs straight from them (no staging copy)."""
        return cls.from_arrays(xm.numpy()[:int(nbytes)], off.numpy(), rname.numpy(), strand.numpy(), start.numpy(), levels,
                               keepalive=(xm, off, rname, strand, start), device=device)

    @classmethod
    def from_device(cls, xm, nbytes, off, rname, strand, start, levels=None, realign=True):
        """Adopt torch tensors already in HBM (xm: uint8 with capacity >= nbytes rounded up to 16).  realign: the engine
        makes its own position-congruent copy of xm when the batch is created (epi_batch_realign, include/epihip.h) and
        reads neither xm nor off afterwards; drop_source() then gives their memory back.  False: strictly zero-copy."""
        n = int(off.numel()) - 1
        self = cls(n, nbytes, levels)
        self.dev = dict(xm=xm, off=off, rname=rname, strand=strand, start=start)
        self.device = xm.device.index
        self.realign = bool(realign)
        return self

    def drop_source(self):
        """After the batch exists in the engine's own layout: release the adopted xm / off tensors."""
        if self.dev is not None and self._batch is not None and _lib.load().epi_batch_layout(self._batch) > 0:
            self.dev = {k: v for k, v in self.dev.items() if k not in ("xm", "off")}

    def batch(self, device=None):
        """The epi_batch handle (uploads on first use through the library's pinned double-buffered path)."""
        if self._batch is not None:
            return self._batch
        lib = _lib.load()
        h = C.c_void_p()
        if self.dev is not None:
            eng = _engine(self.device)
            d = self.dev
            _lib.check(lib.epi_batch_adopt(eng, C.c_void_p(d["xm"].data_ptr()), d["xm"].numel(), self.nbytes,
                                           C.c_void_p(d["off"].data_ptr()), C.c_void_p(d["rname"].data_ptr()),
                                           C.c_void_p(d["strand"].data_ptr()), C.c_void_p(d["start"].data_ptr()),
                                           self.n, C.byref(h)))
            if getattr(self, "realign", True):
                rc = lib.epi_batch_realign(h, None)
                if rc:
                    lib.epi_batch_free(h)
                    _lib.check(rc)
        else:
            eng = _engine(device if device is not None else self.device)
            self.device = lib.epi_engine_device(eng)
            hst = self.host
            p = lambda a: C.c_void_p(a.ctypes.data) if a.size else None
            _lib.check(lib.epi_batch_upload(eng, p(hst["xm"]), C.c_void_p(hst["off"].ctypes.data), p(hst["rname"]),
                                            p(hst["strand"]), p(hst["start"]), self.n, C.byref(h)))
        self._batch = h
        return h

    def close(self):
        if self._batch is not None:
            _lib.load().epi_batch_free(self._batch)
            self._batch = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _TemplatesOwner:
    """Keeps an epi_templates (epi_preprocess_bam output) alive while numpy views of its buffers are in use."""

    def __init__(self, t):
        self.t = t

    def __del__(self):
        try:
            _lib.load().epi_templates_free(C.byref(self.t))
        except Exception:
            pass


def _as_bam(bam):
    if isinstance(bam, ProcessedBam):
        return bam
    if isinstance(bam, dict):
        return ProcessedBam.from_arrays(bam["xm"], bam["off"], bam["rname"], bam["strand"], bam["start"],
                                        bam.get("levels"))
    raise TypeError("expected a ProcessedBam (preprocessBam() result) or a dict of SoA columns")


def preprocessBam(bam_file, paired=None, min_mapq=0, min_baseq=0, min_prob=-1, highest_prob=True,
                  skip_duplicates=False, skip_secondary=True, skip_qcfail=True, skip_supplementary=True,
                  trim=0, nthreads=1, verbose=False, window_kib=0):
    """R/preprocessBam.R:197-237.  An already preprocessed object is returned untouched (:226-235);
    a path is decoded by the library's host-side producer (epi_preprocess_bam: zlib BGZF reader + the
    reference's template packer), which yields the sorted SoA batch directly."""
    if isinstance(bam_file, (ProcessedBam, dict)):
        return _as_bam(bam_file)
    import os
    lib = _lib.load()
    trim2 = (list(np.atleast_1d(trim)) * 2)[:2]                       # head(rep.int(trim, 2), 2)
    opt = _lib.BamOptions(int(min_mapq), int(min_baseq), int(bool(skip_duplicates)), int(bool(skip_secondary)),
                          int(bool(skip_qcfail)), int(bool(skip_supplementary)), int(trim2[0]), int(trim2[1]),
                          -1 if paired is None else int(bool(paired)), max(int(nthreads), 1), int(min_prob),
                          int(bool(highest_prob)), int(window_kib))
    t = _lib.Templates()
    rc = lib.epi_preprocess_bam(os.path.expanduser(str(bam_file)).encode(), C.byref(opt), C.byref(t))
    if rc != _lib.EPI_OK:
        msg = lib.epi_last_error().decode("utf-8", "replace")
        raise ValueError(msg)                                         # stop(..., call.=FALSE) in the reference
    # the producer's buffers are used in place (xm is pinned when a device is usable: the upload DMAs straight from it);
    # they are released when the ProcessedBam goes away
    keep = _TemplatesOwner(t)
    n = t.n

    def view(ptr, k):
        # every array owns its buffer: .base is a ctypes array that carries the _TemplatesOwner, so
        # `preprocessBam(p).host["xm"]` stays valid after the ProcessedBam itself is gone
        if k <= 0 or not ptr:
            return np.empty(0, dtype=np.dtype(ptr._type_))
        carr = (ptr._type_ * int(k)).from_address(C.addressof(ptr.contents))
        carr._owner = keep
        return np.frombuffer(carr, dtype=np.dtype(ptr._type_))
    levels = tuple(t.target_names[i].decode("latin1") for i in range(t.n_targets))
    bam = ProcessedBam.from_arrays(view(t.xm, t.nbytes), view(t.off, n + 1), view(t.rname, n), view(t.strand, n),
                                   view(t.start, n), levels, keepalive=keep)
    bam.nrecs, bam.npushed, bam.paired, bam.pinned = int(t.nrecs), int(n), bool(t.paired), bool(t.pinned)
    return bam


# ---- Rcpp-level functions ----------------------------------------------------------------------

def rcpp_threshold_reads(df, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n_ctx, min_ctx_meth_frac,
                         max_ooctx_meth_frac, as_device=False):
    """src/rcpp_threshold_reads.cpp:15-74 -> logical vector (numpy bool, or int32 torch tensor in HBM)."""
    torch = _torch()
    bam = _as_bam(df)
    b = bam.batch()
    out = torch.empty(max(bam.n, 1), dtype=torch.int32, device="cuda:%d" % bam.device)
    _lib.check(_lib.load().epi_batch_threshold_reads_dev(
        b, _lib.enc(ctx_meth), _lib.enc(ctx_unmeth), _lib.enc(ooctx_meth), _lib.enc(ooctx_unmeth),
        int(min_n_ctx), float(min_ctx_meth_frac), float(max_ooctx_meth_frac),
        C.c_void_p(out.data_ptr()), _stream(bam.device)))
    out = out[:bam.n]
    return out if as_device else out.cpu().numpy().astype(bool)


def rcpp_get_xm_beta(df, ctx_meth, ctx_unmeth, as_device=False):
    """src/rcpp_get_xm_beta.cpp:10-43 -> per-read beta (float64)."""
    torch = _torch()
    bam = _as_bam(df)
    b = bam.batch()
    out = torch.empty(max(bam.n, 1), dtype=torch.float64, device="cuda:%d" % bam.device)
    _lib.check(_lib.load().epi_batch_get_xm_beta_dev(b, _lib.enc(ctx_meth), _lib.enc(ctx_unmeth),
                                                     C.c_void_p(out.data_ptr()), _stream(bam.device)))
    out = out[:bam.n]
    return out if as_device else out.cpu().numpy()


def _pass_tensor(bam, pass_):
    """R logical -> int32 device tensor (NA stays non-zero = TRUE, src/rcpp_cx_report.cpp:118)."""
    torch = _torch()
    if pass_ is None:
        return None
    if isinstance(pass_, torch.Tensor):
        t = pass_.to(device="cuda:%d" % bam.device, dtype=torch.int32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(pass_).astype(np.int32))).to("cuda:%d" % bam.device)
    if t.numel() != bam.n:
        raise ValueError("pass must have one entry per template")
    return t.contiguous()


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def rcpp_cx_report(df, pass_, ctx, as_device=False):
    """src/rcpp_cx_report.cpp:34-159 -> columns rname,strand,pos,context,meth,unmeth (int32)."""
    torch = _torch()
    lib = _lib.load()
    bam = _as_bam(df)
    b = bam.batch()
    dev = "cuda:%d" % bam.device
    p = _pass_tensor(bam, pass_)
    nrow = C.c_int64(0)
    _lib.check(lib.epi_batch_cx_report_dev(b, C.c_void_p(p.data_ptr()) if p is not None and bam.n else None,
                                           _lib.enc(ctx), _stream(bam.device), C.byref(nrow)))
    n = nrow.value
    cols = list(torch.empty((6, n), dtype=torch.int32, device=dev).unbind(0))   # one allocation: the GPU idles meanwhile
    if n:
        _lib.check(lib.epi_batch_cx_fetch_dev(b, _ptr_array(cols), _stream(bam.device)))
    names = ("rname", "strand", "pos", "context", "meth", "unmeth")
    if not as_device:
        cols = [c.cpu().numpy() for c in cols]
    return Report(dict(zip(names, cols)), bam.levels)


def cytosine_report_fused(df, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n_ctx, min_ctx_meth_frac,
                          max_ooctx_meth_frac, ctx, as_device=False, return_pass=False):
    """rcpp_threshold_reads followed by rcpp_cx_report with its result (what generateCytosineReport does with
    threshold.reads=TRUE, R/generateCytosineReport.R:181-199) as ONE call: epi_batch_cytosine_report_dev decides every
    read inside the tile kernel, from the bytes it has loaded anyway.  Same table as the two calls."""
    torch = _torch()
    lib = _lib.load()
    bam = _as_bam(df)
    b = bam.batch()
    dev = "cuda:%d" % bam.device
    pass_out = torch.empty(max(bam.n, 1), dtype=torch.int32, device=dev) if return_pass else None
    nrow = C.c_int64(0)
    _lib.check(lib.epi_batch_cytosine_report_dev(
        b, _lib.enc(ctx_meth), _lib.enc(ctx_unmeth), _lib.enc(ooctx_meth), _lib.enc(ooctx_unmeth), int(min_n_ctx),
        float(min_ctx_meth_frac), float(max_ooctx_meth_frac), _lib.enc(ctx),
        C.c_void_p(pass_out.data_ptr()) if pass_out is not None else None, _stream(bam.device), C.byref(nrow)))
    n = nrow.value
    cols = list(torch.empty((6, n), dtype=torch.int32, device=dev).unbind(0))
    if n:
        _lib.check(lib.epi_batch_cx_fetch_dev(b, _ptr_array(cols), _stream(bam.device)))
    names = ("rname", "strand", "pos", "context", "meth", "unmeth")
    if not as_device:
        cols = [c.cpu().numpy() for c in cols]
    rep = Report(dict(zip(names, cols)), bam.levels)
    if return_pass:
        p = pass_out[:bam.n]
        return rep, (p if as_device else p.cpu().numpy().astype(bool))
    return rep


def rcpp_mhl_report(df, ctx, hmax, hmin, max_ooctx_meth_frac, as_device=False):
    """src/rcpp_mhl_report.cpp:46-228 -> rname,strand,pos,context,coverage (int32), length,lmhl (float64)."""
    torch = _torch()
    lib = _lib.load()
    bam = _as_bam(df)
    b = bam.batch()
    dev = "cuda:%d" % bam.device
    nrow = C.c_int64(0)
    _lib.check(lib.epi_batch_mhl_report_dev(b, _lib.enc(ctx), int(hmax), int(hmin), float(max_ooctx_meth_frac),
                                            _stream(bam.device), C.byref(nrow)))
    n = nrow.value
    icols = list(torch.empty((5, n), dtype=torch.int32, device=dev).unbind(0))
    dcols = list(torch.empty((2, n), dtype=torch.float64, device=dev).unbind(0))
    if n:
        _lib.check(lib.epi_batch_mhl_fetch_dev(b, _ptr_array(icols), _ptr_array(dcols), _stream(bam.device)))
    cols = icols + dcols
    if not as_device:
        cols = [c.cpu().numpy() for c in cols]
    names = ("rname", "strand", "pos", "context", "coverage", "length", "lmhl")
    return Report(dict(zip(names, cols)), bam.levels)


PATTERN_LEVELS = ("NA1", "H", "A", "C", "NA5", "X", "Z", "NA8", "NA9", "h", "G", "T", "N", "x", "z", "NA16")   # :192-195
NA_INTEGER = -2 ** 31


def rcpp_extract_patterns(df, target_rname, target_start, target_end, min_overlap, ctx, min_ctx_freq, clip,
                          reverse_offset, hlght=()):
    """src/rcpp_extract_patterns.cpp:26-211 -> Report with seqnames, strand, start, end, nbase, beta, pattern (16 hex
    digits) and one int32 column per position (context / base factor codes, levels PATTERN_LEVELS, NA = -2^31), or an
    empty Report when no pattern was found."""
    lib = _lib.load()
    bam = _as_bam(df)
    b = bam.batch()
    hl = np.ascontiguousarray(hlght, dtype=np.int32)
    t = _lib.PatternTable()
    _lib.check(lib.epi_batch_extract_patterns(b, int(target_rname), int(target_start), int(target_end), int(min_overlap),
                                              _lib.enc(ctx), float(min_ctx_freq), int(bool(clip)), int(reverse_offset),
                                              C.c_void_p(hl.ctypes.data) if hl.size else None, int(hl.size),
                                              _stream(bam.device), C.byref(t)))
    try:
        k, m = int(t.npat), int(t.ncol)
        if k == 0:
            return Report({}, bam.levels)
        take = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)
        cols = {"seqnames": np.full(k, int(target_rname), np.int32)}
        for nm in ("strand", "start", "end", "nbase"):
            cols[nm] = take(getattr(t, nm), k, np.int32)
        cols["beta"] = take(t.beta, k, np.float64)
        cols["pattern"] = np.asarray(["%016X" % int(v) for v in take(t.fnv, k, np.uint64)], object)      # :174-176
        pos = take(t.positions, m, np.int32)
        cells = take(t.cells, m * k, np.int32).reshape(m, k)
        for i in range(m):
            cols[str(int(pos[i]))] = cells[i]
    finally:
        lib.epi_pattern_table_free(C.byref(t))
    rep = Report(cols, bam.levels)
    rep.pattern_levels = PATTERN_LEVELS
    return rep


# ---- exported R API ------------------------------------------------------------------------------

def _match_arg(value, choices, name):
    if value is None:
        return choices[0]                       # match.arg: first choice is the default
    if value not in choices:
        raise ValueError("'%s' should be one of %s" % (name, ", ".join(repr(c) for c in choices)))
    return value


_CTX_CHOICES = ("CG", "CHG", "CHH", "CxG", "CX")


def generateCytosineReport(bam, report_file=None, threshold_reads=True, threshold_context=None,
                           min_context_sites=2, min_context_beta=0.5, max_outofcontext_beta=0.1,
                           report_context=None, gzip=False, verbose=False, as_device=False, **preprocess_args):
    """R/generateCytosineReport.R:164-208."""
    threshold_context = _match_arg(threshold_context, _CTX_CHOICES, "threshold.context")
    report_context = threshold_context if report_context is None else _match_arg(report_context, _CTX_CHOICES, "report.context")
    bam = preprocessBam(bam, **preprocess_args)
    if threshold_reads:
        c = CONTEXT_TO_BASES[threshold_context]   # .thresholdReads + .getCytosineReport (:181-199) in one pass over the bytes
        rep = cytosine_report_fused(bam, c["ctx_meth"], c["ctx_unmeth"], c["ooctx_meth"], c["ooctx_unmeth"],
                                    min_context_sites, min_context_beta, max_outofcontext_beta,
                                    CONTEXT_TO_BASES[report_context]["ctx_meth"], as_device=as_device)
    else:                                       # pass <- rep(TRUE, nrow(bam)), :193-195
        rep = rcpp_cx_report(bam, None, CONTEXT_TO_BASES[report_context]["ctx_meth"], as_device=as_device)
    if report_file is None:
        return rep
    writeReport(rep, report_file, gzip)
    return None


def generateMhlReport(bam, report_file=None, haplotype_context=None, max_haplotype_window=0,
                      min_haplotype_length=0, max_outofcontext_beta=0.1, gzip=False, verbose=False,
                      as_device=False, **preprocess_args):
    """R/generateMhlReport.R:170-197."""
    haplotype_context = _match_arg(haplotype_context, _CTX_CHOICES, "haplotype.context")
    bam = preprocessBam(bam, **preprocess_args)
    c = CONTEXT_TO_BASES[haplotype_context]
    rep = rcpp_mhl_report(bam, c["ctx_meth"] + c["ctx_unmeth"], max_haplotype_window, min_haplotype_length,
                          max_outofcontext_beta, as_device=as_device)
    if report_file is None:
        return rep
    writeReport(rep, report_file, gzip)
    return None


def writeReport(report, report_file, gzip=False, nthreads=None):
    """R/internal.R:274-287 (.writeReport): TSV with header, factors written as their labels, NA / NaN as empty
    fields -- data.table::fwrite's conventions.  Numeric and factor columns go through the library's threaded writer
    (epi_write_report); a table with a text column (extractPatterns' hash) is written row by row here."""
    import os
    cols, keep = [], []
    n = report.nrow
    text_cols = False
    for k, v in report.items():
        a = v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)
        lev = report.levels.get(k)
        if a.dtype.kind == "f":
            a = np.ascontiguousarray(a, np.float64)
            cols.append((k, 1, a, None))
        elif a.dtype.kind in "iub":
            a = np.ascontiguousarray(a, np.int32)
            cols.append((k, 2 if lev is not None else 0, a, lev))
        else:
            text_cols = True
            cols.append((k, -1, a, lev))
    if text_cols or not cols:
        _write_report_py(cols, n, report_file, gzip)
        return
    lib = _lib.load()
    arr = (_lib.ReportColumn * len(cols))()
    for i, (k, kind, a, lev) in enumerate(cols):
        arr[i].name = k.encode()
        arr[i].kind = kind
        arr[i].data = a.ctypes.data if a.size else None
        if lev is not None:
            lv = (C.c_char_p * len(lev))(*[str(x).encode("latin1") for x in lev])
            keep.append(lv)
            arr[i].levels = lv
            arr[i].nlevels = len(lev)
        keep.append(a)
    if nthreads is None:
        nthreads = min(os.cpu_count() or 1, 16)
    rc = lib.epi_write_report(os.path.expanduser(str(report_file)).encode(), arr, len(cols), n, int(bool(gzip)), int(nthreads))
    if rc != _lib.EPI_OK:
        raise OSError(lib.epi_last_error().decode("utf-8", "replace"))


def _write_report_py(cols, n, report_file, gzip):
    opener = (lambda p: _gzip.open(p, "wt")) if gzip else (lambda p: open(p, "w"))

    def cell(kind, a, lev, i):
        v = a[i]
        if kind == 1:
            return "" if np.isnan(v) else ("%.15g" % v)
        if kind == -1:
            return "" if v is None else str(v)
        if int(v) == -2 ** 31:
            return ""
        if lev is not None:
            return str(lev[int(v) - 1]) if 1 <= int(v) <= len(lev) else ""
        return str(int(v))

    with opener(report_file) as f:
        f.write("\t".join(k for k, _, _, _ in cols) + "\n")
        for i in range(n):
            f.write("\t".join(cell(kind, a, lev, i) for _, kind, a, lev in cols) + "\n")
