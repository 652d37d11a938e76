"""ctypes wrapper over oracle/libepi_oracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by anything under epialleler_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C restatement with gcc (no-op when up to date)."""
    so = os.path.join(_HERE, "libepi_oracle.so")
    src = os.path.join(_HERE, "epi_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libepi_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_free.argtypes = [C.c_void_p]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _prep(xm, off, rname=None, strand=None, start=None, passv=None):
    xm = np.ascontiguousarray(xm, dtype=np.uint8)
    if xm.size == 0:
        xm = np.zeros(1, np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    cv = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
    return xm, off, cv(rname), cv(strand), cv(start), cv(passv)


def threshold_reads(xm, off, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth,
                    min_n_ctx=2, min_ctx_meth_frac=0.5, max_ooctx_meth_frac=0.1):
    xm, off, *_ = _prep(xm, off)
    n = off.size - 1
    res = np.zeros(max(n, 1), np.int32)
    lib().orc_threshold_reads(
        _p(xm, C.c_uint8), _p(off, C.c_int64), None, C.c_int64(n),
        ctx_meth.encode("latin1"), ctx_unmeth.encode("latin1"),
        ooctx_meth.encode("latin1"), ooctx_unmeth.encode("latin1"),
        C.c_uint(min_n_ctx), C.c_double(min_ctx_meth_frac), C.c_double(max_ooctx_meth_frac),
        _p(res, C.c_int32))
    return res[:n]


def get_xm_beta(xm, off, ctx_meth, ctx_unmeth):
    xm, off, *_ = _prep(xm, off)
    n = off.size - 1
    res = np.zeros(max(n, 1), np.float64)
    lib().orc_get_xm_beta(_p(xm, C.c_uint8), _p(off, C.c_int64), None, C.c_int64(n),
                          ctx_meth.encode("latin1"), ctx_unmeth.encode("latin1"), _p(res, C.c_double))
    return res[:n]


def _take(ptr, n, dtype):
    if n == 0 or not ptr:
        out = np.zeros(0, dtype)
    else:
        ct = {np.int32: C.c_int32, np.uint64: C.c_uint64}.get(dtype, C.c_double)
        out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).astype(dtype, copy=True)
    if ptr:
        lib().orc_free(ptr)
    return out


def cx_report(xm, off, rname, strand, start, passv, ctx):
    """Returns dict of int32 columns rname,strand,pos,context,meth,unmeth."""
    xm, off, rname, strand, start, passv = _prep(xm, off, rname, strand, start, passv)
    n = off.size - 1
    nrow = C.c_int64(0)
    cols = (C.c_void_p * 6)()
    lib().orc_cx_report(_p(xm, C.c_uint8), _p(off, C.c_int64), None,
                        _p(rname, C.c_int32), _p(strand, C.c_int32), _p(start, C.c_int32),
                        _p(passv, C.c_int32), C.c_int64(n), ctx.encode("latin1"),
                        C.byref(nrow), cols)
    names = ["rname", "strand", "pos", "context", "meth", "unmeth"]
    return {k: _take(cols[i], nrow.value, np.int32) for i, k in enumerate(names)}


def mhl_report(xm, off, rname, strand, start, ctx, hmax=0, hmin=0, max_ooctx_meth_frac=0.1):
    """Returns dict: int32 rname,strand,pos,context,coverage; float64 length,lmhl."""
    xm, off, rname, strand, start, _ = _prep(xm, off, rname, strand, start)
    n = off.size - 1
    nrow = C.c_int64(0)
    icols = (C.c_void_p * 5)()
    dcols = (C.c_void_p * 2)()
    lib().orc_mhl_report(_p(xm, C.c_uint8), _p(off, C.c_int64), None,
                         _p(rname, C.c_int32), _p(strand, C.c_int32), _p(start, C.c_int32),
                         C.c_int64(n), ctx.encode("latin1"), C.c_int(hmax), C.c_int(hmin),
                         C.c_double(max_ooctx_meth_frac), C.byref(nrow), icols, dcols)
    out = {k: _take(icols[i], nrow.value, np.int32)
           for i, k in enumerate(["rname", "strand", "pos", "context", "coverage"])}
    out["length"] = _take(dcols[0], nrow.value, np.float64)
    out["lmhl"] = _take(dcols[1], nrow.value, np.float64)
    return out


def extract_patterns(xm, off, rname, strand, start, target_rname, target_start, target_end, min_overlap, ctx,
                     min_ctx_freq, clip, reverse_offset, hlght=()):
    """rcpp_extract_patterns: dict with per-pattern strand/start/end/nbase (int32), beta (float64), fnv (uint64),
    the sorted column positions and cells[ncol, npat] (context / base factor codes, INT32_MIN = NA)."""
    xm, off, rname, strand, start, _ = _prep(xm, off, rname, strand, start)
    n = off.size - 1
    hl = np.ascontiguousarray(hlght, dtype=np.int32)
    npat, ncol = C.c_int64(0), C.c_int32(0)
    ptrs = [C.c_void_p() for _ in range(8)]
    lib().orc_extract_patterns(_p(xm, C.c_uint8), _p(off, C.c_int64), None, _p(rname, C.c_int32), _p(strand, C.c_int32),
                               _p(start, C.c_int32), C.c_int64(n), C.c_uint(target_rname), C.c_uint(target_start),
                               C.c_uint(target_end), C.c_int(min_overlap), ctx.encode("latin1"), C.c_double(min_ctx_freq),
                               C.c_int(int(bool(clip))), C.c_uint(reverse_offset), _p(hl, C.c_int32) if hl.size else None,
                               C.c_int32(hl.size), C.byref(npat), C.byref(ncol), *[C.byref(p) for p in ptrs])
    k, m = npat.value, ncol.value
    out = {nm: _take(ptrs[i], k, np.int32) for i, nm in enumerate(["strand", "start", "end", "nbase"])}
    out["beta"] = _take(ptrs[4], k, np.float64)
    out["fnv"] = _take(ptrs[5], k, np.uint64)
    out["positions"] = _take(ptrs[6], m, np.int32)
    out["cells"] = _take(ptrs[7], m * k, np.int32).reshape(m, k)
    return out
