"""ctypes binding of libepihip.so (the C ABI in include/epihip.h).

The product path has no CPU fallback: if the library has not been built, or no
HIP device is usable, loading / engine creation raises EpihipError.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("EPIHIP_LIB") or os.path.join(CSRC, "libepihip.so")   # EPIHIP_LIB: a development build

EPI_OK, EPI_ERR_ARG, EPI_ERR_HIP, EPI_ERR_UNSORTED, EPI_ERR_NOMEM, EPI_ERR_NODEVICE, EPI_ERR_STATE = range(7)


class EpihipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("epihip error %d: %s" % (code, msg))
        self.code = code


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_total", C.c_int64), ("row_first", C.c_int64), ("n", C.c_int64),
                ("read_len", C.c_int32), ("n_chr", C.c_int32), ("depth", C.c_int32),
                ("gap_from", C.c_int32), ("gap_len", C.c_int32)]


class BamOptions(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("min_mapq", "min_baseq", "skip_duplicates", "skip_secondary", "skip_qcfail",
                                         "skip_supplementary", "trim5", "trim3", "paired", "nthreads", "min_prob",
                                         "highest_prob", "window_kib")]


class Templates(C.Structure):
    _fields_ = [("n", C.c_int64), ("nbytes", C.c_int64), ("xm_capacity", C.c_int64), ("nrecs", C.c_int64),
                ("xm", C.POINTER(C.c_uint8)), ("off", C.POINTER(C.c_int64)),
                ("rname", C.POINTER(C.c_int32)), ("strand", C.POINTER(C.c_int32)), ("start", C.POINTER(C.c_int32)),
                ("n_targets", C.c_int32), ("target_names", C.POINTER(C.c_char_p)),
                ("paired", C.c_int32), ("pinned", C.c_int32)]


class CxTable(C.Structure):
    _fields_ = [("nrow", C.c_int64)] + [(k, C.POINTER(C.c_int32)) for k in
                                        ("rname", "strand", "pos", "context", "meth", "unmeth")]


class MhlTable(C.Structure):
    _fields_ = [("nrow", C.c_int64)] + [(k, C.POINTER(C.c_int32)) for k in
                                        ("rname", "strand", "pos", "context", "coverage")] + \
               [("length", C.POINTER(C.c_double)), ("lmhl", C.POINTER(C.c_double))]


class ReportColumn(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int32), ("data", C.c_void_p), ("levels", C.POINTER(C.c_char_p)),
                ("nlevels", C.c_int32)]


class PatternTable(C.Structure):
    _fields_ = [("npat", C.c_int64), ("ncol", C.c_int32), ("positions", C.POINTER(C.c_int32))] + \
               [(k, C.POINTER(C.c_int32)) for k in ("strand", "start", "end", "nbase")] + \
               [("beta", C.POINTER(C.c_double)), ("fnv", C.POINTER(C.c_uint64)), ("cells", C.POINTER(C.c_int32))]


def build(force=False):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8", "libepihip.so"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None

_VP, _I64, _I32, _U32, _F64, _CS = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_double, C.c_char_p

_SIGS = {
    "epi_last_error": (C.c_char_p, []),
    "epi_version": (C.c_int, []),
    "epi_cx_table_free": (None, [C.POINTER(CxTable)]),
    "epi_mhl_table_free": (None, [C.POINTER(MhlTable)]),
    "epi_threshold_reads": (C.c_int, [_VP, _VP, _I64, _CS, _CS, _CS, _CS, _U32, _F64, _F64, _VP]),
    "epi_get_xm_beta": (C.c_int, [_VP, _VP, _I64, _CS, _CS, _VP]),
    "epi_cx_report": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _CS, C.POINTER(CxTable)]),
    "epi_mhl_report": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I64, _CS, C.c_int, C.c_int, _F64, C.POINTER(MhlTable)]),
    "epi_preprocess_bam": (C.c_int, [_CS, C.POINTER(BamOptions), C.POINTER(Templates)]),
    "epi_templates_free": (None, [C.POINTER(Templates)]),
    "epi_write_report": (C.c_int, [_CS, C.POINTER(ReportColumn), _I32, _I64, _I32, _I32]),
    "epi_engine_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "epi_engine_destroy": (None, [_VP]),
    "epi_engine_device": (C.c_int, [_VP]),
    "epi_batch_upload": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, C.POINTER(_VP)]),
    "epi_batch_adopt": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _VP, _VP, _VP, _I64, C.POINTER(_VP)]),
    "epi_batch_free": (None, [_VP]),
    "epi_batch_threshold_reads": (C.c_int, [_VP, _CS, _CS, _CS, _CS, _U32, _F64, _F64, _VP]),
    "epi_batch_get_xm_beta": (C.c_int, [_VP, _CS, _CS, _VP]),
    "epi_batch_cx_report": (C.c_int, [_VP, _VP, _CS, C.POINTER(CxTable)]),
    "epi_batch_cytosine_report": (C.c_int, [_VP, _CS, _CS, _CS, _CS, _U32, _F64, _F64, _CS, _VP, C.POINTER(CxTable)]),
    "epi_batch_mhl_report": (C.c_int, [_VP, _CS, C.c_int, C.c_int, _F64, C.POINTER(MhlTable)]),
    "epi_batch_cx_report_begin": (C.c_int, [_VP, _VP, _CS, C.POINTER(_I64)]),
    "epi_batch_cytosine_report_begin": (C.c_int, [_VP, _CS, _CS, _CS, _CS, _U32, _F64, _F64, _CS, _VP, C.POINTER(_I64)]),
    "epi_batch_mhl_report_begin": (C.c_int, [_VP, _CS, C.c_int, C.c_int, _F64, C.POINTER(_I64)]),
    "epi_default_engine": (C.c_int, [C.POINTER(_VP)]),
    "epi_batch_nrows": (_I64, [_VP]),
    "epi_batch_realign": (C.c_int, [_VP, _VP]),
    "epi_batch_layout": (C.c_int, [_VP]),
    "epi_batch_view": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_I64)]),
    "epi_batch_threshold_reads_dev": (C.c_int, [_VP, _CS, _CS, _CS, _CS, _U32, _F64, _F64, _VP, _VP]),
    "epi_batch_get_xm_beta_dev": (C.c_int, [_VP, _CS, _CS, _VP, _VP]),
    "epi_batch_match_target_dev": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP]),
    "epi_batch_extract_patterns": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _CS, _F64, _I32, _I32, _VP, _I32, _VP,
                                           C.POINTER(PatternTable)]),
    "epi_pattern_table_free": (None, [C.POINTER(PatternTable)]),
    "epi_batch_cx_report_dev": (C.c_int, [_VP, _VP, _CS, _VP, C.POINTER(_I64)]),
    "epi_batch_cytosine_report_dev": (C.c_int, [_VP, _CS, _CS, _CS, _CS, _U32, _F64, _F64, _CS, _VP, _VP, C.POINTER(_I64)]),
    "epi_batch_cx_fetch_dev": (C.c_int, [_VP, C.POINTER(_VP), _VP]),
    "epi_batch_cx_fetch_host": (C.c_int, [_VP, C.POINTER(_VP), _VP]),
    "epi_batch_mhl_report_dev": (C.c_int, [_VP, _CS, C.c_int, C.c_int, _F64, _VP, C.POINTER(_I64)]),
    "epi_batch_mhl_fetch_dev": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), _VP]),
    "epi_batch_mhl_fetch_host": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), _VP]),
    "epi_tile_positions": (C.c_int, []),
    "epi_cx_tile_positions": (C.c_int, [_CS]),
    "epi_batch_tile_key_range": (C.c_int, [_VP, _VP, C.POINTER(_I64), C.POINTER(_I64)]),
    "epi_batch_cx_set_shared": (C.c_int, [_VP, _VP, _VP, _I32, _VP]),
    "epi_batch_cx_finish_shared": (C.c_int, [_VP, _CS, _VP, C.POINTER(_I64)]),
    "epi_mhl_tile_positions": (C.c_int, []),
    "epi_mhl_slab_sums": (C.c_int, []),
    "epi_batch_tile_key_range_for": (C.c_int, [_VP, C.c_int, _VP, C.POINTER(_I64), C.POINTER(_I64)]),
    "epi_batch_mhl_set_shared": (C.c_int, [_VP, _VP, _VP, _I32, _VP, _VP]),
    "epi_batch_mhl_finish_shared": (C.c_int, [_VP, _VP, C.POINTER(_I64)]),
    "epi_mhl_fused_tile_positions": (C.c_int, []),
    "epi_batch_mhl_fused_ok": (C.c_int, [_VP, _CS, _VP, C.POINTER(_I32)]),
    "epi_batch_mhl_set_shared_fused": (C.c_int, [_VP, _VP, _VP, _I32, _VP, _VP]),
    "epi_shared_tile_keys": (C.c_int, [_VP, _I32, _VP, _VP, _I32, C.POINTER(_I32)]),
    "epi_comm_unique_id": (C.c_int, [_VP]),
    "epi_comm_create": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "epi_comm_free": (None, [_VP]),
    "epi_comm_rank": (C.c_int, [_VP]),
    "epi_comm_world": (C.c_int, [_VP]),
    "epi_comm_last_exchange_bytes": (_I64, [_VP]),
    "epi_comm_set_test_shared": (None, [_VP, C.c_int]),
    "epi_batch_cytosine_report_sharded": (C.c_int, [_VP, _VP, _CS, _CS, _CS, _CS, _U32, _F64, _F64, _VP, _CS, _VP, _VP, C.POINTER(_I64)]),
    "epi_batch_mhl_report_sharded": (C.c_int, [_VP, _VP, _CS, C.c_int, C.c_int, _F64, _VP, C.POINTER(_I64)]),
    "epi_synth_generate_dev": (C.c_int, [C.POINTER(SynthParams), _VP, _VP, _VP, _VP, _VP, _VP]),
    "epi_synth_fill_dev": (C.c_int, [C.c_uint64, _I64, _I64, _VP, _VP, _VP, _I64, _I32, _I32, _VP, _VP, _VP]),
    "epi_prof_enable": (None, [C.c_int]),
    "epi_prof_get": (C.c_int, [_CS, C.POINTER(_F64), C.POINTER(_I64)]),
    "epi_prof_reset": (None, []),
    "epi_options_reload": (None, []),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())


def load():
    """Load libepihip.so (importing torch first so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EpihipError(EPI_ERR_NODEVICE,
                          "libepihip.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                          "there is no CPU fallback" % LIB_PATH)
    host_only = bool(os.environ.get("EPIHIP_HOST_ONLY"))     # `make asan` library: BAM producer + report writer only
    if not host_only:
        try:
            import torch  # noqa: F401  (loads torch's libamdhip64 first; ours resolves to the same SONAME)
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if host_only:
                continue
            raise EpihipError(EPI_ERR_STATE, "%s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != EPI_OK:
        msg = load().epi_last_error()
        raise EpihipError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")


def enc(s):
    return (s or "").encode("latin1")
