"""epialleler_amd -- MI355X-native engine for epialleleR's per-read methylation-call
aggregation hot path (rcpp_threshold_reads / rcpp_get_xm_beta / rcpp_cx_report /
rcpp_mhl_report) behind the reference's own R-level interface.

Compute lives in csrc/ (hand-written HIP for gfx950 behind the C ABI of
include/epihip.h); this package is the host-side mirror of the R functions.
"""
from .api import (CONTEXT_TO_BASES, CONTEXT_LEVELS, STRAND_LEVELS, ProcessedBam, Report,  # noqa: F401
                  cytosine_report_fused, generateCytosineReport, generateMhlReport, preprocessBam, rcpp_cx_report,
                  rcpp_extract_patterns, rcpp_get_xm_beta, rcpp_mhl_report, rcpp_threshold_reads, writeReport)
from .bed import (Bed, Ecdf, extractPatterns, generateAmpliconReport, generateBedEcdf, generateBedReport,  # noqa: F401
                  generateCaptureReport, readBed)
from ._lib import EpihipError  # noqa: F401
