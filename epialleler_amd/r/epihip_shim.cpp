// Rcpp shim: epialleleR's hot-path exports re-implemented as thin calls into libepihip.so (include/epihip.h).
// Drop these definitions in place of src/rcpp_threshold_reads.cpp, src/rcpp_get_xm_beta.cpp, src/rcpp_cx_report.cpp,
// src/rcpp_mhl_report.cpp and (optionally) the three readers of src/rcpp_read_bam.cpp: the [[Rcpp::export]] names and
// signatures are the reference's, so R/RcppExports.R, src/RcppExports.cpp and every R caller stay unchanged (see
// INTEGRATION.md).  NOT compiled in this repository's image (no R, Rcpp or HTSlib here); everything that does not
// touch an SEXP lives in epihip_shim_core.hpp, which IS compiled and tested here (tests/cpp/test_shim_core.cpp).
//
// PKG_LIBS     += -L<prefix>/lib -lepihip -Wl,-rpath,<prefix>/lib
// PKG_CPPFLAGS += -I<prefix>/include
//
// Residency: the packed templates of a preprocessBam() object are uploaded to HBM ONCE -- by the first call that needs
// them -- and the handle is cached on the data.frame as the attribute `seqxm_hip_xptr` (an external pointer whose
// finalizer frees the device memory).  rcpp_threshold_reads followed by rcpp_cx_report, or any number of reports on one
// preprocessed object, therefore move the bytes over PCIe one time.  `seqxm_xptr` keeps its type and meaning.
#include <Rcpp.h>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "epihip.h"
#include "epihip_shim_core.hpp"

using epihip_shim::Resident;

namespace {

void check(int rc) { if (rc != EPI_OK) Rcpp::stop("%s", epi_last_error()); }

// The resident batch of `df`: cached handle, else upload from the producer's pinned SoA (`seqxm_soa_xptr`, set by the
// rcpp_read_bam_* shims below), else gather the strings behind `seqxm_xptr` in row order and upload those.
Resident &resident_of(Rcpp::DataFrame &df) {
  Rcpp::IntegerVector rname = df["rname"], strand = df["strand"], start = df["start"], templid = df["templid"];
  const R_xlen_t n = rname.size();
  // the rows this call is about: the cached batch is reused only if it was built from exactly these (ADVICE round 2:
  // a by-reference setorder / setkey or a subset that kept its attributes would otherwise be served the old rows)
  const epihip_shim::RowOrder order = epihip_shim::row_order_of(templid.begin(), (int64_t)n);
  SEXP cached = df.attr("seqxm_hip_xptr");
  if (cached != R_NilValue) {
    Rcpp::XPtr<Resident> xp(cached);
    if (xp->order == order) return *xp;
    df.attr("seqxm_hip_xptr") = R_NilValue;                // stale: rebuild below (the old batch goes with its finalizer)
  }
  Resident *r = nullptr;
  SEXP soa = df.attr("seqxm_soa_xptr");
  Rcpp::checkUserInterrupt();
  try {
    if (soa != R_NilValue) {
      Rcpp::XPtr<epihip_shim::TemplatesGuard> tg(soa);
      if (order.identity && tg->t.n == (int64_t)n) {
        // rows of the producer's batch are the table's rows: upload in place (pinned source, no staging copy)
        r = Resident::upload(tg->t.xm, tg->t.off, rname.begin(), strand.begin(), start.begin(), n);
      } else {
        // re-ordered / subset table: row x is template templid[x], as seqxm->at(templid[x]) is for the reference
        epihip_shim::Soa s;
        epihip_shim::gather_soa(tg->t, templid.begin(), (int64_t)n, s, []() { Rcpp::checkUserInterrupt(); });
        r = Resident::upload(s.xm.data(), s.off.data(), rname.begin(), strand.begin(), start.begin(), n);
      }
    } else {
      Rcpp::XPtr<std::vector<std::string>> seqxm((SEXP)df.attr("seqxm_xptr"));
      epihip_shim::Soa s;
      epihip_shim::gather_rows(*seqxm, templid.begin(), (int64_t)n, s, []() { Rcpp::checkUserInterrupt(); });
      r = Resident::upload(s.xm.data(), s.off.data(), rname.begin(), strand.begin(), start.begin(), n);
    }
  } catch (const std::exception &e) {
    Rcpp::stop("%s", e.what());
  }
  r->order = order;
  Rcpp::XPtr<Resident> xp(r, true);                      // finalizer: ~Resident -> epi_batch_free
  df.attr("seqxm_hip_xptr") = xp;
  Rcpp::checkUserInterrupt();
  return *r;
}

void set_factors(Rcpp::DataFrame &res, Rcpp::IntegerVector &rname, Rcpp::IntegerVector &strand) {
  Rcpp::IntegerVector col_rname = res["rname"];          // src/rcpp_cx_report.cpp:142-155
  col_rname.attr("class") = "factor";
  col_rname.attr("levels") = rname.attr("levels");
  Rcpp::IntegerVector col_strand = res["strand"];
  col_strand.attr("class") = "factor";
  col_strand.attr("levels") = strand.attr("levels");
  Rcpp::IntegerVector col_context = res["context"];
  col_context.attr("class") = "factor";
  col_context.attr("levels") = Rcpp::CharacterVector::create("NA1", "CHH", "NA3", "NA4", "NA5", "CHG", "CG");
}

// the six columns of a CX table as R vectors the library fills in place (src/rcpp_cx_report.cpp:133-140)
struct CxColumns {
  Rcpp::IntegerVector v[6];
  void operator()(int64_t nrow, int32_t *(&cols)[6]) {
    for (int i = 0; i < 6; i++) { v[i] = Rcpp::IntegerVector(Rcpp::no_init((R_xlen_t)nrow)); cols[i] = v[i].begin(); }
  }
  Rcpp::DataFrame frame(Rcpp::DataFrame &df) {
    Rcpp::IntegerVector rname = df["rname"], strand = df["strand"];
    Rcpp::DataFrame res = Rcpp::DataFrame::create(
        Rcpp::Named("rname") = v[0], Rcpp::Named("strand") = v[1], Rcpp::Named("pos") = v[2],
        Rcpp::Named("context") = v[3], Rcpp::Named("meth") = v[4], Rcpp::Named("unmeth") = v[5]);
    set_factors(res, rname, strand);
    return res;
  }
};

}  // namespace

// ---- the four hot-path exports (same names and signatures as the reference) ----------------------------------------

// [[Rcpp::export("rcpp_threshold_reads")]]
std::vector<bool> rcpp_threshold_reads(Rcpp::DataFrame &df, const std::string ctx_meth, const std::string ctx_unmeth,
                                       const std::string ooctx_meth, const std::string ooctx_unmeth,
                                       const unsigned int min_n_ctx, const double min_ctx_meth_frac,
                                       const double max_ooctx_meth_frac) {
  Resident &r = resident_of(df);
  std::vector<int32_t> pass((size_t)r.n + 1);
  check(epi_batch_threshold_reads(r.batch, ctx_meth.c_str(), ctx_unmeth.c_str(), ooctx_meth.c_str(), ooctx_unmeth.c_str(),
                                  min_n_ctx, min_ctx_meth_frac, max_ooctx_meth_frac, pass.data()));
  Rcpp::checkUserInterrupt();
  std::vector<bool> res((size_t)r.n);
  for (int64_t x = 0; x < r.n; x++) res[(size_t)x] = pass[(size_t)x] != 0;
  return res;
}

// [[Rcpp::export("rcpp_get_xm_beta")]]
std::vector<double> rcpp_get_xm_beta(Rcpp::DataFrame &df, const std::string ctx_meth, const std::string ctx_unmeth) {
  Resident &r = resident_of(df);
  std::vector<double> res((size_t)r.n);
  check(epi_batch_get_xm_beta(r.batch, ctx_meth.c_str(), ctx_unmeth.c_str(), res.data()));
  return res;
}

// [[Rcpp::export("rcpp_cx_report")]]
Rcpp::DataFrame rcpp_cx_report(Rcpp::DataFrame &df, Rcpp::LogicalVector &pass, const std::string ctx) {
  Resident &r = resident_of(df);
  if (pass.size() != r.n) Rcpp::stop("pass must have one entry per row");
  CxColumns cols;
  // an R logical vector is int32 with NA = INT_MIN; the ABI treats any non-zero value as TRUE (:118)
  try { epihip_shim::cx_report_into(r.batch, pass.begin(), ctx.c_str(), std::ref(cols)); } catch (const std::exception &e) { Rcpp::stop("%s", e.what()); }
  Rcpp::checkUserInterrupt();
  return cols.frame(df);
}

// [[Rcpp::export]]
Rcpp::DataFrame rcpp_mhl_report(Rcpp::DataFrame &df, const std::string ctx, int hmax, const int hmin,
                                const double max_ooctx_meth_frac) {
  Resident &r = resident_of(df);
  Rcpp::IntegerVector iv[5];
  Rcpp::NumericVector dv[2];
  auto alloc = [&](int64_t nrow, int32_t *(&ic)[5], double *(&dc)[2]) {
    for (int i = 0; i < 5; i++) { iv[i] = Rcpp::IntegerVector(Rcpp::no_init((R_xlen_t)nrow)); ic[i] = iv[i].begin(); }
    for (int i = 0; i < 2; i++) { dv[i] = Rcpp::NumericVector(Rcpp::no_init((R_xlen_t)nrow)); dc[i] = dv[i].begin(); }
  };
  try { epihip_shim::mhl_report_into(r.batch, ctx.c_str(), hmax, hmin, max_ooctx_meth_frac, alloc); } catch (const std::exception &e) { Rcpp::stop("%s", e.what()); }
  Rcpp::checkUserInterrupt();
  Rcpp::IntegerVector rname = df["rname"], strand = df["strand"];
  Rcpp::DataFrame res = Rcpp::DataFrame::create(
      Rcpp::Named("rname") = iv[0], Rcpp::Named("strand") = iv[1], Rcpp::Named("pos") = iv[2],
      Rcpp::Named("context") = iv[3], Rcpp::Named("coverage") = iv[4],
      Rcpp::Named("length") = dv[0], Rcpp::Named("lmhl") = dv[1]);
  set_factors(res, rname, strand);
  return res;
}

// ---- optional: thresholding + report in one pass over the bytes ----------------------------------------------------
// generateCytosineReport(threshold.reads=TRUE) calls .thresholdReads and then .getCytosineReport with its result
// (R/generateCytosineReport.R:181-199).  With this export the two lines become one
//   cx.report <- rcpp_cytosine_report(bam, ctx.meth, ctx.unmeth, ooctx.meth, ooctx.unmeth, min.n, min.beta, max.oobeta, ctx)
// and the tile kernel decides every read from the bytes it loads anyway.

// [[Rcpp::export]]
Rcpp::DataFrame rcpp_cytosine_report(Rcpp::DataFrame &df, const std::string ctx_meth, const std::string ctx_unmeth,
                                     const std::string ooctx_meth, const std::string ooctx_unmeth,
                                     const unsigned int min_n_ctx, const double min_ctx_meth_frac,
                                     const double max_ooctx_meth_frac, const std::string ctx) {
  Resident &r = resident_of(df);
  CxColumns cols;
  try {
    epihip_shim::cytosine_report_into(r.batch, ctx_meth.c_str(), ctx_unmeth.c_str(), ooctx_meth.c_str(), ooctx_unmeth.c_str(),
                                      min_n_ctx, min_ctx_meth_frac, max_ooctx_meth_frac, ctx.c_str(), nullptr, std::ref(cols));
  } catch (const std::exception &e) { Rcpp::stop("%s", e.what()); }
  Rcpp::checkUserInterrupt();
  return cols.frame(df);
}

// ---- the readers (src/rcpp_read_bam.cpp:19-579) over the library's producer ------------------------------------------
// epi_preprocess_bam decodes the file (zlib BGZF reader, the reference's packers on `nthreads` threads) straight into
// the SoA layout, rows already in (rname,start) order, xm in pinned host memory.  .readBam then adds templid = 0..N-1
// and its setorder(rname, start) finds the rows in place (R/internal.R:193-195).  The data.frame carries
//   seqxm_soa_xptr  the producer's buffers (what the hot path uploads from, no gather), and
//   seqxm_xptr      the reference's std::vector<std::string>, filled here when `keep_strings` is TRUE (the functions
//                   outside the hot path -- rcpp_extract_patterns, rcpp_match_*, rcpp_get_base_freqs -- read it) or by
//                   rcpp_hip_materialize_seqxm(df) on first need.
namespace {

Rcpp::DataFrame read_bam(std::string fn, const epi_bam_options &opt) {
  Rcpp::XPtr<epihip_shim::TemplatesGuard> tg(new epihip_shim::TemplatesGuard(), true);
  const int rc = epi_preprocess_bam(fn.c_str(), &opt, &tg->t);
  if (rc != EPI_OK) Rcpp::stop("%s", epi_last_error());          // e.g. "Unable to open BAM file for reading" (:34)
  const epi_templates &t = tg->t;
  Rcpp::checkUserInterrupt();
  Rcpp::DataFrame res = Rcpp::DataFrame::create(
      Rcpp::Named("rname") = Rcpp::IntegerVector(t.rname, t.rname + t.n),
      Rcpp::Named("strand") = Rcpp::IntegerVector(t.strand, t.strand + t.n),
      Rcpp::Named("start") = Rcpp::IntegerVector(t.start, t.start + t.n));
  Rcpp::CharacterVector chromosomes(t.n_targets);
  for (int32_t i = 0; i < t.n_targets; i++) chromosomes[i] = t.target_names[i];
  Rcpp::IntegerVector col_rname = res["rname"];                   // :173-183
  col_rname.attr("class") = "factor";
  col_rname.attr("levels") = chromosomes;
  Rcpp::IntegerVector col_strand = res["strand"];
  col_strand.attr("class") = "factor";
  col_strand.attr("levels") = Rcpp::CharacterVector::create("+", "-");
  std::vector<std::string> *seqxm = new std::vector<std::string>();
  Rcpp::XPtr<std::vector<std::string>> seqxm_xptr(seqxm, true);
  if (Rcpp::as<bool>(Rcpp::Function("getOption")("epialleleR.keep.strings", false))) epihip_shim::materialize(t, *seqxm);
  res.attr("seqxm_xptr") = seqxm_xptr;                            // :185-186
  res.attr("seqxm_soa_xptr") = tg;
  res.attr("nrecs") = (double)t.nrecs;                            // :188-189
  res.attr("npushed") = (double)t.n;
  return res;
}

}  // namespace

// [[Rcpp::export]]
Rcpp::DataFrame rcpp_read_bam_paired(std::string fn, int min_mapq, int min_baseq, int skip_flags, int trim5, int trim3, int nthreads) {
  return read_bam(fn, epihip_shim::bam_options(min_mapq, min_baseq, skip_flags, trim5, trim3, nthreads, /*paired*/ 1));
}

// [[Rcpp::export]]
Rcpp::DataFrame rcpp_read_bam_single(std::string fn, int min_mapq, int min_baseq, int skip_flags, int trim5, int trim3, int nthreads) {
  return read_bam(fn, epihip_shim::bam_options(min_mapq, min_baseq, skip_flags, trim5, trim3, nthreads, /*paired*/ 0));
}

// [[Rcpp::export]]
Rcpp::DataFrame rcpp_read_bam_mm_single(std::string fn, int min_mapq, int min_baseq, int min_prob, bool highest_prob,
                                        int skip_flags, int trim5, int trim3, int nthreads) {
  return read_bam(fn, epihip_shim::bam_options(min_mapq, min_baseq, skip_flags, trim5, trim3, nthreads, /*paired*/ 0, min_prob, highest_prob));
}

// Fills the std::vector<std::string> behind seqxm_xptr from the SoA when it is still empty (for the functions outside
// the hot path; the R wrappers of those call this first -- three one-line edits).
// [[Rcpp::export]]
void rcpp_hip_materialize_seqxm(Rcpp::DataFrame &df) {
  SEXP soa = df.attr("seqxm_soa_xptr");
  if (soa == R_NilValue) return;                                  // a table of the reference's own readers: nothing to do
  Rcpp::XPtr<std::vector<std::string>> seqxm((SEXP)df.attr("seqxm_xptr"));
  Rcpp::XPtr<epihip_shim::TemplatesGuard> tg(soa);
  if (seqxm->empty() && tg->t.n > 0) epihip_shim::materialize(tg->t, *seqxm);
}
