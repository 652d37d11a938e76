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
  SEXP cached = df.attr("seqxm_hip_xptr");
  if (cached != R_NilValue) return *Rcpp::XPtr<Resident>(cached);
  Rcpp::IntegerVector rname = df["rname"], strand = df["strand"], start = df["start"];
  const R_xlen_t n = rname.size();
  Resident *r = nullptr;
  SEXP soa = df.attr("seqxm_soa_xptr");
  Rcpp::checkUserInterrupt();
  try {
    if (soa != R_NilValue) {
      // rows of the producer's batch are in table order as long as the table was not re-ordered or subset
      Rcpp::XPtr<epihip_shim::TemplatesGuard> tg(soa);
      if (tg->t.n != (int64_t)n) Rcpp::stop("the preprocessed table was subset after reading: call preprocessBam again");
      r = Resident::upload(tg->t.xm, tg->t.off, rname.begin(), strand.begin(), start.begin(), n);
    } else {
      Rcpp::XPtr<std::vector<std::string>> seqxm((SEXP)df.attr("seqxm_xptr"));
      Rcpp::IntegerVector templid = df["templid"];
      epihip_shim::Soa s;
      epihip_shim::gather_rows(*seqxm, templid.begin(), (int64_t)n, s, []() { Rcpp::checkUserInterrupt(); });
      r = Resident::upload(s.xm.data(), s.off.data(), rname.begin(), strand.begin(), start.begin(), n);
    }
  } catch (const std::runtime_error &e) {
    Rcpp::stop("%s", e.what());
  }
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

Rcpp::DataFrame cx_frame(const epi_cx_table &t, Rcpp::DataFrame &df) {
  Rcpp::IntegerVector rname = df["rname"], strand = df["strand"];
  auto col = [&](const int32_t *p) { return Rcpp::IntegerVector(p, p + t.nrow); };
  Rcpp::DataFrame res = Rcpp::DataFrame::create(
      Rcpp::Named("rname") = col(t.rname), Rcpp::Named("strand") = col(t.strand), Rcpp::Named("pos") = col(t.pos),
      Rcpp::Named("context") = col(t.context), Rcpp::Named("meth") = col(t.meth), Rcpp::Named("unmeth") = col(t.unmeth));
  set_factors(res, rname, strand);
  return res;
}

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
  epihip_shim::CxTableGuard g;
  // an R logical vector is int32 with NA = INT_MIN; the ABI treats any non-zero value as TRUE (:118)
  check(epi_batch_cx_report(r.batch, pass.begin(), ctx.c_str(), &g.t));
  Rcpp::checkUserInterrupt();
  return cx_frame(g.t, df);
}

// [[Rcpp::export]]
Rcpp::DataFrame rcpp_mhl_report(Rcpp::DataFrame &df, const std::string ctx, int hmax, const int hmin,
                                const double max_ooctx_meth_frac) {
  Resident &r = resident_of(df);
  epihip_shim::MhlTableGuard g;
  check(epi_batch_mhl_report(r.batch, ctx.c_str(), hmax, hmin, max_ooctx_meth_frac, &g.t));
  Rcpp::checkUserInterrupt();
  const epi_mhl_table &t = g.t;
  Rcpp::IntegerVector rname = df["rname"], strand = df["strand"];
  auto icol = [&](const int32_t *p) { return Rcpp::IntegerVector(p, p + t.nrow); };
  auto dcol = [&](const double *p) { return Rcpp::NumericVector(p, p + t.nrow); };
  Rcpp::DataFrame res = Rcpp::DataFrame::create(
      Rcpp::Named("rname") = icol(t.rname), Rcpp::Named("strand") = icol(t.strand), Rcpp::Named("pos") = icol(t.pos),
      Rcpp::Named("context") = icol(t.context), Rcpp::Named("coverage") = icol(t.coverage),
      Rcpp::Named("length") = dcol(t.length), Rcpp::Named("lmhl") = dcol(t.lmhl));
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
  epihip_shim::CxTableGuard g;
  check(epi_batch_cytosine_report(r.batch, ctx_meth.c_str(), ctx_unmeth.c_str(), ooctx_meth.c_str(), ooctx_unmeth.c_str(),
                                  min_n_ctx, min_ctx_meth_frac, max_ooctx_meth_frac, ctx.c_str(), nullptr, &g.t));
  Rcpp::checkUserInterrupt();
  return cx_frame(g.t, df);
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
