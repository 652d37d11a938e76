// Rcpp shim: the four hot-path exports of epialleleR re-implemented as thin calls
// into libepihip.so (include/epihip.h).  Drop these definitions in place of
// src/rcpp_threshold_reads.cpp, src/rcpp_get_xm_beta.cpp, src/rcpp_cx_report.cpp
// and src/rcpp_mhl_report.cpp: the [[Rcpp::export]] names and signatures are the
// reference's, so R/RcppExports.R, src/RcppExports.cpp and every R caller stay
// unchanged.  NOT compiled in this repository's image (no R,
// Rcpp or HTSlib here); kept as the reference-side binding a maintainer adds.
//
// PKG_LIBS   += -L<prefix>/lib -lepihip -Wl,-rpath,<prefix>/lib
// PKG_CPPFLAGS += -I<prefix>/include
#include <Rcpp.h>
#include <cstring>
#include <string>
#include <vector>
#include "epihip.h"

namespace {

// Gathers the templates, in ROW order, into the SoA layout of the C ABI:
// row x is seqxm->at(templid[x]) (src/rcpp_cx_report.cpp:119).
struct Soa {
  std::vector<uint8_t> xm;
  std::vector<int64_t> off;
};

Soa gather_rows(Rcpp::DataFrame &df, R_xlen_t n) {
  Rcpp::XPtr<std::vector<std::string>> seqxm((SEXP)df.attr("seqxm_xptr"));
  Rcpp::IntegerVector templid = df["templid"];
  Soa s;
  s.off.resize((size_t)n + 1);
  int64_t total = 0;
  for (R_xlen_t x = 0; x < n; x++) { s.off[x] = total; total += (int64_t)seqxm->at(templid[x]).size(); }
  s.off[n] = total;
  s.xm.resize((size_t)total + 16);
  for (R_xlen_t x = 0; x < n; x++) {
    const std::string &t = seqxm->at(templid[x]);
    std::memcpy(s.xm.data() + s.off[x], t.data(), t.size());
    if ((x & 0xFFFFF) == 0) Rcpp::checkUserInterrupt();
  }
  return s;
}

void check(int rc) { if (rc != EPI_OK) Rcpp::stop("%s", epi_last_error()); }

}  // namespace

// [[Rcpp::export("rcpp_threshold_reads")]]
std::vector<bool> rcpp_threshold_reads(Rcpp::DataFrame &df, const std::string ctx_meth, const std::string ctx_unmeth,
                                       const std::string ooctx_meth, const std::string ooctx_unmeth,
                                       const unsigned int min_n_ctx, const double min_ctx_meth_frac,
                                       const double max_ooctx_meth_frac) {
  Rcpp::XPtr<std::vector<std::string>> seqxm((SEXP)df.attr("seqxm_xptr"));
  const R_xlen_t n = (R_xlen_t)seqxm->size();           // the reference iterates seqxm->size() (:28)
  Soa s = gather_rows(df, n);
  std::vector<int32_t> pass((size_t)n + 1);
  check(epi_threshold_reads(s.xm.data(), s.off.data(), n, ctx_meth.c_str(), ctx_unmeth.c_str(), ooctx_meth.c_str(),
                            ooctx_unmeth.c_str(), min_n_ctx, min_ctx_meth_frac, max_ooctx_meth_frac, pass.data()));
  std::vector<bool> res((size_t)n);
  for (R_xlen_t x = 0; x < n; x++) res[x] = pass[x] != 0;
  return res;
}

// [[Rcpp::export("rcpp_get_xm_beta")]]
std::vector<double> rcpp_get_xm_beta(Rcpp::DataFrame &df, const std::string ctx_meth, const std::string ctx_unmeth) {
  Rcpp::XPtr<std::vector<std::string>> seqxm((SEXP)df.attr("seqxm_xptr"));
  const R_xlen_t n = (R_xlen_t)seqxm->size();
  Soa s = gather_rows(df, n);
  std::vector<double> res((size_t)n);
  check(epi_get_xm_beta(s.xm.data(), s.off.data(), n, ctx_meth.c_str(), ctx_unmeth.c_str(), res.data()));
  return res;
}

static void set_factors(Rcpp::DataFrame &res, Rcpp::IntegerVector &rname, Rcpp::IntegerVector &strand) {
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

// [[Rcpp::export("rcpp_cx_report")]]
Rcpp::DataFrame rcpp_cx_report(Rcpp::DataFrame &df, Rcpp::LogicalVector &pass, const std::string ctx) {
  Rcpp::IntegerVector rname = df["rname"], strand = df["strand"], start = df["start"];
  const R_xlen_t n = rname.size();
  Soa s = gather_rows(df, n);
  epi_cx_table t;
  // an R logical vector is int32 with NA = INT_MIN; the ABI treats any non-zero value as TRUE (:118)
  check(epi_cx_report(s.xm.data(), s.off.data(), rname.begin(), strand.begin(), start.begin(), pass.begin(), n,
                      ctx.c_str(), &t));
  auto col = [&](const int32_t *p) { return Rcpp::IntegerVector(p, p + t.nrow); };
  Rcpp::DataFrame res = Rcpp::DataFrame::create(
      Rcpp::Named("rname") = col(t.rname), Rcpp::Named("strand") = col(t.strand), Rcpp::Named("pos") = col(t.pos),
      Rcpp::Named("context") = col(t.context), Rcpp::Named("meth") = col(t.meth), Rcpp::Named("unmeth") = col(t.unmeth));
  epi_cx_table_free(&t);
  set_factors(res, rname, strand);
  return res;
}

// [[Rcpp::export]]
Rcpp::DataFrame rcpp_mhl_report(Rcpp::DataFrame &df, const std::string ctx, int hmax, const int hmin,
                                const double max_ooctx_meth_frac) {
  Rcpp::IntegerVector rname = df["rname"], strand = df["strand"], start = df["start"];
  const R_xlen_t n = rname.size();
  Soa s = gather_rows(df, n);
  epi_mhl_table t;
  check(epi_mhl_report(s.xm.data(), s.off.data(), rname.begin(), strand.begin(), start.begin(), n, ctx.c_str(), hmax, hmin,
                       max_ooctx_meth_frac, &t));
  auto icol = [&](const int32_t *p) { return Rcpp::IntegerVector(p, p + t.nrow); };
  auto dcol = [&](const double *p) { return Rcpp::NumericVector(p, p + t.nrow); };
  Rcpp::DataFrame res = Rcpp::DataFrame::create(
      Rcpp::Named("rname") = icol(t.rname), Rcpp::Named("strand") = icol(t.strand), Rcpp::Named("pos") = icol(t.pos),
      Rcpp::Named("context") = icol(t.context), Rcpp::Named("coverage") = icol(t.coverage),
      Rcpp::Named("length") = dcol(t.length), Rcpp::Named("lmhl") = dcol(t.lmhl));
  epi_mhl_table_free(&t);
  set_factors(res, rname, strand);
  return res;
}
