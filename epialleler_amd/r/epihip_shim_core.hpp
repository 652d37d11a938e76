// The SEXP-free half of the Rcpp shim (epihip_shim.cpp): everything between R's objects and the C ABI of
// libepihip.so that does not need Rcpp -- so it can be compiled and tested without R (tests/cpp/test_shim_core.cpp).
//
//  * gather_rows        std::vector<std::string> behind seqxm_xptr + templid  ->  one byte stream + offsets, row order
//  * Resident           what the new attribute `seqxm_hip_xptr` points at: the batch in HBM, uploaded ONCE per
//                       preprocessBam() object and reused by every later call (threshold -> cx without re-upload)
//  * bam_options        .readBam's arguments (R/internal.R:154-199: skip.flags as a sum) -> epi_bam_options
//  * materialize        SoA -> std::vector<std::string> for the out-of-scope functions that still read seqxm_xptr
//                       (rcpp_extract_patterns, rcpp_match_*, rcpp_get_base_freqs)
//  * gather_soa, RowOrder  the same row semantics for the producer's SoA, and the staleness check of the cached batch
//  * *_report_into      the report tables copied straight into columns the caller owns (R vectors): one D2H copy
//  * table guards       epi_cx_table / epi_mhl_table of the one-call entry points stay library-owned until freed
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdexcept>
#include <string>
#include <vector>
#include "epihip.h"

namespace epihip_shim {

struct Soa {
  std::vector<uint8_t> xm;
  std::vector<int64_t> off;
};

// row x is seqxm.at(templid[x]) (src/rcpp_cx_report.cpp:119); `poll` is called every 2^20 rows (interrupt check)
template <class Poll>
inline void gather_rows(const std::vector<std::string> &seqxm, const int32_t *templid, int64_t n, Soa &s, Poll poll) {
  s.off.resize((size_t)n + 1);
  int64_t total = 0;
  for (int64_t x = 0; x < n; x++) { s.off[(size_t)x] = total; total += (int64_t)seqxm.at((size_t)templid[x]).size(); }
  s.off[(size_t)n] = total;
  s.xm.resize((size_t)total + 16);
  for (int64_t x = 0; x < n; x++) {
    const std::string &t = seqxm[(size_t)templid[x]];
    if (!t.empty()) memcpy(s.xm.data() + s.off[(size_t)x], t.data(), t.size());
    if ((x & 0xFFFFF) == 0) poll();
  }
}

// The same for a table that still has the producer's SoA behind `seqxm_soa_xptr`: row x is template templid[x] of the
// producer's batch (what seqxm->at(templid[x]) means for the reference), so a table that was re-ordered by reference
// (setorder / setkey keep attributes), subset, or holds a template twice is served like the reference serves it.
template <class Poll>
inline void gather_soa(const epi_templates &t, const int32_t *templid, int64_t n, Soa &s, Poll poll) {
  s.off.resize((size_t)n + 1);
  int64_t total = 0;
  for (int64_t x = 0; x < n; x++) {
    const int64_t id = templid[x];
    if (id < 0 || id >= t.n) throw std::out_of_range("templid outside the preprocessed templates");   // vector::at
    s.off[(size_t)x] = total;
    total += t.off[id + 1] - t.off[id];
  }
  s.off[(size_t)n] = total;
  s.xm.resize((size_t)total + 16);
  for (int64_t x = 0; x < n; x++) {
    const int64_t id = templid[x], len = t.off[id + 1] - t.off[id];
    if (len > 0) memcpy(s.xm.data() + s.off[(size_t)x], t.xm + t.off[id], (size_t)len);
    if ((x & 0xFFFFF) == 0) poll();
  }
}

// What a resident batch was built from: the row count and the `templid` column (identity 0..n-1, or a hash of it).  A
// cached handle is reused only while the data.frame still has exactly this row order (a by-reference setorder or a subset
// that kept the attributes changes it).
struct RowOrder {
  int64_t n = -1;
  bool identity = false;
  uint64_t hash = 0;
  bool operator==(const RowOrder &o) const { return n == o.n && identity == o.identity && hash == o.hash; }
  bool operator!=(const RowOrder &o) const { return !(*this == o); }
};
inline RowOrder row_order_of(const int32_t *templid, int64_t n) {
  RowOrder r;
  r.n = n;
  r.identity = true;
  int64_t x = 0;
  for (; x < n; x++) if (templid[x] != (int32_t)x) { r.identity = false; break; }
  if (!r.identity) {
    uint64_t h = 1469598103934665603ull;                   // FNV-1a over the column
    for (x = 0; x < n; x++) { h ^= (uint32_t)templid[x]; h *= 1099511628211ull; }
    r.hash = h;
  }
  return r;
}

inline void check(int rc) {                      // the Rcpp shim turns this into Rcpp::stop (BEGIN_RCPP / END_RCPP)
  if (rc != EPI_OK) throw std::runtime_error(epi_last_error());
}

// The batch of one preprocessBam() object, resident in HBM.  Owned by an R external pointer (finalizer = delete).
struct Resident {
  epi_batch *batch = nullptr;
  int64_t n = 0;
  RowOrder order;                                  // the table rows this batch was uploaded from
  Resident() = default;
  Resident(const Resident &) = delete;
  Resident &operator=(const Resident &) = delete;
  ~Resident() { if (batch) epi_batch_free(batch); }

  // host SoA (gathered strings, or the producer's pinned buffers) -> HBM, once
  static Resident *upload(const uint8_t *xm, const int64_t *off, const int32_t *rname, const int32_t *strand,
                          const int32_t *start, int64_t n) {
    epi_engine *eng = nullptr;
    check(epi_default_engine(&eng));
    Resident *r = new Resident();
    const int rc = epi_batch_upload(eng, xm, off, rname, strand, start, n, &r->batch);
    if (rc != EPI_OK) { delete r; check(rc); }
    r->n = n;
    return r;
  }
};

// The report tables straight into columns the caller owns (R's IntegerVector / NumericVector): `alloc(nrow)` is called
// once the row count is known and returns the destinations.  One device-to-host copy, no library-owned table in between
// (the reference fills its vectors in place too, src/rcpp_cx_report.cpp:133-140).
template <class Alloc6>
inline int64_t cx_report_into(epi_batch *b, const int32_t *pass, const char *ctx, Alloc6 alloc) {
  int64_t nrow = 0;
  check(epi_batch_cx_report_begin(b, pass, ctx, &nrow));
  int32_t *cols[6];
  alloc(nrow, cols);
  if (nrow > 0) check(epi_batch_cx_fetch_host(b, cols, nullptr));
  return nrow;
}
template <class Alloc6>
inline int64_t cytosine_report_into(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                                    const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                                    double max_ooctx_meth_frac, const char *ctx, int32_t *pass_out, Alloc6 alloc) {
  int64_t nrow = 0;
  check(epi_batch_cytosine_report_begin(b, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n_ctx, min_ctx_meth_frac,
                                        max_ooctx_meth_frac, ctx, pass_out, &nrow));
  int32_t *cols[6];
  alloc(nrow, cols);
  if (nrow > 0) check(epi_batch_cx_fetch_host(b, cols, nullptr));
  return nrow;
}
template <class Alloc7>
inline int64_t mhl_report_into(epi_batch *b, const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac, Alloc7 alloc) {
  int64_t nrow = 0;
  check(epi_batch_mhl_report_begin(b, ctx, hmax, hmin, max_ooctx_meth_frac, &nrow));
  int32_t *ic[5];
  double *dc[2];
  alloc(nrow, ic, dc);
  if (nrow > 0) check(epi_batch_mhl_fetch_host(b, ic, dc, nullptr));
  return nrow;
}

// .readBam's numeric arguments -> epi_bam_options (skip.flags = 4 [+256] [+512] [+1024] [+2048] [+8 when paired])
inline epi_bam_options bam_options(int min_mapq, int min_baseq, int skip_flags, int trim5, int trim3, int nthreads,
                                   int paired, int min_prob = -1, bool highest_prob = true) {
  epi_bam_options o;
  memset(&o, 0, sizeof(o));
  o.min_mapq = min_mapq;
  o.min_baseq = min_baseq;
  o.skip_secondary = (skip_flags & 256) != 0;
  o.skip_qcfail = (skip_flags & 512) != 0;
  o.skip_duplicates = (skip_flags & 1024) != 0;
  o.skip_supplementary = (skip_flags & 2048) != 0;
  o.trim5 = trim5;
  o.trim3 = trim3;
  o.paired = paired;
  o.nthreads = nthreads < 1 ? 1 : nthreads;
  o.min_prob = min_prob;
  o.highest_prob = highest_prob ? 1 : 0;
  return o;
}

// the std::vector<std::string> view of a producer batch (rows are already in (rname,start) order: templid = 0..n-1)
inline void materialize(const epi_templates &t, std::vector<std::string> &seqxm) {
  seqxm.clear();
  seqxm.reserve((size_t)t.n);
  for (int64_t x = 0; x < t.n; x++)
    seqxm.emplace_back(reinterpret_cast<const char *>(t.xm) + t.off[x], (size_t)(t.off[x + 1] - t.off[x]));
}

// frees a library-owned table when it goes out of scope (also when copying into R vectors throws)
struct CxTableGuard {
  epi_cx_table t;
  CxTableGuard() { memset(&t, 0, sizeof(t)); }
  ~CxTableGuard() { epi_cx_table_free(&t); }
};
struct MhlTableGuard {
  epi_mhl_table t;
  MhlTableGuard() { memset(&t, 0, sizeof(t)); }
  ~MhlTableGuard() { epi_mhl_table_free(&t); }
};
struct TemplatesGuard {                       // owner behind `seqxm_soa_xptr`
  epi_templates t;
  TemplatesGuard() { memset(&t, 0, sizeof(t)); }
  TemplatesGuard(const TemplatesGuard &) = delete;
  TemplatesGuard &operator=(const TemplatesGuard &) = delete;
  ~TemplatesGuard() { epi_templates_free(&t); }
};

}  // namespace epihip_shim
