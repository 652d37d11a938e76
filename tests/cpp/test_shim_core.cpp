// The SEXP-free half of the Rcpp shim (epialleler_amd/r/epihip_shim_core.hpp) driven from plain C++: what the R
// binding does between R's objects and the C ABI, without R.
//   test_shim_core cpu <bam>            gather / options / materialize / producer, no GPU needed
//   test_shim_core gpu <bam> <outdir>   the resident flow: upload once, threshold -> cx (pass vector), the fused
//                                       one-pass report, lMHL; tables are written as raw little-endian columns for
//                                       the Python test to compare with the oracle
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <numeric>
#include <string>
#include <vector>
#include "epihip.h"
#include "epihip_shim_core.hpp"

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static int dump(const std::string &path, const void *p, size_t bytes) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return 1;
  const size_t w = bytes ? fwrite(p, 1, bytes, f) : 0;
  fclose(f);
  return w == bytes ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: test_shim_core cpu|gpu <bam> [outdir]\n"); return 2; }
  const std::string mode = argv[1];
  using namespace epihip_shim;

  // .readBam's arguments -> options; the producer; the string view the out-of-scope functions read
  epi_bam_options opt = bam_options(/*min_mapq*/ 0, /*min_baseq*/ 0, /*skip_flags*/ 4 + 256 + 512 + 2048 + 8, 0, 0, /*nthreads*/ 2, /*paired*/ -1);
  REQUIRE(opt.skip_secondary == 1 && opt.skip_qcfail == 1 && opt.skip_duplicates == 0 && opt.skip_supplementary == 1);
  TemplatesGuard tg;
  if (epi_preprocess_bam(argv[2], &opt, &tg.t) != EPI_OK) { fprintf(stderr, "%s\n", epi_last_error()); return 1; }
  const epi_templates &t = tg.t;
  REQUIRE(t.n > 0 && t.off[0] == 0 && t.off[t.n] == t.nbytes);
  std::vector<std::string> seqxm;
  materialize(t, seqxm);
  REQUIRE((int64_t)seqxm.size() == t.n);

  // the legacy route: strings in BAM order behind seqxm_xptr, rows through templid (here: a reversal)
  std::vector<std::string> shuffled(seqxm.rbegin(), seqxm.rend());
  std::vector<int32_t> templid((size_t)t.n);
  for (int64_t x = 0; x < t.n; x++) templid[(size_t)x] = (int32_t)(t.n - 1 - x);
  Soa s;
  int polls = 0;
  gather_rows(shuffled, templid.data(), t.n, s, [&]() { polls++; });
  REQUIRE(polls >= 1 && (int64_t)s.off.size() == t.n + 1 && s.off[(size_t)t.n] == t.nbytes);
  REQUIRE(memcmp(s.xm.data(), t.xm, (size_t)t.nbytes) == 0);
  for (int64_t x = 0; x <= t.n; x++) REQUIRE(s.off[(size_t)x] == t.off[x]);
  bool threw = false;
  try { std::vector<int32_t> bad(1, (int32_t)t.n + 5); Soa z; gather_rows(shuffled, bad.data(), 1, z, []() {}); } catch (const std::out_of_range &) { threw = true; }
  REQUIRE(threw);                                            // seqxm->at(): a bad templid is an error, as in the reference
  // the same row semantics on the producer's SoA (a table re-ordered by reference, a subset, a template held twice)
  {
    Soa g;
    gather_soa(t, templid.data(), t.n, g, []() {});        // reversal of the producer order == rows of `shuffled` through templid
    Soa want;
    std::vector<int32_t> ident((size_t)t.n);
    std::iota(ident.begin(), ident.end(), 0);
    gather_rows(shuffled, ident.data(), t.n, want, []() {});
    REQUIRE(g.off == want.off && memcmp(g.xm.data(), want.xm.data(), (size_t)t.nbytes) == 0);
    std::vector<int32_t> sub = {5, 5, 0, (int32_t)t.n - 1};
    Soa a, b2;
    gather_soa(t, sub.data(), (int64_t)sub.size(), a, []() {});
    gather_rows(seqxm, sub.data(), (int64_t)sub.size(), b2, []() {});
    REQUIRE(a.off == b2.off && memcmp(a.xm.data(), b2.xm.data(), (size_t)a.off.back()) == 0);
    bool oor = false;
    try { std::vector<int32_t> bad(1, -1); Soa z; gather_soa(t, bad.data(), 1, z, []() {}); } catch (const std::out_of_range &) { oor = true; }
    REQUIRE(oor);
    // the staleness check of the cached batch: identity, a permutation, a subset
    const RowOrder o_id = row_order_of(ident.data(), t.n), o_rev = row_order_of(templid.data(), t.n), o_sub = row_order_of(ident.data(), t.n - 1);
    REQUIRE(o_id.identity && !o_rev.identity && o_id != o_rev && o_id != o_sub && o_id == row_order_of(ident.data(), t.n));
    std::vector<int32_t> swapped = ident;
    std::swap(swapped[1], swapped[2]);
    REQUIRE(row_order_of(swapped.data(), t.n) != o_id && row_order_of(swapped.data(), t.n) != o_rev);
  }
  if (mode == "cpu") { printf("shim core cpu ok: %lld templates, %lld bytes\n", (long long)t.n, (long long)t.nbytes); return 0; }

#ifdef EPI_SHIM_CPU_ONLY                                     // (sanitizer build against the host-only library: no epi_batch_* to link)
  (void)dump;
  return 0;
#else
  // ---- the resident flow ----
  REQUIRE(argc >= 4);
  const std::string out = argv[3];
  Resident *r = nullptr;
  try {
    r = Resident::upload(t.xm, t.off, t.rname, t.strand, t.start, t.n);      // once per preprocessBam() object
    std::vector<int32_t> pass((size_t)t.n), pass2((size_t)t.n);
    check(epi_batch_threshold_reads(r->batch, "Z", "z", "XH", "xh", 2, 0.5, 0.1, pass.data()));
    CxTableGuard two, one, all;
    check(epi_batch_cx_report(r->batch, pass.data(), "Z", &two.t));          // no re-upload of the templates
    check(epi_batch_cytosine_report(r->batch, "Z", "z", "XH", "xh", 2, 0.5, 0.1, "Z", pass2.data(), &one.t));
    check(epi_batch_cx_report(r->batch, nullptr, "ZXH", &all.t));
    REQUIRE(pass == pass2 && one.t.nrow == two.t.nrow);
    {
      // the two-step form the Rcpp shim uses: row count first, then the table straight into caller-owned columns
      std::vector<int32_t> col[6], pass3((size_t)t.n);
      auto alloc = [&](int64_t nrow, int32_t *(&cols)[6]) { for (int c = 0; c < 6; c++) { col[c].assign((size_t)nrow + 1, -7); cols[c] = col[c].data(); } };
      const int64_t n1 = cytosine_report_into(r->batch, "Z", "z", "XH", "xh", 2, 0.5, 0.1, "Z", pass3.data(), alloc);
      const int32_t *w[6] = {one.t.rname, one.t.strand, one.t.pos, one.t.context, one.t.meth, one.t.unmeth};
      REQUIRE(n1 == one.t.nrow && pass3 == pass);
      for (int c = 0; c < 6; c++) REQUIRE(memcmp(col[c].data(), w[c], (size_t)n1 * 4) == 0 && col[c][(size_t)n1] == -7);
      const int64_t n2 = cx_report_into(r->batch, nullptr, "ZXH", alloc);
      const int32_t *w2[6] = {all.t.rname, all.t.strand, all.t.pos, all.t.context, all.t.meth, all.t.unmeth};
      REQUIRE(n2 == all.t.nrow);
      for (int c = 0; c < 6; c++) REQUIRE(memcmp(col[c].data(), w2[c], (size_t)n2 * 4) == 0);
    }
    const int32_t *a[6] = {one.t.rname, one.t.strand, one.t.pos, one.t.context, one.t.meth, one.t.unmeth};
    const int32_t *b[6] = {two.t.rname, two.t.strand, two.t.pos, two.t.context, two.t.meth, two.t.unmeth};
    for (int c = 0; c < 6; c++) REQUIRE(memcmp(a[c], b[c], (size_t)one.t.nrow * 4) == 0);
    std::vector<double> beta((size_t)t.n);
    check(epi_batch_get_xm_beta(r->batch, "Z", "z", beta.data()));
    MhlTableGuard m;
    check(epi_batch_mhl_report(r->batch, "Zz", 0, 0, 0.1, &m.t));
    {
      std::vector<int32_t> ic[5];
      std::vector<double> dc[2];
      auto alloc = [&](int64_t nrow, int32_t *(&pi)[5], double *(&pd)[2]) {
        for (int c = 0; c < 5; c++) { ic[c].assign((size_t)nrow + 1, 0); pi[c] = ic[c].data(); }
        for (int c = 0; c < 2; c++) { dc[c].assign((size_t)nrow + 1, 0.0); pd[c] = dc[c].data(); }
      };
      const int64_t n3 = mhl_report_into(r->batch, "Zz", 0, 0, 0.1, alloc);
      REQUIRE(n3 == m.t.nrow && memcmp(ic[2].data(), m.t.pos, (size_t)n3 * 4) == 0 && memcmp(ic[4].data(), m.t.coverage, (size_t)n3 * 4) == 0);
      REQUIRE(memcmp(dc[0].data(), m.t.length, (size_t)n3 * 8) == 0 && memcmp(dc[1].data(), m.t.lmhl, (size_t)n3 * 8) == 0);
    }
    const char *names[6] = {"rname", "strand", "pos", "context", "meth", "unmeth"};
    for (int c = 0; c < 6; c++) REQUIRE(dump(out + "/cx_" + names[c] + ".i32", a[c], (size_t)one.t.nrow * 4) == 0);
    const int32_t *cx[6] = {all.t.rname, all.t.strand, all.t.pos, all.t.context, all.t.meth, all.t.unmeth};
    for (int c = 0; c < 6; c++) REQUIRE(dump(out + "/cxall_" + names[c] + ".i32", cx[c], (size_t)all.t.nrow * 4) == 0);
    REQUIRE(dump(out + "/pass.i32", pass.data(), pass.size() * 4) == 0 && dump(out + "/beta.f64", beta.data(), beta.size() * 8) == 0);
    const char *mn[5] = {"rname", "strand", "pos", "context", "coverage"};
    const int32_t *mi[5] = {m.t.rname, m.t.strand, m.t.pos, m.t.context, m.t.coverage};
    for (int c = 0; c < 5; c++) REQUIRE(dump(out + "/mhl_" + mn[c] + ".i32", mi[c], (size_t)m.t.nrow * 4) == 0);
    REQUIRE(dump(out + "/mhl_length.f64", m.t.length, (size_t)m.t.nrow * 8) == 0 && dump(out + "/mhl_lmhl.f64", m.t.lmhl, (size_t)m.t.nrow * 8) == 0);
    printf("shim core gpu ok: %lld templates, %lld CG rows, %lld CX rows, %lld lMHL rows\n", (long long)t.n, (long long)one.t.nrow,
           (long long)all.t.nrow, (long long)m.t.nrow);
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    delete r;
    return 1;
  }
  delete r;
  return 0;
#endif
}
