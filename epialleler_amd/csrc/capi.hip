// The four drop-in entry points: host pointers in, host results out.  Each is
// upload (pinned double-buffered H2D) -> resident kernels -> download; a lazily
// created default engine (device EPIHIP_DEVICE, default 0) backs them.
#include "common.hpp"
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <map>
#include <mutex>

using namespace epi;

static std::mutex g_mu;
static epi_engine *g_default_engine = nullptr;

static int default_engine(epi_engine **out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_default_engine) {
    EPI_TRY(epi_engine_create(options().device, &g_default_engine));
  }
  *out = g_default_engine;
  return EPI_OK;
}

namespace {
struct BatchGuard {           // frees the temporary batch on every exit path
  epi_batch *b = nullptr;
  ~BatchGuard() { if (b) epi_batch_free(b); }
};
}  // namespace

// per-read calls need no rname/strand/start: upload only xm + off
static int upload_xm_only(epi_engine *eng, const uint8_t *xm, const int64_t *off, int64_t n, BatchGuard &g,
                          std::vector<int32_t> &zeros) {
  zeros.assign((size_t)(n > 0 ? n : 1), 1);
  return epi_batch_upload(eng, xm, off, zeros.data(), zeros.data(), zeros.data(), n, &g.b);
}

// The columns of a library-owned table are ONE allocation (epi_*_table_free releases the first column's pointer):
// 2 MiB-aligned and advised as huge pages when large, so that the first touch by the copy threads faults 2 MiB at a time
// (six separate 28 MB mallocs cost ~25 ms of page faults and unmapping per 7 M-row table).  Large blocks that come back
// through epi_*_table_free are recycled: a caller that asks for report after report (what an R session does) gets the
// pages of the table it has just released.
namespace epi {
namespace {
struct TableBlocks {
  std::mutex mu;
  std::map<void *, size_t> live;                 // large blocks handed out: their sizes
  struct Kept { void *p; size_t bytes; } kept[2] = {{nullptr, 0}, {nullptr, 0}};
  ~TableBlocks() { for (auto &k : kept) free(k.p); }
};
TableBlocks &table_blocks() { static TableBlocks t; return t; }
constexpr size_t kHuge = (size_t)2 << 20, kKeepMin = (size_t)8 << 20, kKeepMax = (size_t)1 << 30;
}  // namespace

void *table_block(size_t bytes) {
  if (bytes >= 4 * kHuge) {
    const size_t r = (bytes + kHuge - 1) & ~(kHuge - 1);
    TableBlocks &t = table_blocks();
    {
      std::lock_guard<std::mutex> lk(t.mu);
      for (auto &k : t.kept)
        if (k.p && k.bytes >= r && k.bytes <= r + r / 2) {     // (no more than half again as large as asked for)
          void *p = k.p;
          t.live[p] = k.bytes;
          k.p = nullptr; k.bytes = 0;
          return p;
        }
    }
    void *p = aligned_alloc(kHuge, r);
    if (p) {
      (void)madvise(p, r, MADV_HUGEPAGE);
      std::lock_guard<std::mutex> lk(t.mu);
      t.live[p] = r;
      return p;
    }
  }
  return malloc(bytes ? bytes : 16);
}

void table_block_release(void *p) {
  if (!p) return;
  TableBlocks &t = table_blocks();
  void *drop = p;
  {
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.live.find(p);
    if (it != t.live.end()) {
      const size_t bytes = it->second;
      t.live.erase(it);
      if (bytes >= kKeepMin && bytes <= kKeepMax) {
        // keep it in place of an empty slot, or of the smaller kept block
        int slot = !t.kept[0].p ? 0 : !t.kept[1].p ? 1 : (t.kept[0].bytes <= t.kept[1].bytes ? 0 : 1);
        if (!t.kept[slot].p || t.kept[slot].bytes <= bytes) { drop = t.kept[slot].p; t.kept[slot].p = p; t.kept[slot].bytes = bytes; }
      }
    }
  }
  free(drop);
}
}  // namespace epi

extern "C" {

// ---- resident batch, host results: what an R / C / C++ caller without device memory of its own uses --------------

int epi_batch_threshold_reads(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                              const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                              double max_ooctx_meth_frac, int32_t *pass_out) {
  if (!b || (b->n > 0 && !pass_out)) return fail(EPI_ERR_ARG, "epi_batch_threshold_reads: bad arguments");
  if (b->n == 0) return EPI_OK;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = b->eng->stream;
  EPI_TRY(b->host_io.ensure((size_t)b->n * 8));
  int32_t *d = b->host_io.as<int32_t>();
  EPI_TRY(epi_batch_threshold_reads_dev(b, ctx_meth, ctx_unmeth, ooctx_meth ? ooctx_meth : "", ooctx_unmeth ? ooctx_unmeth : "",
                                        min_n_ctx, min_ctx_meth_frac, max_ooctx_meth_frac, d, s));
  return copy_to_host(b->eng, pass_out, d, (size_t)b->n * 4, s);
}

int epi_batch_get_xm_beta(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, double *beta_out) {
  if (!b || (b->n > 0 && !beta_out)) return fail(EPI_ERR_ARG, "epi_batch_get_xm_beta: bad arguments");
  if (b->n == 0) return EPI_OK;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = b->eng->stream;
  EPI_TRY(b->host_io.ensure((size_t)b->n * 8));
  double *d = b->host_io.as<double>();
  EPI_TRY(epi_batch_get_xm_beta_dev(b, ctx_meth, ctx_unmeth, d, s));
  return copy_to_host(b->eng, beta_out, d, (size_t)b->n * 8, s);
}

static int cx_table_to_host(epi_batch *b, int64_t nrow, hipStream_t s, epi_cx_table *out) {
  const size_t m = (size_t)(nrow > 0 ? nrow : 1);
  int32_t *base = static_cast<int32_t *>(table_block(6 * m * 4));
  if (!base) return fail(EPI_ERR_NOMEM, "out of host memory for the report table");
  int32_t *cols[6];
  for (int i = 0; i < 6; i++) cols[i] = base + (size_t)i * m;
  const int rc = epi_batch_cx_fetch_host(b, cols, s);
  if (rc) { table_block_release(base); return rc; }
  out->nrow = nrow;
  out->rname = cols[0]; out->strand = cols[1]; out->pos = cols[2];
  out->context = cols[3]; out->meth = cols[4]; out->unmeth = cols[5];
  return EPI_OK;
}

int epi_batch_cx_report_begin(epi_batch *b, const int32_t *pass, const char *ctx, int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_cx_report_begin: bad arguments");
  *nrow_out = 0;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = b->eng->stream;
  const int32_t *d_pass = nullptr;
  if (pass && b->n > 0) {
    EPI_TRY(b->host_io.ensure((size_t)b->n * 8));
    EPI_HIP(hipMemcpyAsync(b->host_io.p, pass, (size_t)b->n * 4, hipMemcpyHostToDevice, s));
    d_pass = b->host_io.as<int32_t>();
  }
  return epi_batch_cx_report_dev(b, d_pass, ctx, s, nrow_out);
}

int epi_batch_cx_report(epi_batch *b, const int32_t *pass, const char *ctx, epi_cx_table *out) {
  if (!out) return fail(EPI_ERR_ARG, "epi_batch_cx_report: out is NULL");
  memset(out, 0, sizeof(*out));
  if (!b || !ctx) return fail(EPI_ERR_ARG, "epi_batch_cx_report: bad arguments");
  int64_t nrow = 0;
  EPI_TRY(epi_batch_cx_report_begin(b, pass, ctx, &nrow));
  return cx_table_to_host(b, nrow, b->eng->stream, out);
}

int epi_batch_cytosine_report_begin(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                                    const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                                    double max_ooctx_meth_frac, const char *ctx, int32_t *pass_out, int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_cytosine_report_begin: bad arguments");
  *nrow_out = 0;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = b->eng->stream;
  int32_t *d_pass = nullptr;
  if (pass_out && b->n > 0) { EPI_TRY(b->host_io.ensure((size_t)b->n * 8)); d_pass = b->host_io.as<int32_t>(); }
  EPI_TRY(epi_batch_cytosine_report_dev(b, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n_ctx, min_ctx_meth_frac,
                                        max_ooctx_meth_frac, ctx, d_pass, s, nrow_out));
  if (d_pass) EPI_TRY(copy_to_host(b->eng, pass_out, d_pass, (size_t)b->n * 4, s));
  return EPI_OK;
}

int epi_batch_cytosine_report(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                              const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                              double max_ooctx_meth_frac, const char *ctx, int32_t *pass_out, epi_cx_table *out) {
  if (!out) return fail(EPI_ERR_ARG, "epi_batch_cytosine_report: out is NULL");
  memset(out, 0, sizeof(*out));
  if (!b || !ctx) return fail(EPI_ERR_ARG, "epi_batch_cytosine_report: bad arguments");
  int64_t nrow = 0;
  EPI_TRY(epi_batch_cytosine_report_begin(b, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n_ctx, min_ctx_meth_frac,
                                          max_ooctx_meth_frac, ctx, pass_out, &nrow));
  return cx_table_to_host(b, nrow, b->eng->stream, out);
}

int epi_batch_mhl_report_begin(epi_batch *b, const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac, int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_mhl_report_begin: bad arguments");
  EPI_HIP(hipSetDevice(b->eng->device));
  return epi_batch_mhl_report_dev(b, ctx, hmax, hmin, max_ooctx_meth_frac, b->eng->stream, nrow_out);
}

int epi_batch_mhl_report(epi_batch *b, const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac, epi_mhl_table *out) {
  if (!out) return fail(EPI_ERR_ARG, "epi_batch_mhl_report: out is NULL");
  memset(out, 0, sizeof(*out));
  if (!b || !ctx) return fail(EPI_ERR_ARG, "epi_batch_mhl_report: bad arguments");
  hipStream_t s = b->eng->stream;
  int64_t nrow = 0;
  EPI_TRY(epi_batch_mhl_report_begin(b, ctx, hmax, hmin, max_ooctx_meth_frac, &nrow));
  const size_t m = (size_t)(nrow > 0 ? nrow : 1);
  double *base = static_cast<double *>(table_block(2 * m * 8 + 5 * m * 4));   // one allocation: the doubles first (alignment)
  if (!base) return fail(EPI_ERR_NOMEM, "out of host memory for the report table");
  double *dc[2] = {base, base + m};
  int32_t *ic[5];
  for (int i = 0; i < 5; i++) ic[i] = reinterpret_cast<int32_t *>(base + 2 * m) + (size_t)i * m;
  const int rc = epi_batch_mhl_fetch_host(b, ic, dc, s);
  if (rc) { table_block_release(base); return rc; }
  out->nrow = nrow;
  out->rname = ic[0]; out->strand = ic[1]; out->pos = ic[2]; out->context = ic[3]; out->coverage = ic[4];
  out->length = dc[0]; out->lmhl = dc[1];
  return EPI_OK;
}

int epi_default_engine(epi_engine **out) {
  if (!out) return fail(EPI_ERR_ARG, "epi_default_engine: out is NULL");
  return default_engine(out);
}

// ---- the four drop-in entry points: upload, the resident call, free ----------------------------------------------

int epi_threshold_reads(const uint8_t *xm, const int64_t *off, int64_t n, const char *ctx_meth,
                        const char *ctx_unmeth, const char *ooctx_meth, const char *ooctx_unmeth,
                        uint32_t min_n_ctx, double min_ctx_meth_frac, double max_ooctx_meth_frac,
                        int32_t *pass_out) {
  if (n < 0 || !off || (n > 0 && !pass_out)) return fail(EPI_ERR_ARG, "epi_threshold_reads: bad arguments");
  epi_engine *eng;
  EPI_TRY(default_engine(&eng));
  if (n == 0) return EPI_OK;
  BatchGuard g;
  std::vector<int32_t> z;
  EPI_TRY(upload_xm_only(eng, xm, off, n, g, z));
  return epi_batch_threshold_reads(g.b, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n_ctx, min_ctx_meth_frac,
                                   max_ooctx_meth_frac, pass_out);
}

int epi_get_xm_beta(const uint8_t *xm, const int64_t *off, int64_t n, const char *ctx_meth, const char *ctx_unmeth,
                    double *beta_out) {
  if (n < 0 || !off || (n > 0 && !beta_out)) return fail(EPI_ERR_ARG, "epi_get_xm_beta: bad arguments");
  epi_engine *eng;
  EPI_TRY(default_engine(&eng));
  if (n == 0) return EPI_OK;
  BatchGuard g;
  std::vector<int32_t> z;
  EPI_TRY(upload_xm_only(eng, xm, off, n, g, z));
  return epi_batch_get_xm_beta(g.b, ctx_meth, ctx_unmeth, beta_out);
}

int epi_cx_report(const uint8_t *xm, const int64_t *off, const int32_t *rname, const int32_t *strand,
                  const int32_t *start, const int32_t *pass, int64_t n, const char *ctx, epi_cx_table *out) {
  if (!out) return fail(EPI_ERR_ARG, "epi_cx_report: out is NULL");
  memset(out, 0, sizeof(*out));
  if (n < 0 || !off || !ctx) return fail(EPI_ERR_ARG, "epi_cx_report: bad arguments");
  epi_engine *eng;
  EPI_TRY(default_engine(&eng));
  BatchGuard g;
  EPI_TRY(epi_batch_upload(eng, xm, off, rname, strand, start, n, &g.b));
  return epi_batch_cx_report(g.b, pass, ctx, out);
}

int epi_mhl_report(const uint8_t *xm, const int64_t *off, const int32_t *rname, const int32_t *strand,
                   const int32_t *start, int64_t n, const char *ctx, int hmax, int hmin,
                   double max_ooctx_meth_frac, epi_mhl_table *out) {
  if (!out) return fail(EPI_ERR_ARG, "epi_mhl_report: out is NULL");
  memset(out, 0, sizeof(*out));
  if (n < 0 || !off || !ctx) return fail(EPI_ERR_ARG, "epi_mhl_report: bad arguments");
  epi_engine *eng;
  EPI_TRY(default_engine(&eng));
  BatchGuard g;
  EPI_TRY(epi_batch_upload(eng, xm, off, rname, strand, start, n, &g.b));
  return epi_batch_mhl_report(g.b, ctx, hmax, hmin, max_ooctx_meth_frac, out);
}

}  // extern "C"
