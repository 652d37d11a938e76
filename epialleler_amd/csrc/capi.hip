// The four drop-in entry points: host pointers in, host results out.  Each is
// upload (pinned double-buffered H2D) -> resident kernels -> download; a lazily
// created default engine (device EPIHIP_DEVICE, default 0) backs them.
#include "common.hpp"
#include <stdlib.h>
#include <string.h>
#include <mutex>

using namespace epi;

static std::mutex g_mu;
static epi_engine *g_default_engine = nullptr;

static int default_engine(epi_engine **out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_default_engine) {
    int dev = 0;
    if (const char *e = getenv("EPIHIP_DEVICE")) dev = atoi(e);
    EPI_TRY(epi_engine_create(dev, &g_default_engine));
  }
  *out = g_default_engine;
  return EPI_OK;
}

namespace {
struct BatchGuard {           // frees the temporary batch on every exit path
  epi_batch *b = nullptr;
  ~BatchGuard() { if (b) epi_batch_free(b); }
};
struct DevTmp {
  void *p = nullptr;
  ~DevTmp() { if (p) (void)hipFree(p); }
};
}  // namespace

// per-read calls need no rname/strand/start: upload only xm + off
static int upload_xm_only(epi_engine *eng, const uint8_t *xm, const int64_t *off, int64_t n, BatchGuard &g,
                          std::vector<int32_t> &zeros) {
  zeros.assign((size_t)(n > 0 ? n : 1), 1);
  return epi_batch_upload(eng, xm, off, zeros.data(), zeros.data(), zeros.data(), n, &g.b);
}

extern "C" {

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
  DevTmp d;
  EPI_HIP(hipMalloc(&d.p, (size_t)n * 4));
  EPI_TRY(epi_batch_threshold_reads_dev(g.b, ctx_meth, ctx_unmeth, ooctx_meth ? ooctx_meth : "",
                                        ooctx_unmeth ? ooctx_unmeth : "", min_n_ctx, min_ctx_meth_frac,
                                        max_ooctx_meth_frac, static_cast<int32_t *>(d.p), eng->stream));
  EPI_HIP(hipMemcpyAsync(pass_out, d.p, (size_t)n * 4, hipMemcpyDeviceToHost, eng->stream));
  EPI_HIP(hipStreamSynchronize(eng->stream));
  return EPI_OK;
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
  DevTmp d;
  EPI_HIP(hipMalloc(&d.p, (size_t)n * 8));
  EPI_TRY(epi_batch_get_xm_beta_dev(g.b, ctx_meth, ctx_unmeth, static_cast<double *>(d.p), eng->stream));
  EPI_HIP(hipMemcpyAsync(beta_out, d.p, (size_t)n * 8, hipMemcpyDeviceToHost, eng->stream));
  EPI_HIP(hipStreamSynchronize(eng->stream));
  return EPI_OK;
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
  DevTmp d;
  if (pass && n > 0) {
    EPI_HIP(hipMalloc(&d.p, (size_t)n * 4));
    EPI_HIP(hipMemcpyAsync(d.p, pass, (size_t)n * 4, hipMemcpyHostToDevice, eng->stream));
  }
  int64_t nrow = 0;
  EPI_TRY(epi_batch_cx_report_dev(g.b, static_cast<const int32_t *>(d.p), ctx, eng->stream, &nrow));
  int32_t *cols[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < 6; i++) {
    cols[i] = static_cast<int32_t *>(malloc((size_t)(nrow > 0 ? nrow : 1) * 4));
    if (!cols[i]) { for (int k = 0; k < i; k++) free(cols[k]); return fail(EPI_ERR_NOMEM, "epi_cx_report: out of host memory"); }
  }
  int rc = epi_batch_cx_fetch_host(g.b, cols, eng->stream);
  if (rc) { for (int i = 0; i < 6; i++) free(cols[i]); return rc; }
  out->nrow = nrow;
  out->rname = cols[0]; out->strand = cols[1]; out->pos = cols[2];
  out->context = cols[3]; out->meth = cols[4]; out->unmeth = cols[5];
  return EPI_OK;
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
  int64_t nrow = 0;
  EPI_TRY(epi_batch_mhl_report_dev(g.b, ctx, hmax, hmin, max_ooctx_meth_frac, eng->stream, &nrow));
  const size_t m = (size_t)(nrow > 0 ? nrow : 1);
  int32_t *ic[5];
  double *dc[2];
  for (int i = 0; i < 5; i++) ic[i] = static_cast<int32_t *>(malloc(m * 4));
  for (int i = 0; i < 2; i++) dc[i] = static_cast<double *>(malloc(m * 8));
  bool ok = true;
  for (int i = 0; i < 5; i++) ok = ok && ic[i];
  for (int i = 0; i < 2; i++) ok = ok && dc[i];
  int rc = ok ? epi_batch_mhl_fetch_host(g.b, ic, dc, eng->stream) : fail(EPI_ERR_NOMEM, "epi_mhl_report: out of host memory");
  if (rc) { for (int i = 0; i < 5; i++) free(ic[i]); for (int i = 0; i < 2; i++) free(dc[i]); return rc; }
  out->nrow = nrow;
  out->rname = ic[0]; out->strand = ic[1]; out->pos = ic[2]; out->context = ic[3]; out->coverage = ic[4];
  out->length = dc[0]; out->lmhl = dc[1];
  return EPI_OK;
}

}  // extern "C"
