// Per-read kernels: rcpp_threshold_reads (src/rcpp_threshold_reads.cpp:15-74)
// and rcpp_get_xm_beta (src/rcpp_get_xm_beta.cpp:10-43).
//
// The reference builds a 16-bin histogram of the XM nibble per read and then
// sums the bins named by each context string.  Here a group of G lanes owns one
// read; every lane streams aligned 16-byte chunks of it (global_load_dwordx4),
// maps four codes at a time to their per-class weights with two v_perm_b32
// byte-LUT lookups (codes 0-7 / 8-15) and sums the four weight bytes with
// v_sad_u8, so only the class totals are ever materialised.  A weight is the
// number of times a code's letter occurs in the class string (the reference's
// for_each over the string counts duplicates twice).  HBM-bound: L+8 bytes in,
// 4 (pass) or 8 (beta) bytes out per read.
#include "common.hpp"
#include <stdlib.h>
#include <string.h>

namespace epi {

#ifndef EPI_PR_UN
#define EPI_PR_UN 10
#endif
constexpr int PR_UN = EPI_PR_UN;   // 16-byte loads a lane keeps in flight

struct ClassLut { uint32_t lo0, lo1, hi0, hi1; };   // weight bytes for codes 0-3, 4-7, 8-11, 12-15
struct Luts { ClassLut c[4]; };

struct ThrParams {
  uint32_t min_n_ctx;
  double min_ctx_meth_frac, max_ooctx_meth_frac;
};

static int make_lut(const char *s, ClassLut *out) {
  unsigned w[16] = {0};
  if (s) for (const unsigned char *c = reinterpret_cast<const unsigned char *>(s); *c; c++) w[ctx_to_idx(*c)]++;
  for (int i = 0; i < 16; i++)
    if (w[i] > 255) return fail(EPI_ERR_ARG, "context string repeats a letter more than 255 times");
  auto pack = [&](int b) { return (uint32_t)(w[b] | (w[b + 1] << 8) | (w[b + 2] << 16) | (w[b + 3] << 24)); };
  out->lo0 = pack(0); out->lo1 = pack(4); out->hi0 = pack(8); out->hi1 = pack(12);
  return EPI_OK;
}

template <int NCLS>
__device__ __forceinline__ void acc_dword(uint32_t w, uint32_t mask, const Luts &L, uint32_t (&acc)[NCLS]) {
#ifdef EPI_PR_NOALU
  acc[0] |= w & mask;                                   // timing experiment: loads only
  return;
#endif
  const uint32_t v = w & 0x0F0F0F0Fu;                 // unpack_ctx_idx, four codes
  const uint32_t lo3 = v & 0x07070707u;
  // per byte: take the codes-8..15 lookup when bit 3 of the code is set (selector j + 4*bit3)
  const uint32_t pick = 0x03020100u | ((v >> 1) & 0x04040404u);
#pragma unroll
  for (int k = 0; k < NCLS; k++) {
    const uint32_t rlo = __builtin_amdgcn_perm(L.c[k].lo1, L.c[k].lo0, lo3);
    const uint32_t rhi = __builtin_amdgcn_perm(L.c[k].hi1, L.c[k].hi0, lo3);
    const uint32_t sel = __builtin_amdgcn_perm(rhi, rlo, pick) & mask;
    acc[k] = __builtin_amdgcn_sad_u8(sel, 0u, acc[k]);  // += sum of the four weight bytes
  }
}

// bytes [lo,hi) of a dword set to 0xFF (lo,hi clamped to [0,4])
__device__ __forceinline__ uint32_t byte_range_mask(int64_t lo, int64_t hi) {
  const int l = (int)(lo < 0 ? 0 : (lo > 4 ? 4 : lo));
  const int h = (int)(hi < 0 ? 0 : (hi > 4 ? 4 : hi));
  if (h <= l) return 0u;
  const uint64_t mh = (1ull << (8 * h)) - 1ull;
  const uint64_t ml = (1ull << (8 * l)) - 1ull;
  return (uint32_t)(mh & ~ml);
}

template <int G, int NCLS, bool BETA>
__global__ __launch_bounds__(256) void k_per_read(const uint8_t *__restrict__ xm, const int64_t *__restrict__ off,
                                                   int64_t n, Luts L, ThrParams prm, int32_t *__restrict__ pass_out,
                                                   double *__restrict__ beta_out) {
  const int sub = threadIdx.x & (G - 1);
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const bool valid = row < n;
  int64_t rs = 0, re = 0;
  if (valid) { rs = off[row]; re = off[row + 1]; }
  const int64_t c0 = rs >> 4;
  const int64_t c1 = re > rs ? (re + 15) >> 4 : c0;
  uint32_t acc[NCLS];
#pragma unroll
  for (int k = 0; k < NCLS; k++) acc[k] = 0;

  // PR_UN independent 16-byte loads per lane are issued before any of them is consumed
  for (int64_t cb = c0 + sub; cb < c1; cb += (int64_t)G * PR_UN) {
    uint4 w[PR_UN];
#pragma unroll
    for (int u = 0; u < PR_UN; u++) {
      const int64_t c = cb + (int64_t)u * G;
      w[u] = c < c1 ? *reinterpret_cast<const uint4 *>(xm + (c << 4)) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < PR_UN; u++) {
      const int64_t c = cb + (int64_t)u * G;
      if (c < c1) {
        const int64_t g0 = c << 4;
        uint32_t m0 = ~0u, m1 = ~0u, m2 = ~0u, m3 = ~0u;
        if (g0 < rs || g0 + 16 > re) {                // first / last chunk of the read: mask foreign bytes
          m0 = byte_range_mask(rs - g0, re - g0);
          m1 = byte_range_mask(rs - g0 - 4, re - g0 - 4);
          m2 = byte_range_mask(rs - g0 - 8, re - g0 - 8);
          m3 = byte_range_mask(rs - g0 - 12, re - g0 - 12);
        }
        acc_dword<NCLS>(w[u].x, m0, L, acc);
        acc_dword<NCLS>(w[u].y, m1, L, acc);
        acc_dword<NCLS>(w[u].z, m2, L, acc);
        acc_dword<NCLS>(w[u].w, m3, L, acc);
      }
    }
  }
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) {
#pragma unroll
    for (int k = 0; k < NCLS; k++) acc[k] += __shfl_xor(acc[k], d, 64);
  }
  if (!valid || sub != 0) return;
  if (BETA) {
    unsigned n_all = acc[0] + acc[1];                 // rcpp_get_xm_beta.cpp:37-39
    if (n_all == 0) n_all = 1;
    beta_out[row] = (double)acc[0] / (double)n_all;
  } else {
    int res = 0;                                      // rcpp_threshold_reads.cpp:43-70
    const unsigned n_m = acc[0];
    if (n_m != 0) {
      const unsigned n_all = n_m + acc[1];
      if (!(n_all < prm.min_n_ctx)) {
        const double frac = (double)n_m / (double)n_all;
        if (!(frac < prm.min_ctx_meth_frac)) {
          res = 1;
          if (NCLS > 2) {
            const unsigned o_m = acc[NCLS > 2 ? 2 : 0];
            if (o_m > 0) {
              const unsigned o_all = o_m + acc[NCLS > 2 ? 3 : 0];
              const double ofrac = (double)o_m / (double)o_all;
              if (ofrac > prm.max_ooctx_meth_frac) res = 0;
            }
          }
        }
      }
    }
    pass_out[row] = res;
  }
}

static int pick_group(const epi_batch *b) {
  const char *env = getenv("EPIHIP_GROUP");
  if (env) { int g = atoi(env); if (g >= 1 && g <= 64 && (g & (g - 1)) == 0) return g; }
  // lanes per read: just enough that PR_UN chunks per lane cover a typical read in one pass (measured
  // fastest on PE150: 2 lanes x 10 chunks; one lane per read loses to uncoalesced access)
  const int64_t mean = b->n > 0 ? b->nbytes / b->n : 0;
  const int64_t chunks = mean / 16 + 2;
  int g = 2;
  while (g < 64 && (int64_t)g * PR_UN < chunks) g <<= 1;
  return g;
}

template <int NCLS, bool BETA>
static int launch_per_read(epi_batch *b, const Luts &L, const ThrParams &prm, int32_t *d_pass, double *d_beta,
                           hipStream_t s) {
  if (b->n == 0) return EPI_OK;
  const int g = pick_group(b);
  const int64_t threads = b->n * g;
  const unsigned nb = (unsigned)((threads + 255) / 256);
  const char *pname = BETA ? "xm_beta" : "threshold";
  prof_begin(pname, s);
#define EPI_LAUNCH(GG)                                                                                         \
  case GG:                                                                                                     \
    hipLaunchKernelGGL((k_per_read<GG, NCLS, BETA>), dim3(nb), dim3(256), 0, s, b->xm, b->off, b->n, L, prm,   \
                       d_pass, d_beta);                                                                        \
    break;
  switch (g) {
    EPI_LAUNCH(1) EPI_LAUNCH(2) EPI_LAUNCH(4) EPI_LAUNCH(8) EPI_LAUNCH(16) EPI_LAUNCH(32) EPI_LAUNCH(64)
    default: return fail(EPI_ERR_ARG, "bad group size");
  }
#undef EPI_LAUNCH
  prof_end(pname, s);
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}

}  // namespace epi

using namespace epi;

extern "C" {

int epi_batch_threshold_reads_dev(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth,
                                  const char *ooctx_meth, const char *ooctx_unmeth, uint32_t min_n_ctx,
                                  double min_ctx_meth_frac, double max_ooctx_meth_frac, int32_t *d_pass_out,
                                  void *stream) {
  if (!b) return fail(EPI_ERR_ARG, "NULL batch");
  if (!ctx_meth || !ctx_unmeth) return fail(EPI_ERR_ARG, "ctx_meth/ctx_unmeth must be non-NULL strings");
  if (b->n > 0 && !d_pass_out) return fail(EPI_ERR_ARG, "NULL output");
  EPI_HIP(hipSetDevice(b->eng->device));
  Luts L;
  EPI_TRY(make_lut(ctx_meth, &L.c[0]));
  EPI_TRY(make_lut(ctx_unmeth, &L.c[1]));
  EPI_TRY(make_lut(ooctx_meth, &L.c[2]));
  EPI_TRY(make_lut(ooctx_unmeth, &L.c[3]));
  ThrParams prm{min_n_ctx, min_ctx_meth_frac, max_ooctx_meth_frac};
  return launch_per_read<4, false>(b, L, prm, d_pass_out, nullptr, pick_stream(b, stream));
}

int epi_batch_get_xm_beta_dev(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, double *d_beta_out,
                              void *stream) {
  if (!b) return fail(EPI_ERR_ARG, "NULL batch");
  if (!ctx_meth || !ctx_unmeth) return fail(EPI_ERR_ARG, "ctx_meth/ctx_unmeth must be non-NULL strings");
  if (b->n > 0 && !d_beta_out) return fail(EPI_ERR_ARG, "NULL output");
  EPI_HIP(hipSetDevice(b->eng->device));
  Luts L;
  memset(&L, 0, sizeof(L));
  EPI_TRY(make_lut(ctx_meth, &L.c[0]));
  EPI_TRY(make_lut(ctx_unmeth, &L.c[1]));
  ThrParams prm{0, 0.0, 0.0};
  return launch_per_read<2, true>(b, L, prm, nullptr, d_beta_out, pick_stream(b, stream));
}

}  // extern "C"
