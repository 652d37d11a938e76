// Per-read kernels: rcpp_threshold_reads (src/rcpp_threshold_reads.cpp:15-74)
// and rcpp_get_xm_beta (src/rcpp_get_xm_beta.cpp:10-43).
//
// The reference builds a 16-bin histogram of the XM nibble per read and then
// sums the bins named by each context string.  Two kernels: k_per_read_wide
// (default; further down: 8+ lanes per read, 4 reads per lane group, one LUT lookup
// per dword for class strings without repeated letters) and the general k_per_read
// below, where a group of G lanes owns one
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

struct Luts { ClassLut c[4]; };                     // weight bytes per class (ClassLut, ThrParams: common.hpp)

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
                                                   const int32_t *__restrict__ len, int64_t n, Luts L, ThrParams prm, int32_t *__restrict__ pass_out,
                                                   double *__restrict__ beta_out) {
  const int sub = threadIdx.x & (G - 1);
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const bool valid = row < n;
  int64_t rs = 0, re = 0;
  if (valid) { rs = off[row]; re = rs + len[row]; }
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

// ---- wide layout -------------------------------------------------------------------------------------------------
// The kernel above gives a read to 2 lanes, so one load instruction of a wavefront touches 32 reads x 32 bytes and
// every 128-byte line is visited by four instructions: 3.9-4.5 TB/s.  Here 8 (or 16, 32, 64) lanes own a read and a
// lane group works on RPG = 4 reads at once: an instruction covers 8 consecutive reads x 128 contiguous bytes, each
// line is touched once, and with 3 loads x 4 reads in flight per lane the same bytes are on their way per wavefront
// (scratch/ubench/row_loads.hip: 6.5 TB/s for this shape against 3.9 for the 2-lane one; a plain stream reads 6.4).
// Class counting is cut to one LUT lookup per dword: the LUT byte holds the four class memberships as 2-bit fields
// (a class string without repeated letters has weights 0/1), three dwords are added field-wise (<= 3), split into
// even / odd fields of 4 bits and accumulated over the (at most three) 16-byte chunks of a lane: <= 12 per field.
// Class strings with a repeated letter (weight 2+) and reads of 64 KiB or more take the kernel above.
constexpr int PW_NU = 3;     // 16-byte loads per lane and read
constexpr int PW_RPG = 4;    // reads per lane group

__device__ __forceinline__ uint32_t pw_lut(uint32_t w, const ClassLut &F) {
  const uint32_t lo3 = w & 0x07070707u;
  const uint32_t pick = ((w >> 1) & 0x04040404u) | 0x03020100u;
  return __builtin_amdgcn_perm(__builtin_amdgcn_perm(F.hi1, F.hi0, lo3), __builtin_amdgcn_perm(F.lo1, F.lo0, lo3), pick);
}

// sum over the G lanes of a group (G <= 16: DPP inside a row of 16 lanes; 32, 64: two more shuffles)
template <int G>
__device__ __forceinline__ uint32_t pw_group_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  if (G >= 4) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  if (G >= 8) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  if (G >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// bytes [lo,hi) of a dword set to 0xFF (32-bit arguments, any sign)
__device__ __forceinline__ uint32_t byte_range_mask32(int lo, int hi) {
  const int l = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
  const int h = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
  if (h <= l) return 0u;
  const uint32_t mh = h == 4 ? ~0u : (1u << (8 * h)) - 1u;
  const uint32_t ml = (1u << (8 * l)) - 1u;             // l < 4 here
  return mh & ~ml;
}

// waves per SIMD the wide kernel is compiled for: the loads in flight take RPG * 12 VGPRs
template <int RPG> constexpr int pw_waves() { return RPG >= 4 ? 5 : (RPG == 3 ? 6 : 8); }

template <int G, int RPG, bool BETA>
__global__ __launch_bounds__(256, (pw_waves<RPG>())) void k_per_read_wide(const uint8_t *__restrict__ xm, const int64_t *__restrict__ off,
                                                                            const int32_t *__restrict__ rlen, int64_t n, ClassLut F, ThrParams prm,
                                                                            int32_t *__restrict__ pass_out, double *__restrict__ beta_out) {
  constexpr int RW = 64 / G;                            // lane groups per wavefront
  const int sub = threadIdx.x & (G - 1);
  const int g_in = (threadIdx.x & 63) / G;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t row0 = wave * (RPG * RW) + g_in;        // reads row0 + q * RW: those of one instruction are consecutive
  const uint8_t *base[RPG];                             // the read's first 16-byte chunk
  int rel[RPG], len[RPG], nch[RPG];                     // first byte inside that chunk, length (< 65536), chunks
#pragma unroll
  for (int q = 0; q < RPG; q++) {
    const int64_t row = row0 + (int64_t)q * RW;
    const int64_t rc = row < n ? row : n - 1;           // every lane loads (the count of loads in flight stays static)
    const int64_t o0 = off[rc];
    rel[q] = (int)(o0 & 15);
    len[q] = row < n ? rlen[rc] : 0;
    nch[q] = len[q] > 0 ? (rel[q] + len[q] + 15) >> 4 : 0;
    base[q] = nch[q] > 0 ? xm + (o0 & ~(int64_t)15) : xm;
  }
  // all loads are unconditional (chunk index clamped into the read): the compiler can then wait for the oldest
  // ones with a counted vmcnt and start on read 0 while reads 1..RPG-1 are still on their way
  uint4 w[RPG][PW_NU];
#pragma unroll
  for (int q = 0; q < RPG; q++) {
    const int last = nch[q] > 0 ? nch[q] - 1 : 0;
#pragma unroll
    for (int u = 0; u < PW_NU; u++) {
      const int k = sub + u * G;
      w[q][u] = *reinterpret_cast<const uint4 *>(base[q] + ((int64_t)(k < last ? k : last) << 4));
    }
  }
  uint32_t S01[RPG], S23[RPG];
#pragma unroll
  for (int q = 0; q < RPG; q++) {
    uint32_t E = 0, O = 0;                              // 4-bit fields: classes 0, 2 (E) and 1, 3 (O) per byte lane
    uint32_t cls[4] = {0, 0, 0, 0};
    const int rs = rel[q], re = rel[q] + len[q];        // the read's bytes, counted from its first chunk
    auto flush = [&]() {
      cls[0] = __builtin_amdgcn_sad_u8(E & 0x0F0F0F0Fu, 0u, cls[0]);
      cls[2] = __builtin_amdgcn_sad_u8((E >> 4) & 0x0F0F0F0Fu, 0u, cls[2]);
      cls[1] = __builtin_amdgcn_sad_u8(O & 0x0F0F0F0Fu, 0u, cls[1]);
      cls[3] = __builtin_amdgcn_sad_u8((O >> 4) & 0x0F0F0F0Fu, 0u, cls[3]);
      E = 0; O = 0;
    };
    auto chunk = [&](const uint4 &v, int k) {
#ifdef EPI_PW_NOALU
      E ^= v.x ^ v.y ^ v.z ^ v.w;                         // timing experiment: loads only
      return;
#endif
      const int g0 = k << 4;
      uint32_t a = pw_lut(v.x, F), b = pw_lut(v.y, F), cc = pw_lut(v.z, F), d = pw_lut(v.w, F);
      if (g0 < rs || g0 + 16 > re) {                    // first / last chunk of the read: mask foreign bytes
        a &= byte_range_mask32(rs - g0, re - g0);
        b &= byte_range_mask32(rs - g0 - 4, re - g0 - 4);
        cc &= byte_range_mask32(rs - g0 - 8, re - g0 - 8);
        d &= byte_range_mask32(rs - g0 - 12, re - g0 - 12);
      }
      const uint32_t t = a + b + cc;                    // 2-bit fields, <= 3
      E += (t & 0x33333333u) + (d & 0x33333333u);       // 4-bit fields, += <= 4
      O += ((t >> 2) & 0x33333333u) + ((d >> 2) & 0x33333333u);
    };
#pragma unroll
    for (int u = 0; u < PW_NU; u++) {
      const int k = sub + u * G;
      if (k < nch[q]) chunk(w[q][u], k);
    }
    flush();
    // reads longer than G * PW_NU chunks (the host picks G from the mean length): the rest, three chunks per flush
    for (int kb = sub + PW_NU * G; kb < nch[q]; kb += PW_NU * G) {
#pragma unroll
      for (int u = 0; u < PW_NU; u++) {
        const int k = kb + u * G;
        if (k < nch[q]) chunk(*reinterpret_cast<const uint4 *>(base[q] + ((int64_t)k << 4)), k);
      }
      flush();
    }
    // a read is shorter than 64 KiB here: two counts per word through the group sum (every lane of the group gets it)
    S01[q] = pw_group_sum<G>(cls[0] | (cls[1] << 16));
    S23[q] = pw_group_sum<G>(cls[2] | (cls[3] << 16));
  }
  // one pass of comparisons and IEEE divisions for all RPG reads of the group: lane q of the group decides read q
  uint32_t s01 = S01[0], s23 = S23[0];
#pragma unroll
  for (int q = 1; q < RPG; q++) {
    if (sub == q) { s01 = S01[q]; s23 = S23[q]; }
  }
  const int64_t row = row0 + (int64_t)sub * RW;
  if (sub < RPG && row < n) {
    const unsigned n_m = s01 & 0xFFFFu, n_u = s01 >> 16;
    if (BETA) {
      unsigned n_all = n_m + n_u;                       // rcpp_get_xm_beta.cpp:37-39
      if (n_all == 0) n_all = 1;
      beta_out[row] = (double)n_m / (double)n_all;
    } else {
      int res = 0;                                      // rcpp_threshold_reads.cpp:43-70
      if (n_m != 0) {
        const unsigned n_all = n_m + n_u;
        if (!(n_all < prm.min_n_ctx)) {
          const double frac = (double)n_m / (double)n_all;
          if (!(frac < prm.min_ctx_meth_frac)) {
            res = 1;
            const unsigned o_m = s23 & 0xFFFFu;
            if (o_m > 0) {
              const unsigned o_all = o_m + (s23 >> 16);
              const double ofrac = (double)o_m / (double)o_all;
              if (ofrac > prm.max_ooctx_meth_frac) res = 0;
            }
          }
        }
      }
      pass_out[row] = res;
    }
  }
}

// 2-bit membership fields of up to four classes in one LUT; false when a class string repeats a letter
bool make_field_lut(const char *const cls[4], ClassLut *out) {
  unsigned f[16] = {0};
  for (int k = 0; k < 4; k++) {
    unsigned w[16] = {0};
    if (cls[k]) for (const unsigned char *c = reinterpret_cast<const unsigned char *>(cls[k]); *c; c++) w[ctx_to_idx(*c)]++;
    for (int i = 0; i < 16; i++) {
      if (w[i] > 1) return false;
      f[i] |= w[i] << (2 * k);
    }
  }
  auto pack = [&](int b) { return (uint32_t)(f[b] | (f[b + 1] << 8) | (f[b + 2] << 16) | (f[b + 3] << 24)); };
  out->lo0 = pack(0); out->lo1 = pack(4); out->hi0 = pack(8); out->hi1 = pack(12);
  return true;
}

static int pick_group(const epi_batch *b) {
  { const int g = options().pr_group; if (g >= 1 && g <= 64 && (g & (g - 1)) == 0) return g; }   // EPIHIP_GROUP
  // lanes per read: just enough that PR_UN chunks per lane cover a typical read in one pass (measured
  // fastest on PE150: 2 lanes x 10 chunks; one lane per read loses to uncoalesced access)
  const int64_t mean = b->n > 0 ? b->nbytes / b->n : 0;
  const int64_t chunks = mean / 16 + 2;
  int g = 2;
  while (g < 64 && (int64_t)g * PR_UN < chunks) g <<= 1;
  return g;
}

// the wide kernel when the class strings allow it (F != nullptr) and no read reaches 64 KiB
template <bool BETA>
static int launch_per_read_wide(epi_batch *b, const ClassLut &F, const ThrParams &prm, int32_t *d_pass, double *d_beta,
                                hipStream_t s) {
  const int64_t mean = b->n > 0 ? b->nbytes / b->n : 0;
  const int64_t chunks = mean / 16 + 2;
  int g = 2;                                              // short reads: fewer lanes per read (a 50-byte read is 4-5 chunks)
  while (g < 64 && (int64_t)g * PW_NU < chunks) g <<= 1;
  const int rpg_env = options().pr_rpg >= 2 && options().pr_rpg <= 4 ? options().pr_rpg : PW_RPG;   // EPIHIP_PR_RPG
  const int rpg = g == 2 ? 2 : (g == 4 ? 4 : rpg_env);    // lane q of a group decides read q: RPG <= G
  const int64_t rows_per_wg = 4 * (int64_t)rpg * (64 / g);
  const unsigned nb = (unsigned)((b->n + rows_per_wg - 1) / rows_per_wg);
  const char *pname = BETA ? "xm_beta" : "threshold";
  prof_begin(pname, s);
#define EPI_PW(GG, RR) hipLaunchKernelGGL((k_per_read_wide<GG, RR, BETA>), dim3(nb), dim3(256), 0, s, b->xm, b->off, b->len, b->n, F, prm, d_pass, d_beta)
#define EPI_PW_G(RR)                                                                   \
  switch (g) { case 8: EPI_PW(8, RR); break; case 16: EPI_PW(16, RR); break; case 32: EPI_PW(32, RR); break; default: EPI_PW(64, RR); break; }
  if (g == 2) EPI_PW(2, 2);
  else if (g == 4) EPI_PW(4, 4);
  else if (rpg == 2) { EPI_PW_G(2) } else if (rpg == 3) { EPI_PW_G(3) } else { EPI_PW_G(4) }
#undef EPI_PW_G
#undef EPI_PW
  prof_end(pname, s);
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}

template <int NCLS, bool BETA>
static int launch_per_read(epi_batch *b, const Luts &L, const ThrParams &prm, int32_t *d_pass, double *d_beta,
                           hipStream_t s, const ClassLut *F = nullptr) {
  if (b->n == 0) return EPI_OK;
  {
    if (F && options().pr_wide) {                         // EPIHIP_PR_WIDE=0: the 2-lane layout for every call
      EPI_TRY(fetch_row_stats(b, s));                     // the longest read (known since ingest)
      if (!b->h_stats.bad_len && b->h_stats.max_len < 65536 && b->nbytes >= 16) return launch_per_read_wide<BETA>(b, *F, prm, d_pass, d_beta, s);
    }
  }
  const int g = pick_group(b);
  const int64_t threads = b->n * g;
  const unsigned nb = (unsigned)((threads + 255) / 256);
  const char *pname = BETA ? "xm_beta" : "threshold";
  prof_begin(pname, s);
#define EPI_LAUNCH(GG)                                                                                         \
  case GG:                                                                                                     \
    hipLaunchKernelGGL((k_per_read<GG, NCLS, BETA>), dim3(nb), dim3(256), 0, s, b->xm, b->off, b->len, b->n, L, prm,   \
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
  ClassLut F;
  const char *const cls[4] = {ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth};
  const bool fields = make_field_lut(cls, &F);
  return launch_per_read<4, false>(b, L, prm, d_pass_out, nullptr, pick_stream(b, stream), fields ? &F : nullptr);
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
  ClassLut F;
  const char *const cls[4] = {ctx_meth, ctx_unmeth, nullptr, nullptr};
  const bool fields = make_field_lut(cls, &F);
  return launch_per_read<2, true>(b, L, prm, nullptr, d_beta_out, pick_stream(b, stream), fields ? &F : nullptr);
}

}  // extern "C"
