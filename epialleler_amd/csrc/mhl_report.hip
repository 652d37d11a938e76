// rcpp_mhl_report (src/rcpp_mhl_report.cpp:46-228) on the GPU: linearised
// Methylated Haplotype Load per cytosine.
//
// Pass 1 of the reference (:158-182) is a sequential run-length walk over each
// read: in-context bases are the haplotype; a maximal run of methylated
// (upper-case) in-context bases, not interrupted by an unmethylated in-context
// base, with m members gets num[i] = S(m) for EVERY byte between its first and
// last member.  k_mhl_rows restates that as two segmented scans (segments are
// cut by lower-case in-context bytes): A(i) = members at or before i, B(i) =
// members at or after i, so byte i lies in a span iff A>0 and B>0 and
// m = A + B - member(i).  A group of G lanes owns a read and walks it in blocks
// of 16*G bytes (16 bytes per lane, sequential in registers; G-lane shuffles
// scans across lanes; a carried state across blocks).  It stores m per byte
// (u16) and per read the haplotype size h, or -1 when the read is skipped
// (h < hmin or out-of-context beta too high, :176-179).
//
// Pass 2 (:185-195) is the CX histogram plus three 64-bit sums per
// (pos,strand): k_mhl_tiles is the CX tile kernel with ds_add_u64 for
// sum(h), sum(S(m_i)), sum(S(h)) and 512-position tiles (56 KiB of LDS).
// S(n) = n(n+1)(n+2)/6 is computed arithmetically (no 64K-entry table).
#include "common.hpp"
#include <stdlib.h>
#include <string.h>

namespace epi {

constexpr int MHL_WG = 512;
constexpr int MHL_PPT = kMhlTile / MHL_WG;      // 1 position per thread in the emit phase
static_assert(MHL_PPT == 1, "emit phase assumes one position per thread");

__host__ __device__ __forceinline__ uint64_t nrS(uint64_t n) { return n < 2 ? n : (n * (n + 1) * (n + 2)) / 6; }   // :39-43
// mhl_lookup[n] (:110-116) without the table; indices clamp at 65535 (the reference's table ends there)
__device__ __forceinline__ uint64_t mhl_lut(uint32_t n, uint32_t H) {
  if (n > 65535u) n = 65535u;
  return n < H ? nrS(n) : nrS(H);
}

struct Seg { uint32_t has; uint32_t cnt; };     // scan element: saw a cut? members since the last cut
__device__ __forceinline__ Seg seg_combine(Seg left, Seg right) {   // state after `left` then `right`
  Seg r;
  r.has = left.has | right.has;
  r.cnt = right.has ? right.cnt : left.cnt + right.cnt;
  return r;
}

template <int G>
__global__ __launch_bounds__(256) void k_mhl_rows(const uint8_t *__restrict__ xm, const int64_t *__restrict__ off,
                                                   int64_t n, uint32_t ctx_mask, int32_t hmin, double max_oo,
                                                   uint16_t *__restrict__ m_out, int32_t *__restrict__ rowinfo) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (G - 1);
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const bool valid = row < n;
  int64_t rs = 0, re = 0;
  if (valid) { rs = off[row]; re = off[row + 1]; }
  const int64_t c0 = rs >> 4;
  const int64_t c1 = re > rs ? (re + 15) >> 4 : c0;
  const int64_t nblk = (c1 - c0 + G - 1) / G;

  uint32_t h = 0, oo_m = 0, oo_u = 0;
  // ---- forward: A(i) -> m_out (temporarily) ----
  Seg carry = {0u, 0u};
  for (int64_t blk = 0; blk < nblk; blk++) {
    const int64_t c = c0 + blk * G + sub;
    const int64_t g0 = c << 4;
    uint4 w = make_uint4(0, 0, 0, 0);
    const bool live = c < c1;
    if (live) w = *reinterpret_cast<const uint4 *>(xm + g0);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    uint32_t a0[16];
    uint32_t cutseen = 0;      // bit i: a cut at index <= i inside this chunk
    Seg me = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int64_t g = g0 + i;
      const bool inrow = live && g >= rs && g < re;
      const uint32_t code = (ww[i >> 2] >> (8 * (i & 3))) & 15u;
      const bool in = inrow && ((ctx_mask >> code) & 1u);
      const bool U = in && code < 8u, Lw = in && code >= 8u;
      if (inrow) {
        h += in;
        if (!in) {
          oo_m += (code == 2u) | (code == 5u) | (code == 6u) | (code == 7u);
          oo_u += (code == 10u) | (code == 13u) | (code == 14u) | (code == 15u);
        }
      }
      if (Lw) { me.has = 1u; me.cnt = 0u; } else me.cnt += U;
      if (me.has) cutseen |= 1u << i;
      a0[i] = Lw ? 0u : me.cnt;
    }
    // inclusive scan of chunk summaries over the group, then the state entering this chunk
    Seg inc = me;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
      Seg l;
      l.has = __shfl_up(inc.has, d, G);
      l.cnt = __shfl_up(inc.cnt, d, G);
      if (sub >= d) inc = seg_combine(l, inc);
    }
    Seg ex;
    ex.has = __shfl_up(inc.has, 1, G);
    ex.cnt = __shfl_up(inc.cnt, 1, G);
    if (sub == 0) { ex.has = 0u; ex.cnt = 0u; }
    const Seg entering = seg_combine(carry, ex);
    Seg last;
    last.has = __shfl(inc.has, G - 1, G);
    last.cnt = __shfl(inc.cnt, G - 1, G);
    carry = seg_combine(carry, last);
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int64_t g = g0 + i;
      if (live && g >= rs && g < re) {
        uint32_t a = a0[i];
        const uint32_t code = (ww[i >> 2] >> (8 * (i & 3))) & 15u;
        const bool Lw = ((ctx_mask >> code) & 1u) && code >= 8u;
        if (!Lw && !((cutseen >> i) & 1u)) a += entering.cnt;
        m_out[g] = (uint16_t)(a > 65535u ? 65535u : a);
      }
    }
  }
  // row totals
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) {
    h += __shfl_xor(h, d, 64);
    oo_m += __shfl_xor(oo_m, d, 64);
    oo_u += __shfl_xor(oo_u, d, 64);
  }
  bool keep = true;
  {
    const double frac = (double)oo_m / (double)((uint64_t)oo_m + oo_u);      // :178 (0/0 = NaN -> kept)
    if ((int)h < hmin || frac > max_oo) keep = false;                        // :179
  }
  if (valid && sub == 0) rowinfo[row] = keep ? (int32_t)h : -1;

  // ---- backward: B(i), then m = A + B - member ----
  carry.has = 0u; carry.cnt = 0u;
  for (int64_t blk = nblk - 1; blk >= 0; blk--) {
    const int64_t c = c0 + blk * G + sub;
    const int64_t g0 = c << 4;
    uint4 w = make_uint4(0, 0, 0, 0);
    const bool live = c < c1;
    if (live) w = *reinterpret_cast<const uint4 *>(xm + g0);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    uint32_t b0[16];
    uint32_t cutseen = 0;      // bit i: a cut at index >= i inside this chunk
    Seg me = {0u, 0u};
#pragma unroll
    for (int i = 15; i >= 0; i--) {
      const int64_t g = g0 + i;
      const bool inrow = live && g >= rs && g < re;
      const uint32_t code = (ww[i >> 2] >> (8 * (i & 3))) & 15u;
      const bool in = inrow && ((ctx_mask >> code) & 1u);
      const bool U = in && code < 8u, Lw = in && code >= 8u;
      if (Lw) { me.has = 1u; me.cnt = 0u; } else me.cnt += U;
      if (me.has) cutseen |= 1u << i;
      b0[i] = Lw ? 0u : me.cnt;
    }
    // suffix scan: state entering this chunk from the right
    Seg inc = me;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
      Seg r;
      r.has = __shfl_down(inc.has, d, G);
      r.cnt = __shfl_down(inc.cnt, d, G);
      if (sub + d < G) inc = seg_combine(r, inc);      // walking leftwards: `r` was seen first
    }
    Seg ex;
    ex.has = __shfl_down(inc.has, 1, G);
    ex.cnt = __shfl_down(inc.cnt, 1, G);
    if (sub == G - 1) { ex.has = 0u; ex.cnt = 0u; }
    const Seg entering = seg_combine(carry, ex);
    Seg first;
    first.has = __shfl(inc.has, 0, G);
    first.cnt = __shfl(inc.cnt, 0, G);
    carry = seg_combine(carry, first);
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int64_t g = g0 + i;
      if (live && g >= rs && g < re) {
        const uint32_t code = (ww[i >> 2] >> (8 * (i & 3))) & 15u;
        const bool in = (ctx_mask >> code) & 1u;
        const bool U = in && code < 8u, Lw = in && code >= 8u;
        uint32_t bv = b0[i];
        if (!Lw && !((cutseen >> i) & 1u)) bv += entering.cnt;
        const uint32_t av = m_out[g];
        uint32_t m = 0;
        if (!Lw && av > 0u && bv > 0u) { m = av + bv - (U ? 1u : 0u); if (m > 65535u) m = 65535u; }
        m_out[g] = keep ? (uint16_t)m : (uint16_t)0;
      }
    }
  }
}

struct MhlArgs {
  const uint8_t *xm;
  const uint16_t *m;
  const int64_t *off;
  const int32_t *start, *strand, *rowinfo;
  const Tile *tiles;
  uint32_t ctx_mask, H;
  uint32_t *pool_key, *pool_cov;
  double *pool_len, *pool_lmhl;
  uint32_t pool_cap;
  uint32_t *cursor, *tile_nrow, *tile_base;
};

constexpr uint64_t kMhlSlotMap = 0x7510831164111211ull;   // same slots as the CX kernel

__global__ __launch_bounds__(MHL_WG) void k_mhl_tiles(MhlArgs a) {
  constexpr int T = kMhlTile;
  __shared__ unsigned long long sum64[2 * 3 * T];   // [strand][hsum, num, den][pos]
  __shared__ uint32_t cnt[16 * T];                  // [strand][8][pos]
  __shared__ uint32_t s_scan[MHL_WG / 64 + 2];
  const int tile = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = MHL_WG / 64;
  for (int i = threadIdx.x; i < 16 * T; i += MHL_WG) cnt[i] = 0;
  for (int i = threadIdx.x; i < 6 * T; i += MHL_WG) sum64[i] = 0ull;
  __syncthreads();
  const Tile td = a.tiles[tile];
  const int rot = (lane >> 3) & 3;

  for (int rbase = td.row_lo + wave * 64; rbase < td.row_hi; rbase += NW * 64) {
    const int r = rbase + lane;
    int i_lo = 0, i_hi = 0, pbase = 0, sflag = 0, hrow = -1;
    int64_t o = 0;
    if (r < td.row_hi) {
      hrow = a.rowinfo[r];
      if (hrow >= 0) {
        const int64_t st = a.start[r];
        o = a.off[r];
        const int64_t len = a.off[r + 1] - o;
        const int64_t rel = td.pos0 - st;
        const int64_t lo = rel > 0 ? rel : 0;
        const int64_t hi = len < rel + T ? len : rel + T;
        if (hi > lo) { i_lo = (int)lo; i_hi = (int)hi; }
        pbase = (int)(-rel);
        sflag = a.strand[r] - 1;
      }
    }
    const int nrows = td.row_hi - rbase < 64 ? td.row_hi - rbase : 64;
    for (int j = 0; j < nrows; j++) {
      const int jl = __builtin_amdgcn_readlane(i_lo, j), jh = __builtin_amdgcn_readlane(i_hi, j);
      if (jl >= jh) continue;
      const int lo32 = __builtin_amdgcn_readlane((int)(uint32_t)o, j);
      const int hi32 = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)o >> 32), j);
      const int64_t jo = (int64_t)(((uint64_t)(uint32_t)hi32 << 32) | (uint32_t)lo32);
      const int jp = __builtin_amdgcn_readlane(pbase, j);
      const int js = __builtin_amdgcn_readlane(sflag, j);
      const uint32_t jh_size = (uint32_t)__builtin_amdgcn_readlane(hrow, j);
      const uint64_t den_inc = mhl_lut(jh_size, a.H);                       // :194
      uint32_t *cb = cnt + js * (8 * T);
      unsigned long long *sb = sum64 + js * (3 * T);
      const int64_t b0 = jo + jl, b1 = jo + jh;
      for (int64_t ad = (b0 & ~3LL) + 4 * lane; ad < b1; ad += 256) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(a.xm + ad);
        const uint2 mm = *reinterpret_cast<const uint2 *>(a.m + ad);         // four u16 stretch sizes
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int bb = (q + rot) & 3;
          const int64_t bad = ad + bb;
          if (bad >= b0 && bad < b1) {
            const uint32_t code = (w >> (8 * bb)) & 15u;
            if (code != 11u) {                                               // :187
              const uint32_t slot = (uint32_t)(kMhlSlotMap >> (4 * code)) & 15u;
              const int p = jp + (int)(bad - jo);
              atomicAdd(&cb[slot * T + p], code == 9u ? 2u : 1u);            // :190-191
              const uint32_t mi = ((bb & 2) ? mm.y : mm.x) >> (16 * (bb & 1)) & 0xFFFFu;
              // the reference's counter slots 8/3/4 double as the sums (:190 vs :192-194)
              atomicAdd(&sb[0 * T + p], (unsigned long long)jh_size + (code == 8u ? 1ull : 0ull));   // :192
              const unsigned long long ni = (mi ? mhl_lut(mi, a.H) : 0ull) + (code == 3u ? 1ull : 0ull);
              if (ni) atomicAdd(&sb[1 * T + p], ni);                         // :193
              atomicAdd(&sb[2 * T + p], den_inc + (code == 4u ? 1ull : 0ull));   // :194
            }
          }
        }
      }
    }
  }
  __syncthreads();

  // emit: one position per thread, '+' then '-'
  const int p = threadIdx.x;
  uint32_t key[2], cov[2];
  double len[2], lm[2];
  bool ok[2];
  int nr = 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    uint32_t c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = cnt[(s * 8 + k) * T + p];
    const uint32_t nH = c[SLOT_H] + c[SLOT_h], nX = c[SLOT_X] + c[SLOT_x], nZ = c[SLOT_Z] + c[SLOT_z];
    const uint32_t cv = c[SLOT_DOT] + c[SLOT_OTHER] + nH + nX + nZ;
    const uint32_t half = cv >> 1;                                           // :77
    int k = 0;
    uint32_t cc = 0;
    if (cv == 0) k = 0;                                                      // :76
    else if (c[SLOT_DOT] > half) k = 0;                                      // :78
    else if (nH > half) { k = 2; cc = nH; }
    else if (nX > half) { k = 6; cc = nX; }
    else if (nZ > half) { k = 7; cc = nZ; }
    if (k && !((a.ctx_mask >> k) & 1u)) k = 0;                               // :86
    ok[s] = k != 0;
    key[s] = ((uint32_t)p << 4) | ((uint32_t)s << 3) | (uint32_t)k;
    cov[s] = cc;                                                             // :90
    const unsigned long long hs = sum64[(s * 3 + 0) * T + p], nu = sum64[(s * 3 + 1) * T + p], de = sum64[(s * 3 + 2) * T + p];
    len[s] = (double)hs / (double)(int)cc;                                   // :92
    lm[s] = (double)nu / (double)de;                                         // :93
    nr += k != 0;
  }
  uint32_t inc = (uint32_t)nr;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int w = 0; w < NW; w++) { const uint32_t t = s_scan[w]; s_scan[w] = acc; acc += t; }
    s_scan[NW] = acc;
    uint32_t base = 0;
    if (acc) base = atomicAdd(a.cursor, acc);
    s_scan[NW + 1] = base;
    a.tile_nrow[tile] = acc;
    a.tile_base[tile] = base;
  }
  __syncthreads();
  const uint32_t total = s_scan[NW], base = s_scan[NW + 1];
  if ((uint64_t)base + total <= a.pool_cap) {
    uint32_t w = base + inc - (uint32_t)nr + s_scan[wave];
#pragma unroll
    for (int s = 0; s < 2; s++) {
      if (ok[s]) {
        a.pool_key[w] = key[s];
        a.pool_cov[w] = cov[s];
        a.pool_len[w] = len[s];
        a.pool_lmhl[w] = lm[s];
        w++;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_mhl_gather(const Tile *__restrict__ tiles, const uint32_t *__restrict__ tile_out,
                                                     const uint32_t *__restrict__ tile_base, int32_t ntiles, int64_t nrow,
                                                     const uint32_t *__restrict__ pool_key, const uint32_t *__restrict__ pool_cov,
                                                     const double *__restrict__ pool_len, const double *__restrict__ pool_lmhl,
                                                     int32_t *__restrict__ o_rname, int32_t *__restrict__ o_strand,
                                                     int32_t *__restrict__ o_pos, int32_t *__restrict__ o_ctx,
                                                     int32_t *__restrict__ o_cov, double *__restrict__ o_len,
                                                     double *__restrict__ o_lmhl) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nrow) return;
  int32_t lo = 0, hi = ntiles;
  while (hi - lo > 1) {
    const int32_t mid = (lo + hi) >> 1;
    if ((int64_t)tile_out[mid] <= i) lo = mid; else hi = mid;
  }
  const Tile td = tiles[lo];
  const uint32_t src = tile_base[lo] + (uint32_t)(i - tile_out[lo]);
  const uint32_t key = pool_key[src];
  o_rname[i] = td.rname;
  o_strand[i] = 1 + (int32_t)((key >> 3) & 1u);
  o_pos[i] = (int32_t)(td.pos0 + (int64_t)(key >> 4));
  o_ctx[i] = (int32_t)(key & 7u);
  o_cov[i] = (int32_t)pool_cov[src];
  o_len[i] = pool_len[src];
  o_lmhl[i] = pool_lmhl[src];
}

static size_t mhl_pool_rows(const epi_batch *b) { return b->pool_cap < b->pool_cap2 ? b->pool_cap : b->pool_cap2; }

static int ensure_mhl_pool(epi_batch *b, size_t rows) {
  if (rows > b->pool_cap || !b->pool_key.p) {
    EPI_TRY(b->pool_key.ensure(rows * 4));
    EPI_TRY(b->pool_a.ensure(rows * 4));
    EPI_TRY(b->pool_b.ensure(rows * 4));
    b->pool_cap = rows;
  }
  if (rows > b->pool_cap2 || !b->pool_d.p) {
    EPI_TRY(b->pool_d.ensure(rows * 8));
    EPI_TRY(b->pool_e.ensure(rows * 8));
    b->pool_cap2 = rows;
  }
  return EPI_OK;
}

static int pick_mhl_group(int32_t max_len) {
  const char *env = getenv("EPIHIP_MHL_GROUP");
  if (env) { int g = atoi(env); if (g >= 1 && g <= 64 && (g & (g - 1)) == 0) return g; }
  const int64_t chunks = max_len / 16 + 2;      // so that most reads are a single block
  int g = 1;
  while (g < chunks && g < 64) g <<= 1;
  return g;
}

}  // namespace epi

using namespace epi;

extern "C" {

int epi_batch_mhl_report_dev(epi_batch *b, const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac,
                             void *stream, int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_mhl_report_dev: NULL argument");
  *nrow_out = 0;
  b->last_kind = 0;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  uint32_t ctx_mask = 0;                                                     // :104-107
  for (const unsigned char *c = reinterpret_cast<const unsigned char *>(ctx); *c; c++) ctx_mask |= 1u << ctx_to_idx(*c);
  const uint32_t H = hmax > 0 ? (hmax < 65536 ? (uint32_t)hmax : 65536u) : 65536u;   // :112

  RowStats st;
  int32_t nt = 0;
  EPI_TRY(build_tiles(b, s, kMhlTile, &st, &nt));
  b->last_ntiles = nt;
  if (nt == 0) { b->last_kind = 2; b->last_nrow = 0; return EPI_OK; }

  // pass 1: per-read stretch sizes and haplotype info
  EPI_TRY(b->mhl_m.ensure(((size_t)b->nbytes + 64) * 2));
  EPI_TRY(b->mhl_h.ensure((size_t)b->n * 4));
  {
    const int g = pick_mhl_group(st.max_len);
    const int64_t threads = b->n * g;
    const unsigned nb = (unsigned)((threads + 255) / 256);
    prof_begin("mhl_rows", s);
#define EPI_LAUNCH(GG)                                                                                          \
  case GG:                                                                                                      \
    hipLaunchKernelGGL((k_mhl_rows<GG>), dim3(nb), dim3(256), 0, s, b->xm, b->off, b->n, ctx_mask, (int32_t)hmin, \
                       max_ooctx_meth_frac, b->mhl_m.as<uint16_t>(), b->mhl_h.as<int32_t>());                   \
    break;
    switch (g) {
      EPI_LAUNCH(1) EPI_LAUNCH(2) EPI_LAUNCH(4) EPI_LAUNCH(8) EPI_LAUNCH(16) EPI_LAUNCH(32) EPI_LAUNCH(64)
      default: return fail(EPI_ERR_ARG, "bad group size");
    }
#undef EPI_LAUNCH
    prof_end("mhl_rows", s);
    EPI_HIP(hipGetLastError());
  }

  EPI_TRY(b->tile_nrow.ensure((size_t)nt * 4));
  EPI_TRY(b->tile_base.ensure((size_t)nt * 4));
  EPI_TRY(b->tile_out.ensure((size_t)(nt + 1) * 4));
  if (mhl_pool_rows(b) == 0) EPI_TRY(ensure_mhl_pool(b, (size_t)nt * (kMhlTile / 4) + 65536));
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;

  MhlArgs a;
  a.xm = b->xm; a.m = b->mhl_m.as<uint16_t>(); a.off = b->off; a.start = b->start; a.strand = b->strand;
  a.rowinfo = b->mhl_h.as<int32_t>();
  a.tiles = b->tiles.as<Tile>();
  a.ctx_mask = ctx_mask; a.H = H;
  a.cursor = cursor;
  a.tile_nrow = b->tile_nrow.as<uint32_t>();
  a.tile_base = b->tile_base.as<uint32_t>();
  uint32_t used_total[2] = {0, 0};
  for (int attempt = 0; attempt < 2; attempt++) {
    a.pool_key = b->pool_key.as<uint32_t>();
    a.pool_cov = b->pool_a.as<uint32_t>();
    a.pool_len = b->pool_d.as<double>();
    a.pool_lmhl = b->pool_e.as<double>();
    a.pool_cap = (uint32_t)(mhl_pool_rows(b) > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : mhl_pool_rows(b));
    EPI_HIP(hipMemsetAsync(cursor, 0, 8, s));
    prof_begin("mhl_tiles", s);
    hipLaunchKernelGGL(k_mhl_tiles, dim3((unsigned)nt), dim3(MHL_WG), 0, s, a);
    prof_end("mhl_tiles", s);
    EPI_HIP(hipGetLastError());
    EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
    EPI_TRY(read_scalars(b, s, cursor, 8, used_total));
    if (used_total[0] <= a.pool_cap) break;
    if (attempt == 1) return fail(EPI_ERR_STATE, "row pool overflow after regrow");
    EPI_TRY(ensure_mhl_pool(b, (size_t)used_total[0] + (used_total[0] >> 4) + 1024));
  }
  b->last_kind = 2;
  b->last_nrow = used_total[1];
  *nrow_out = used_total[1];
  return EPI_OK;
}

int epi_batch_mhl_fetch_dev(epi_batch *b, int32_t *const d_icols[5], double *const d_dcols[2], void *stream) {
  if (!b || !d_icols || !d_dcols) return fail(EPI_ERR_ARG, "epi_batch_mhl_fetch_dev: NULL argument");
  if (b->last_kind != 2) return fail(EPI_ERR_STATE, "epi_batch_mhl_fetch_dev: no finished lMHL report on this batch");
  if (b->last_nrow == 0) return EPI_OK;
  for (int i = 0; i < 5; i++) if (!d_icols[i]) return fail(EPI_ERR_ARG, "epi_batch_mhl_fetch_dev: NULL column");
  for (int i = 0; i < 2; i++) if (!d_dcols[i]) return fail(EPI_ERR_ARG, "epi_batch_mhl_fetch_dev: NULL column");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  const unsigned nb = (unsigned)((b->last_nrow + 255) / 256);
  hipLaunchKernelGGL(k_mhl_gather, dim3(nb), dim3(256), 0, s, b->tiles.as<Tile>(), b->tile_out.as<uint32_t>(),
                     b->tile_base.as<uint32_t>(), b->last_ntiles, b->last_nrow, b->pool_key.as<uint32_t>(),
                     b->pool_a.as<uint32_t>(), b->pool_d.as<double>(), b->pool_e.as<double>(), d_icols[0], d_icols[1],
                     d_icols[2], d_icols[3], d_icols[4], d_dcols[0], d_dcols[1]);
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}

int epi_batch_mhl_fetch_host(epi_batch *b, int32_t *const h_icols[5], double *const h_dcols[2], void *stream) {
  if (!b || !h_icols || !h_dcols) return fail(EPI_ERR_ARG, "epi_batch_mhl_fetch_host: NULL argument");
  if (b->last_kind != 2) return fail(EPI_ERR_STATE, "epi_batch_mhl_fetch_host: no finished lMHL report on this batch");
  const int64_t nrow = b->last_nrow;
  if (nrow == 0) return EPI_OK;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  EPI_TRY(b->pool_c.ensure((size_t)nrow * (8 * 2 + 4 * 5) + 64));
  double *dd = b->pool_c.as<double>();
  double *dc[2] = {dd, dd + nrow};
  int32_t *di = reinterpret_cast<int32_t *>(dd + 2 * nrow);
  int32_t *ic[5];
  for (int i = 0; i < 5; i++) ic[i] = di + (int64_t)i * nrow;
  EPI_TRY(epi_batch_mhl_fetch_dev(b, ic, dc, s));
  for (int i = 0; i < 5; i++) EPI_HIP(hipMemcpyAsync(h_icols[i], ic[i], (size_t)nrow * 4, hipMemcpyDeviceToHost, s));
  for (int i = 0; i < 2; i++) EPI_HIP(hipMemcpyAsync(h_dcols[i], dc[i], (size_t)nrow * 8, hipMemcpyDeviceToHost, s));
  EPI_HIP(hipStreamSynchronize(s));
  return EPI_OK;
}

}  // extern "C"
