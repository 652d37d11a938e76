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
#include "tile_common.hpp"
#include <stdlib.h>
#include <string.h>

namespace epi {

constexpr int MHL_WG = 512;

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

  if (nblk == 1) {
    // ---- whole read inside one block of 16*G bytes: everything stays in registers ----
    const int64_t c = c0 + sub;
    const int64_t g0 = c << 4;
    const bool live = c < c1;
    uint4 w = make_uint4(0, 0, 0, 0);
    if (live) w = *reinterpret_cast<const uint4 *>(xm + g0);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    uint32_t Ub = 0, Lb = 0, Mb = 0, Nb = 0, Vb = 0;       // per-byte bit masks: member, cut, ooctx meth/unmeth, in-row
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int64_t g = g0 + i;
      const uint32_t code = (ww[i >> 2] >> (8 * (i & 3))) & 15u;
      const uint32_t inrow = (live && g >= rs && g < re) ? 1u : 0u;
      const uint32_t in = inrow & (ctx_mask >> code) & 1u;
      Vb |= inrow << i;
      Ub |= (in & (code < 8u ? 1u : 0u)) << i;
      Lb |= (in & (code >= 8u ? 1u : 0u)) << i;
      Mb |= (inrow & ~in & ((0x00E4u >> code) & 1u)) << i;   // codes 2,5,6,7   (:176)
      Nb |= (inrow & ~in & ((0xE400u >> code) & 1u)) << i;   // codes 10,13,14,15 (:177)
    }
    uint32_t h = __popc(Ub | Lb), oo_m = __popc(Mb), oo_u = __popc(Nb);
    // forward
    uint32_t a0[16], b0[16];
    Seg mf = {0u, 0u}, mb = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if ((Lb >> i) & 1u) { mf.has = 1u; mf.cnt = 0u; } else mf.cnt += (Ub >> i) & 1u;
      a0[i] = mf.cnt;
    }
#pragma unroll
    for (int i = 15; i >= 0; i--) {
      if ((Lb >> i) & 1u) { mb.has = 1u; mb.cnt = 0u; } else mb.cnt += (Ub >> i) & 1u;
      b0[i] = mb.cnt;
    }
    Seg incf = mf, incb = mb;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
      Seg l, r2;
      l.has = __shfl_up(incf.has, d, G); l.cnt = __shfl_up(incf.cnt, d, G);
      if (sub >= d) incf = seg_combine(l, incf);
      r2.has = __shfl_down(incb.has, d, G); r2.cnt = __shfl_down(incb.cnt, d, G);
      if (sub + d < G) incb = seg_combine(r2, incb);
    }
    Seg ef, eb;
    ef.has = __shfl_up(incf.has, 1, G); ef.cnt = __shfl_up(incf.cnt, 1, G);
    if (sub == 0) { ef.has = 0u; ef.cnt = 0u; }
    eb.has = __shfl_down(incb.has, 1, G); eb.cnt = __shfl_down(incb.cnt, 1, G);
    if (sub == G - 1) { eb.has = 0u; eb.cnt = 0u; }
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
    if (!keep || !live) return;
    // bits at or after the first cut / at or before the last cut of this chunk
    const uint32_t cut_f = Lb ? ~((Lb & (0u - Lb)) - 1u) : 0u;
    const uint32_t cut_b = Lb ? ((2u << (31 - __clz(Lb))) - 1u) : 0u;
    uint32_t mv[8];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      uint32_t av = a0[i], bv = b0[i];
      if (!((cut_f >> i) & 1u)) av += ef.cnt;
      if (!((cut_b >> i) & 1u)) bv += eb.cnt;
      uint32_t m = 0;
      if (!((Lb >> i) & 1u) && av > 0u && bv > 0u) { m = av + bv - ((Ub >> i) & 1u); if (m > 65535u) m = 65535u; }
      if (i & 1) mv[i >> 1] |= m << 16; else mv[i >> 1] = m;
    }
    uint16_t *dst = m_out + g0;
    if (Vb == 0xFFFFu) {                                  // whole chunk inside the read: two 16-byte stores
      reinterpret_cast<uint4 *>(dst)[0] = make_uint4(mv[0], mv[1], mv[2], mv[3]);
      reinterpret_cast<uint4 *>(dst)[1] = make_uint4(mv[4], mv[5], mv[6], mv[7]);
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++)
        if ((Vb >> i) & 1u) dst[i] = (uint16_t)((mv[i >> 1] >> (16 * (i & 1))) & 0xFFFFu);
    }
    return;
  }

  uint32_t h = 0, oo_m = 0, oo_u = 0;
  // ---- reads longer than one block: forward pass stores A(i) in m_out, backward pass turns it into m ----
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
  RowCols c;                              // pass is always null here (no lower-casing in lMHL)
  const uint16_t *m;
  const int32_t *rowinfo;
  const Tile *tiles;
  uint32_t ctx_mask, H;
  uint32_t *pool_key, *pool_cov;
  double *pool_len, *pool_lmhl;
  uint32_t pool_cap;
  uint32_t *cursor, *tile_nrow, *tile_base;
  // ultra-deep tiles are set aside and split over many workgroups (as in the CX kernel)
  int heavy_rows, heavy_chunk;
  uint32_t *heavy_count, *heavy_max, *heavy_list;
  uint32_t *heavy_cnt;                    // [heavy tile][16][T] counters
  unsigned long long *heavy_sums;         // [heavy tile][2T + 4(T+1)] numerator sums and difference arrays
  // tiles shared with other ranks of a sharded run: same two slabs, indexed by shared slot
  uint32_t *shared_cnt;
  unsigned long long *shared_sums;
};

constexpr int MHL_T = kMhlTile;
constexpr int MHL_NSUM = 2 * kMhlTile + 4 * (kMhlTile + 1);

// nibble -> flags of the rarely taken per-byte path: 1 = '+'/'-' (skipped, :187), 2/4/8 = stray nibbles
// 3/4/8, whose counter slot IS the numerator / denominator / haplotype-size sum in the reference (:190)
//   code:  0 1 2 3 4 5 6 7 | 8 9 10 11 | 12..15
//   flag:  0 0 0 2 4 0 0 0 | 8 0  0  1 |  0
constexpr uint32_t kFlagLo0 = 0x02000000u, kFlagLo1 = 0x00000004u, kFlagHi0 = 0x01000008u, kFlagHi1 = 0x00000000u;

struct MhlSlice {
  RowSlice rs;
  const uint2 *msrc;                      // this lane's first four stretch sizes (u16 each), parallel to rs.src
  int pos0;                               // tile position of byte 0 of this lane's first dword (may be -1..-3)
  int pf, pe;                             // tile positions [pf, pe) the slice covers
  int sidx;                               // 0 '+', 1 '-'
  uint32_t h;                             // haplotype size of the row
};

template <int G>
__device__ __forceinline__ MhlSlice mhl_row_slice(const MhlArgs &a, const Tile &td, int r, int sub, uint32_t *cnt) {
  MhlSlice m;
  m.rs = cx_row_slice<MHL_T, G>(a.c, td, r, sub, cnt);
  m.msrc = nullptr; m.pos0 = 0; m.pf = 0; m.pe = 0; m.sidx = 0; m.h = 0;
  if (m.rs.nd > 0) {
    const int32_t hrow = a.rowinfo[r];
    if (hrow < 0) { m.rs.nd = 0; return m; }            // read skipped by pass 1 (:179)
    m.h = (uint32_t)hrow;
    const int32_t st = a.c.start[r];
    const int64_t o = a.c.off[r];
    const int32_t len = (int32_t)((uint32_t)a.c.off[r + 1] - (uint32_t)o);
    const int32_t rel = (int32_t)((uint32_t)td.pos0 - (uint32_t)st);
    const int32_t lo = rel > 0 ? rel : 0;
    const int32_t hi = len < rel + MHL_T ? len : rel + MHL_T;
    const int64_t b0 = o + lo;
    const int32_t e_lo = (int32_t)b0 & 3;
    m.msrc = reinterpret_cast<const uint2 *>(a.m + (b0 - e_lo)) + sub;
    m.pf = lo - rel;
    m.pe = hi - rel;
    m.pos0 = m.pf - e_lo + 4 * sub;
    m.sidx = a.c.strand[r] - 1;
  }
  return m;
}

struct MhlLds {
  uint32_t *cnt;                          // [2][8][T] code counters (as the CX kernel)
  unsigned long long *num;                // [2][T]    sum of S(m_i)                       (:193)
  unsigned long long *dh, *dd;            // [2][T+1]  difference arrays of sum(h) (:192) and sum(S(h)) (:194)
};

// One dword of a row: the CX counters, plus -- only where a byte needs it -- the numerator sum, the
// corrections for skipped bytes and the stray-nibble increments.  sum(h) and sum(S(h)) are the same for
// every counted byte of a row, so they are added as an interval (+v at the slice start, -v after its end,
// prefix-summed at emit time) instead of one 64-bit atomic per byte.
template <int OFF, bool FIRST>
__device__ __forceinline__ void mhl_add_dword(uint32_t w, uint2 mm, int k, const MhlSlice &m, const MhlLds &L,
                                              uint32_t H, unsigned long long sh) {
  cx_add_dword<MHL_T, OFF, FIRST>(w, k, m.rs);
  const uint32_t c4 = w & 0x0F0F0F0Fu;
  const uint32_t lo3 = c4 & 0x07070707u;
  const uint32_t pick = 0x03020100u | ((c4 >> 1) & 0x04040404u);
  uint32_t vm = k == m.rs.nd - 1 ? m.rs.mask_last : ~0u;
  if (FIRST) vm &= m.rs.mask_first;
  const uint32_t f4 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(kFlagHi1, kFlagHi0, lo3),
                                            __builtin_amdgcn_perm(kFlagLo1, kFlagLo0, lo3), pick) & vm;
  if ((f4 | mm.x | mm.y) == 0u) return;                  // common case: nothing but the counters
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (!((vm >> (8 * j)) & 1u)) continue;               // byte outside the slice (its m may be anything)
    const uint32_t fl = (f4 >> (8 * j)) & 0xFFu;
    const uint32_t mi = ((j & 2) ? mm.y : mm.x) >> (16 * (j & 1)) & 0xFFFFu;
    const int p = m.pos0 + OFF + j;
    unsigned long long *dh = L.dh + m.sidx * (MHL_T + 1) + p;
    unsigned long long *dd = L.dd + m.sidx * (MHL_T + 1) + p;
    if (fl & 1u) {                                       // '+'/'-': not counted, take the row's interval add back
      atomicAdd(dh, 0ull - (unsigned long long)m.h); atomicAdd(dh + 1, (unsigned long long)m.h);
      atomicAdd(dd, 0ull - sh); atomicAdd(dd + 1, sh);
    } else {
      const unsigned long long ni = (mi ? mhl_lut(mi, H) : 0ull) + ((fl & 2u) ? 1ull : 0ull);
      if (ni) atomicAdd(L.num + m.sidx * MHL_T + p, ni);
      if (fl & 4u) { atomicAdd(dd, 1ull); atomicAdd(dd + 1, 0ull - 1ull); }
      if (fl & 8u) { atomicAdd(dh, 1ull); atomicAdd(dh + 1, 0ull - 1ull); }
    }
  }
}

template <int G, int WG>
__device__ __forceinline__ void mhl_accumulate(const MhlArgs &a, const Tile &td, const MhlLds &L) {
  constexpr int R = 64 / G;
  constexpr int NW = WG / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1), grp = lane / G;
  int r = td.row_lo + wave * R + grp;
  MhlSlice cur = mhl_row_slice<G>(a, td, r, sub, L.cnt);
  for (int rbase = td.row_lo + wave * R; rbase < td.row_hi; rbase += NW * R) {
    uint32_t w[CX_UN];
    uint2 mm[CX_UN];
#pragma unroll
    for (int u = 0; u < CX_UN; u++) {
      const bool on = sub + u * G < cur.rs.nd;
      w[u] = on ? cur.rs.src[u * G] : 0u;
      mm[u] = on ? cur.msrc[u * G] : make_uint2(0u, 0u);
    }
    r += NW * R;
    const MhlSlice nxt = mhl_row_slice<G>(a, td, r, sub, L.cnt);
    const unsigned long long sh = mhl_lut(cur.h, a.H);   // S(h), :194
    if (cur.rs.nd > 0 && sub == 0) {                     // the row's interval adds
      unsigned long long *dh = L.dh + cur.sidx * (MHL_T + 1);
      unsigned long long *dd = L.dd + cur.sidx * (MHL_T + 1);
      atomicAdd(dh + cur.pf, (unsigned long long)cur.h); atomicAdd(dh + cur.pe, 0ull - (unsigned long long)cur.h);
      atomicAdd(dd + cur.pf, sh); atomicAdd(dd + cur.pe, 0ull - sh);
    }
    if (sub < cur.rs.nd) mhl_add_dword<0, true>(w[0], mm[0], sub, cur, L, a.H, sh);
    if (sub + G < cur.rs.nd) mhl_add_dword<4 * G, false>(w[1], mm[1], sub + G, cur, L, a.H, sh);
    if (sub + 2 * G < cur.rs.nd) mhl_add_dword<8 * G, false>(w[2], mm[2], sub + 2 * G, cur, L, a.H, sh);
    if (sub + 3 * G < cur.rs.nd) mhl_add_dword<12 * G, false>(w[3], mm[3], sub + 3 * G, cur, L, a.H, sh);
    if (sub + 4 * G < cur.rs.nd) mhl_add_dword<16 * G, false>(w[4], mm[4], sub + 4 * G, cur, L, a.H, sh);
    for (int k = sub + CX_UN * G; k < cur.rs.nd; k += G) {
      MhlSlice t = cur;
#pragma unroll
      for (int j = 0; j < 4; j++) t.rs.dst[j] = cur.rs.dst[j] + 4 * (k - sub);
      t.pos0 = cur.pos0 + 4 * (k - sub);
      mhl_add_dword<0, false>(cur.rs.src[k - sub], cur.msrc[k - sub], k, t, L, a.H, sh);
    }
    cur = nxt;
  }
}

// inclusive prefix sum of one u64 per thread over the workgroup (NW wavefronts)
template <int NW>
__device__ __forceinline__ unsigned long long block_incl_scan_u64(unsigned long long v, unsigned long long *s_w) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t lo = __shfl_up((uint32_t)v, d, 64);
    const uint32_t hi = __shfl_up((uint32_t)(v >> 32), d, 64);
    if (lane >= d) v += ((unsigned long long)hi << 32) | lo;
  }
  if (lane == 63) s_w[wave] = v;
  __syncthreads();
  unsigned long long add = 0;
  for (int w = 0; w < wave; w++) add += s_w[w];
  __syncthreads();
  return v + add;
}

// Rule, prefix sums of the interval arrays, ordered compaction of one tile (one position per thread).
template <int WG>
__device__ __forceinline__ void mhl_emit(const MhlArgs &a, int tile, const MhlLds &L, unsigned long long *s_w,
                                         uint32_t *s_scan) {
  constexpr int T = MHL_T;
  constexpr int NW = WG / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // emit: one position per thread, '+' then '-'
  const int p = threadIdx.x;
  const bool live = p < T;
  uint32_t key[2], cov[2];
  double len[2], lm[2];
  bool ok[2];
  int nr = 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const unsigned long long hs = block_incl_scan_u64<NW>(live ? L.dh[s * (T + 1) + p] : 0ull, s_w);   // :192
    const unsigned long long de = block_incl_scan_u64<NW>(live ? L.dd[s * (T + 1) + p] : 0ull, s_w);   // :194
    uint32_t c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = live ? L.cnt[(s * 8 + k) * T + p] : 0u;
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
    const unsigned long long nu = live ? L.num[s * T + p] : 0ull;
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

template <int G, int WG>
__global__ __launch_bounds__(WG) void k_mhl_tiles(MhlArgs a, int ntiles) {
  constexpr int T = MHL_T;
  constexpr int NW = WG / 64;
  static_assert(WG >= T, "one position per thread in the emit phase");
  __shared__ __attribute__((aligned(16))) uint32_t cnt_raw[16 * T + 2 * kCxGuard];
  __shared__ __attribute__((aligned(16))) unsigned long long sums[MHL_NSUM];
  __shared__ unsigned long long s_w[NW];
  __shared__ uint32_t s_scan[NW + 2];
  MhlLds L;
  L.cnt = cnt_raw + kCxGuard;
  L.num = sums;
  L.dh = sums + 2 * T;
  L.dd = sums + 2 * T + 2 * (T + 1);
  const int chunk = (ntiles + 7) >> 3;                   // XCD-aware tile order, as the CX kernel
  const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const Tile td = a.tiles[tile];
  if (td.row_hi - td.row_lo > a.heavy_rows) {            // pile-up: k_mhl_heavy splits it by row chunks
    if (threadIdx.x == 0) {
      const uint32_t h = atomicAdd(a.heavy_count, 1u);
      a.heavy_list[h] = (uint32_t)tile;
      atomicMax(a.heavy_max, (uint32_t)(td.row_hi - td.row_lo));
      a.tile_nrow[tile] = 0;
      a.tile_base[tile] = 0;
    }
    return;
  }
  for (int i = threadIdx.x; i < 16 * T + 2 * kCxGuard; i += WG) cnt_raw[i] = 0;
  for (int i = threadIdx.x; i < MHL_NSUM; i += WG) sums[i] = 0ull;
  __syncthreads();
  mhl_accumulate<G, WG>(a, td, L);
  __syncthreads();
  if (td.slot >= 0) {                                    // shared with another rank: hand the raw sums over
    uint32_t *dc = a.shared_cnt + (int64_t)td.slot * (16 * T);
    unsigned long long *ds = a.shared_sums + (int64_t)td.slot * MHL_NSUM;
    for (int i = threadIdx.x; i < 16 * T; i += WG) { const uint32_t v = L.cnt[i]; if (v) atomicAdd(dc + i, v); }
    for (int i = threadIdx.x; i < MHL_NSUM; i += WG) { const unsigned long long v = sums[i]; if (v) atomicAdd(ds + i, v); }
    if (threadIdx.x == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
    return;
  }
  mhl_emit<WG>(a, tile, L, s_w, s_scan);
}

// One chunk of the candidate rows of one heavy tile -> added into that tile's slab in HBM.
template <int G, int WG>
__global__ __launch_bounds__(WG) void k_mhl_heavy(MhlArgs a) {
  constexpr int T = MHL_T;
  __shared__ __attribute__((aligned(16))) uint32_t cnt_raw[16 * T + 2 * kCxGuard];
  __shared__ __attribute__((aligned(16))) unsigned long long sums[MHL_NSUM];
  MhlLds L;
  L.cnt = cnt_raw + kCxGuard;
  L.num = sums;
  L.dh = sums + 2 * T;
  L.dd = sums + 2 * T + 2 * (T + 1);
  const int tile = (int)a.heavy_list[blockIdx.y];
  Tile td = a.tiles[tile];
  const int lo = td.row_lo + (int)blockIdx.x * a.heavy_chunk;
  if (lo >= td.row_hi) return;
  td.row_lo = lo;
  if (td.row_hi - lo > a.heavy_chunk) td.row_hi = lo + a.heavy_chunk;
  for (int i = threadIdx.x; i < 16 * T + 2 * kCxGuard; i += WG) cnt_raw[i] = 0;
  for (int i = threadIdx.x; i < MHL_NSUM; i += WG) sums[i] = 0ull;
  __syncthreads();
  mhl_accumulate<G, WG>(a, td, L);
  __syncthreads();
  uint32_t *dc = td.slot >= 0 ? a.shared_cnt + (int64_t)td.slot * (16 * T) : a.heavy_cnt + (int64_t)blockIdx.y * (16 * T);
  unsigned long long *ds = td.slot >= 0 ? a.shared_sums + (int64_t)td.slot * MHL_NSUM : a.heavy_sums + (int64_t)blockIdx.y * MHL_NSUM;
  for (int i = threadIdx.x; i < 16 * T; i += WG) { const uint32_t v = L.cnt[i]; if (v) atomicAdd(dc + i, v); }
  for (int i = threadIdx.x; i < MHL_NSUM; i += WG) { const unsigned long long v = sums[i]; if (v) atomicAdd(ds + i, v); }
}

template <int WG>
__global__ __launch_bounds__(WG) void k_mhl_emit_heavy(MhlArgs a) {
  constexpr int T = MHL_T;
  constexpr int NW = WG / 64;
  __shared__ __attribute__((aligned(16))) uint32_t cnt[16 * T];
  __shared__ __attribute__((aligned(16))) unsigned long long sums[MHL_NSUM];
  __shared__ unsigned long long s_w[NW];
  __shared__ uint32_t s_scan[NW + 2];
  MhlLds L;
  L.cnt = cnt;
  L.num = sums;
  L.dh = sums + 2 * T;
  L.dd = sums + 2 * T + 2 * (T + 1);
  const int tile = (int)a.heavy_list[blockIdx.x];
  if (a.tiles[tile].slot >= 0) return;                   // emitted after the cross-rank reduce
  const uint32_t *sc = a.heavy_cnt + (int64_t)blockIdx.x * (16 * T);
  const unsigned long long *ss = a.heavy_sums + (int64_t)blockIdx.x * MHL_NSUM;
  for (int i = threadIdx.x; i < 16 * T; i += WG) cnt[i] = sc[i];
  for (int i = threadIdx.x; i < MHL_NSUM; i += WG) sums[i] = ss[i];
  __syncthreads();
  mhl_emit<WG>(a, tile, L, s_w, s_scan);
}

// Emits the shared tiles this rank owns from the (already cross-rank reduced) slabs: one workgroup per slot.
template <int WG>
__global__ __launch_bounds__(WG) void k_mhl_emit_slab(MhlArgs a, const int32_t *__restrict__ owned,
                                                       const int32_t *__restrict__ slot_tile) {
  constexpr int T = MHL_T;
  constexpr int NW = WG / 64;
  __shared__ __attribute__((aligned(16))) uint32_t cnt[16 * T];
  __shared__ __attribute__((aligned(16))) unsigned long long sums[MHL_NSUM];
  __shared__ unsigned long long s_w[NW];
  __shared__ uint32_t s_scan[NW + 2];
  if (!owned[blockIdx.x]) return;
  const int tile = slot_tile[blockIdx.x];
  if (tile < 0) return;
  MhlLds L;
  L.cnt = cnt;
  L.num = sums;
  L.dh = sums + 2 * T;
  L.dd = sums + 2 * T + 2 * (T + 1);
  const uint32_t *sc = a.shared_cnt + (int64_t)blockIdx.x * (16 * T);
  const unsigned long long *ss = a.shared_sums + (int64_t)blockIdx.x * MHL_NSUM;
  for (int i = threadIdx.x; i < 16 * T; i += WG) cnt[i] = sc[i];
  for (int i = threadIdx.x; i < MHL_NSUM; i += WG) sums[i] = ss[i];
  __syncthreads();
  mhl_emit<WG>(a, tile, L, s_w, s_scan);
}

// one wavefront per tile: pool rows -> their place in the final table (see k_cx_gather)
__global__ __launch_bounds__(256) void k_mhl_gather(const Tile *__restrict__ tiles, const uint32_t *__restrict__ tile_out,
                                                     const uint32_t *__restrict__ tile_nrow, const uint32_t *__restrict__ tile_base,
                                                     int32_t ntiles, const uint32_t *__restrict__ pool_key,
                                                     const uint32_t *__restrict__ pool_cov, const double *__restrict__ pool_len,
                                                     const double *__restrict__ pool_lmhl, int32_t *__restrict__ o_rname,
                                                     int32_t *__restrict__ o_strand, int32_t *__restrict__ o_pos,
                                                     int32_t *__restrict__ o_ctx, int32_t *__restrict__ o_cov,
                                                     double *__restrict__ o_len, double *__restrict__ o_lmhl) {
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const uint32_t n = tile_nrow[tile];
  if (n == 0) return;
  const int lane = threadIdx.x & 63;
  const Tile td = tiles[tile];
  const uint32_t src0 = tile_base[tile], dst0 = tile_out[tile];
  for (uint32_t i = lane; i < n; i += 64) {
    const uint32_t key = pool_key[src0 + i];
    const uint32_t o = dst0 + i;
    o_rname[o] = td.rname;
    o_strand[o] = 1 + (int32_t)((key >> 3) & 1u);
    o_pos[o] = (int32_t)(td.pos0 + (int64_t)(key >> 4));
    o_ctx[o] = (int32_t)(key & 7u);
    o_cov[o] = (int32_t)pool_cov[src0 + i];
    o_len[o] = pool_len[src0 + i];
    o_lmhl[o] = pool_lmhl[src0 + i];
  }
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

static void launch_mhl_tiles(int g, int nt, hipStream_t s, const MhlArgs &a) {
  const unsigned grid = (unsigned)(((nt + 7) / 8) * 8);
  switch (g) {
    case 8: hipLaunchKernelGGL((k_mhl_tiles<8, MHL_WG>), dim3(grid), dim3(MHL_WG), 0, s, a, nt); break;
    case 16: hipLaunchKernelGGL((k_mhl_tiles<16, MHL_WG>), dim3(grid), dim3(MHL_WG), 0, s, a, nt); break;
    case 32: hipLaunchKernelGGL((k_mhl_tiles<32, MHL_WG>), dim3(grid), dim3(MHL_WG), 0, s, a, nt); break;
    default: hipLaunchKernelGGL((k_mhl_tiles<64, MHL_WG>), dim3(grid), dim3(MHL_WG), 0, s, a, nt); break;
  }
}

static void launch_mhl_heavy(int g, uint32_t nheavy, uint32_t nchunks, hipStream_t s, const MhlArgs &a) {
  const dim3 grid(nchunks, nheavy);
  switch (g) {
    case 8: hipLaunchKernelGGL((k_mhl_heavy<8, MHL_WG>), grid, dim3(MHL_WG), 0, s, a); break;
    case 16: hipLaunchKernelGGL((k_mhl_heavy<16, MHL_WG>), grid, dim3(MHL_WG), 0, s, a); break;
    case 32: hipLaunchKernelGGL((k_mhl_heavy<32, MHL_WG>), grid, dim3(MHL_WG), 0, s, a); break;
    default: hipLaunchKernelGGL((k_mhl_heavy<64, MHL_WG>), grid, dim3(MHL_WG), 0, s, a); break;
  }
  hipLaunchKernelGGL((k_mhl_emit_heavy<MHL_WG>), dim3(nheavy), dim3(MHL_WG), 0, s, a);
}

// lanes per row in the tile kernel (as pick_cx_group)
static int pick_mhl_tile_group(int32_t max_len) {
  const char *env = getenv("EPIHIP_MHL_TILE_GROUP");
  if (env) { int g = atoi(env); if (g == 8 || g == 16 || g == 32 || g == 64) return g; }
  const int slice = (max_len < MHL_T ? max_len : MHL_T) + 3;
  const int nd = (slice + 3) / 4;
  int g = 8;
  while (g < 64 && g * CX_UN < nd) g <<= 1;
  return g;
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
  a.c.xm = b->xm; a.c.off = b->off; a.c.start = b->start; a.c.strand = b->strand; a.c.pass = nullptr;
  a.m = b->mhl_m.as<uint16_t>();
  a.rowinfo = b->mhl_h.as<int32_t>();
  a.tiles = b->tiles.as<Tile>();
  a.ctx_mask = ctx_mask; a.H = H;
  a.cursor = cursor;
  a.tile_nrow = b->tile_nrow.as<uint32_t>();
  a.tile_base = b->tile_base.as<uint32_t>();
  a.heavy_rows = 16384;
  if (const char *env = getenv("EPIHIP_HEAVY_ROWS")) { const int v = atoi(env); if (v > 0) a.heavy_rows = v; }
  a.heavy_chunk = a.heavy_rows / 4 > 64 ? a.heavy_rows / 4 : 64;
  EPI_TRY(b->heavy_list.ensure((size_t)nt * 4));
  a.heavy_list = b->heavy_list.as<uint32_t>();
  a.heavy_count = b->misc.as<uint32_t>() + 3;             // misc layout as in the CX report
  a.heavy_max = b->misc.as<uint32_t>() + 8;
  a.heavy_cnt = nullptr;
  a.heavy_sums = nullptr;
  a.shared_cnt = reinterpret_cast<uint32_t *>(b->d_mhl_cnt_slab);
  a.shared_sums = reinterpret_cast<unsigned long long *>(b->d_mhl_sum_slab);
  const int32_t nshared = (int32_t)b->shared_keys.size();
  if (nshared > 0 && (!a.shared_cnt || !a.shared_sums)) return fail(EPI_ERR_STATE, "shared tiles set without lMHL slabs (use epi_batch_mhl_set_shared)");
  const size_t headroom = nshared > 0 ? (size_t)nshared * 2 * MHL_T : 0;
  b->mhl_ctx_mask = ctx_mask;
  uint32_t used_total[2] = {0, 0};
  const int tg = pick_mhl_tile_group(st.max_len);
  for (int attempt = 0; attempt < 2; attempt++) {
    a.pool_key = b->pool_key.as<uint32_t>();
    a.pool_cov = b->pool_a.as<uint32_t>();
    a.pool_len = b->pool_d.as<double>();
    a.pool_lmhl = b->pool_e.as<double>();
    a.pool_cap = (uint32_t)(mhl_pool_rows(b) > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : mhl_pool_rows(b));
    EPI_HIP(hipMemsetAsync(cursor, 0, 12, s));           // cursor, total, heavy count
    EPI_HIP(hipMemsetAsync(a.heavy_max, 0, 4, s));
    prof_begin("mhl_tiles", s);
    launch_mhl_tiles(tg, nt, s, a);
    prof_end("mhl_tiles", s);
    EPI_HIP(hipGetLastError());
    EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
    uint32_t host[8];
    EPI_TRY(read_scalars(b, s, cursor, 32, host));         // misc[1..8]
    if (host[2] > 0) {                                     // pile-ups: split, reduce in HBM, emit, rescan
      const uint32_t nheavy = host[2], nchunks = (host[7] + (uint32_t)a.heavy_chunk - 1) / (uint32_t)a.heavy_chunk;
      EPI_TRY(b->heavy_slab.ensure((size_t)nheavy * 16 * MHL_T * 4));
      EPI_TRY(b->heavy_sums.ensure((size_t)nheavy * MHL_NSUM * 8));
      a.heavy_cnt = b->heavy_slab.as<uint32_t>();
      a.heavy_sums = b->heavy_sums.as<unsigned long long>();
      EPI_HIP(hipMemsetAsync(a.heavy_cnt, 0, (size_t)nheavy * 16 * MHL_T * 4, s));
      EPI_HIP(hipMemsetAsync(a.heavy_sums, 0, (size_t)nheavy * MHL_NSUM * 8, s));
      prof_begin("mhl_heavy", s);
      launch_mhl_heavy(tg, nheavy, nchunks, s, a);
      prof_end("mhl_heavy", s);
      EPI_HIP(hipGetLastError());
      EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
      EPI_TRY(read_scalars(b, s, cursor, 8, host));
    }
    used_total[0] = host[0];
    used_total[1] = host[1];
    if ((size_t)used_total[0] + headroom <= a.pool_cap) break;
    if (attempt == 1) return fail(EPI_ERR_STATE, "row pool overflow after regrow");
    EPI_TRY(ensure_mhl_pool(b, (size_t)used_total[0] + (used_total[0] >> 4) + 1024 + headroom));
    if (nshared > 0) {   // the rerun adds into the shared slabs again
      EPI_HIP(hipMemsetAsync(a.shared_cnt, 0, (size_t)nshared * 16 * MHL_T * 4, s));
      EPI_HIP(hipMemsetAsync(a.shared_sums, 0, (size_t)nshared * MHL_NSUM * 8, s));
    }
  }
  if (nshared > 0) { b->last_kind = 4; return EPI_OK; }     // caller continues with epi_batch_mhl_finish_shared
  b->last_kind = 2;
  b->last_nrow = used_total[1];
  *nrow_out = used_total[1];
  return EPI_OK;
}

int epi_mhl_tile_positions(void) { return MHL_T; }
int epi_mhl_slab_sums(void) { return MHL_NSUM; }

int epi_batch_mhl_set_shared(epi_batch *b, const int64_t *h_keys, const int32_t *h_owned, int32_t nshared,
                             int32_t *d_cnt_slab, int64_t *d_sum_slab) {
  if (nshared > 0 && (!d_cnt_slab || !d_sum_slab)) return fail(EPI_ERR_ARG, "epi_batch_mhl_set_shared: NULL slab");
  // same key/owner bookkeeping as the CX report (the slot of a tile is assigned when the tile table is built)
  int32_t dummy = 0;
  EPI_TRY(epi_batch_cx_set_shared(b, h_keys, h_owned, nshared, nshared > 0 ? &dummy : nullptr));
  b->d_slab = nullptr;
  b->d_mhl_cnt_slab = nshared > 0 ? d_cnt_slab : nullptr;
  b->d_mhl_sum_slab = nshared > 0 ? d_sum_slab : nullptr;
  return EPI_OK;
}

// Second half of a sharded lMHL report: the slabs have been sum-reduced across ranks.
int epi_batch_mhl_finish_shared(epi_batch *b, void *stream, int64_t *nrow_out) {
  if (!b || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_mhl_finish_shared: NULL argument");
  if (b->last_kind != 4) return fail(EPI_ERR_STATE, "epi_batch_mhl_finish_shared without a sharded epi_batch_mhl_report_dev");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  const int32_t nt = b->last_ntiles;
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;
  MhlArgs a;
  memset(&a, 0, sizeof(a));
  a.tiles = b->tiles.as<Tile>();
  a.ctx_mask = b->mhl_ctx_mask;
  a.cursor = cursor;
  a.tile_nrow = b->tile_nrow.as<uint32_t>();
  a.tile_base = b->tile_base.as<uint32_t>();
  a.shared_cnt = reinterpret_cast<uint32_t *>(b->d_mhl_cnt_slab);
  a.shared_sums = reinterpret_cast<unsigned long long *>(b->d_mhl_sum_slab);
  a.pool_key = b->pool_key.as<uint32_t>();
  a.pool_cov = b->pool_a.as<uint32_t>();
  a.pool_len = b->pool_d.as<double>();
  a.pool_lmhl = b->pool_e.as<double>();
  a.pool_cap = (uint32_t)(mhl_pool_rows(b) > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : mhl_pool_rows(b));
  hipLaunchKernelGGL((k_mhl_emit_slab<MHL_WG>), dim3((unsigned)b->shared_keys.size()), dim3(MHL_WG), 0, s, a,
                     b->d_shared_owned.as<int32_t>(), b->d_slot_tile.as<int32_t>());
  EPI_HIP(hipGetLastError());
  EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
  uint32_t ut[2] = {0, 0};
  EPI_TRY(read_scalars(b, s, cursor, 8, ut));
  if (ut[0] > a.pool_cap) return fail(EPI_ERR_STATE, "row pool overflow in sharded lMHL report");
  b->last_kind = 2;
  b->last_nrow = ut[1];
  *nrow_out = ut[1];
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
  const unsigned nb = (unsigned)((b->last_ntiles + 3) / 4);
  hipLaunchKernelGGL(k_mhl_gather, dim3(nb), dim3(256), 0, s, b->tiles.as<Tile>(), b->tile_out.as<uint32_t>(),
                     b->tile_nrow.as<uint32_t>(), b->tile_base.as<uint32_t>(), b->last_ntiles, b->pool_key.as<uint32_t>(),
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
