// rcpp_mhl_report (src/rcpp_mhl_report.cpp:46-228) on the GPU: linearised
// Methylated Haplotype Load per cytosine.
//
// Pass 1 of the reference (:158-182) is a sequential run-length walk over each
// read: in-context bases are the haplotype; a maximal run of methylated
// (upper-case) in-context bases, not interrupted by an unmethylated in-context
// base ("cut"), with M members gives num[i] = S(M) to EVERY byte between its first
// and last member.  Pass 2 (:185-195) adds, for every counted byte (code != 11)
// of a kept read, its code counter, h (:192), num[i] (:193) and S(h) (:194) to
// the byte's (pos,strand).  All three sums are constant over intervals of a read,
// so nothing is stored per byte here:
//
//  k_mhl_rows    G lanes own a read, 32 to 64 bytes per lane.  Bit masks per lane (member,
//                cut, skipped) come from a v_perm LUT; two segmented scans over the
//                lanes give every lane the members of its open segment to the left
//                and to the right; the spans are then found bit-parallel inside the
//                lane (segmented fill of the member bits up to the next cut, both
//                directions) and written as records (first, last, M), one per run of
//                counted span bytes of a lane.  Reads with skipped bytes also get
//                their counted runs as records (M = 0).  Per read: h, or -1 when the
//                read is dropped (h < hmin or out-of-context beta too high, :176-179).
//  k_mhl_tiles   the CX tile kernel (packed LDS histogram, tile_common.hpp) plus three
//                DIFFERENCE arrays per strand in LDS (u32 when no position can reach
//                2^31, else u64): a read adds +h/-h and +S(h)/-S(h) at the ends of its
//                in-tile slice and +S(M)/-S(M) at the ends of every record that reaches
//                into the tile; the emit phase prefix-sums them (DPP wave scans).
//                S(n) = n(n+1)(n+2)/6 is computed arithmetically (no 64K-entry table).
//                512-position tiles, 30 (u32) or 44 (u64) KiB of LDS, three workgroups
//                per CU; rows go to the tile's own slot of the row pool.
// Lane exchanges inside groups of up to 16 lanes are DPP moves, not __shfl (ds_bpermute).
#include "common.hpp"
#include "tile_common.hpp"
#include "mhl_common.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace epi {



struct MhlRec { uint32_t first, last, m; };       // bytes [first,last] of the row; m = members of the stretch, 0 = counted run



struct RowsArgs {
  const uint8_t *xm;
  const int64_t *off;                     // row r owns xm[off[r] .. off[r] + len[r])
  const int32_t *len;
  int64_t n;
  MhlLut lut;
  int32_t hmin;
  double max_oo;
  int4 *rowinfo;                          // per read: h or -1 (dropped), 1 if it has skipped bytes, (first record, records) of the read
  uint2 *blkrec;                          // multi: (first record, records) per 2 KiB block of a read (rowinfo.z/.w unused)
  MhlRec *recs;
  uint32_t rec_cap;
  unsigned long long *rec_cursor;         // MHL_REGIONS cursors, MHL_CUR_STRIDE apart: a workgroup takes its records from region
                                          // blockIdx % MHL_REGIONS (one cursor for all serialises 3 M atomics: 25 ms); a cursor
                                          // may run past its region's capacity rec_cap / MHL_REGIONS: the caller regrows and reruns
  uint32_t *cont;                         // multi: members entering a block from the right
  uint32_t *max_h;                        // largest haplotype size among the kept reads (sizes the LDS sums of pass 2)
};



template <int C>
__device__ __forceinline__ ChunkRaw<C> mhl_chunk_load(const uint8_t *__restrict__ xm, int64_t g0, bool live, int64_t rs, int64_t re) {
  constexpr int W = 16 * C;
  ChunkRaw<C> r;
  r.lo = 0; r.hi = 0;
#pragma unroll
  for (int j = 0; j < 4 * C; j++) r.ww[j] = 0u;
  if (!live) return r;
  int64_t lo = rs - g0, hi = re - g0;
  if (lo < 0) lo = 0;
  if (hi > W) hi = W;
  if (hi <= lo) return r;
  r.lo = (int)lo; r.hi = (int)hi;
#pragma unroll
  for (int j = 0; j < C; j++) {
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (j == 0 || g0 + 16 * j < re) w = *reinterpret_cast<const uint4 *>(xm + g0 + 16 * j);
    r.ww[4 * j] = w.x; r.ww[4 * j + 1] = w.y; r.ww[4 * j + 2] = w.z; r.ww[4 * j + 3] = w.w;
  }
  return r;
}

// ... and turned into per-byte bit masks.
template <int C>
__device__ __forceinline__ Chunk<typename MaskOf<C>::T> mhl_chunk_masks(const ChunkRaw<C> &r, const MhlLut &lut) {
  using M = typename MaskOf<C>::T;
  constexpr int W = 16 * C;
  Chunk<M> c = {0, 0, 0, 0, 0u, 0u};
  if (r.hi <= r.lo) return c;
  const int lo = r.lo, hi = r.hi;
  c.V = bm_below<M>(hi) & ~bm_below<M>(lo);
  // bytes outside the row get flag 0: byte masks of the row within each dword (only edge chunks need them)
  const bool edge = lo > 0 || hi < W;
  uint32_t f8[4 * C], kacc = 0, cm = 0, cn = 0;
  // Bit planes by v_dot4_u32_u8: the flag bit of four bytes times the weights (1,2,4,8) is their nibble; a second
  // dword with weights (16,32,64,128) completes a mask byte, which is shifted into place (the cut flag is bit 1, so
  // its sums come out doubled: one bit further down).  Counts of the out-of-context flags the same way with weights 1.
  uint32_t ulo = 0, uhi = 0, llo = 0, lhi = 0;           // mask bits 0-31 / 32-63
#pragma unroll
  for (int e = 0; e < 2 * C; e++) {
    uint32_t f[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int d = 2 * e + h;
      const uint32_t lo3 = r.ww[d] & 0x07070707u;           // low three bits of the codes; bit 3 picks the LUT half
      const uint32_t pick = ((r.ww[d] >> 1) & 0x04040404u) | 0x03020100u;
      uint32_t v = __builtin_amdgcn_perm(__builtin_amdgcn_perm(lut.hi1, lut.hi0, lo3),
                                         __builtin_amdgcn_perm(lut.lo1, lut.lo0, lo3), pick);
      if (edge) {
        int a = lo - 4 * d, b = hi - 4 * d;                   // valid bytes [a, b) of this dword
        a = a < 0 ? 0 : (a > 4 ? 4 : a);
        b = b < 0 ? 0 : (b > 4 ? 4 : b);
        const uint32_t bm = b > a ? ((b >= 4 ? ~0u : ((1u << (8 * b)) - 1u)) & ~((1u << (8 * a)) - 1u)) : 0u;
        v &= bm;
      }
      f[h] = v;
      f8[d] = v;
      kacc |= v;
      cm = __builtin_amdgcn_udot4(v & 0x08080808u, 0x01010101u, cm, false);     // 8 x count
      cn = __builtin_amdgcn_udot4(v & 0x10101010u, 0x01010101u, cn, false);     // 16 x count
    }
    const uint32_t ub = __builtin_amdgcn_udot4(f[1] & 0x01010101u, 0x80402010u,
                                               __builtin_amdgcn_udot4(f[0] & 0x01010101u, 0x08040201u, 0u, false), false);
    const uint32_t lb2 = __builtin_amdgcn_udot4(f[1] & 0x02020202u, 0x80402010u,
                                                __builtin_amdgcn_udot4(f[0] & 0x02020202u, 0x08040201u, 0u, false), false);
    const int sh = 8 * (e & 3);
    if (e < 4) { ulo |= ub << sh; llo |= sh ? lb2 << (sh - 1) : lb2 >> 1; }
    else { uhi |= ub << sh; lhi |= sh ? lb2 << (sh - 1) : lb2 >> 1; }
  }
  c.U = (M)ulo; c.L = (M)llo;
  if constexpr (sizeof(M) == 8) { c.U |= (M)uhi << 32; c.L |= (M)lhi << 32; }
  c.oom = cm >> 3;
  c.oou = cn >> 4;
  if (kacc & 0x04040404u) {                                   // skipped bytes are rare ('+'/'-', filler between mates)
#pragma unroll
    for (int d = 0; d < 4 * C; d++) c.K |= (M)plane_nibble(f8[d], 2) << (4 * d);
  }
  return c;
}

template <int C>
__device__ __forceinline__ Chunk<typename MaskOf<C>::T> mhl_chunk(const uint8_t *__restrict__ xm, int64_t g0, bool live,
                                                                   int64_t rs, int64_t re, const MhlLut &lut) {
  return mhl_chunk_masks<C>(mhl_chunk_load<C>(xm, g0, live, rs, re), lut);
}



// Span bytes of the chunk: bytes that have a member of their stretch at or before them AND at or after them
// (segmented fills of the member bits, stopped by cuts; `enter` / `cont` = members of the open segment in the lanes
// to the left / right).  Counted span bytes are what pass 2 adds S(M) for.
template <int W, class M>
__device__ __forceinline__ M span_bits(const Chunk<M> &c, uint32_t enter, uint32_t cont) {
  const M nl = ~c.L & bm_below<M>(W);
  M x = c.U | ((enter > 0u && !(c.L & (M)1)) ? (M)1 : (M)0);
  M p = nl;
#pragma unroll
  for (int sft = 1; sft < W; sft <<= 1) { x |= (x << sft) & p; p &= p << sft; }
  M y = c.U | ((cont > 0u && !((c.L >> (W - 1)) & (M)1)) ? ((M)1 << (W - 1)) : (M)0);
  M q = nl;
#pragma unroll
  for (int sft = 1; sft < W; sft <<= 1) { y |= (y >> sft) & q; q &= q >> sft; }
  return x & y & nl & ~c.K & c.V;
}

template <class M> __device__ __forceinline__ uint32_t run_count(M bits) { return (uint32_t)bm_popc(bits & ~(bits << 1)); }

// Writes one record per run of set bits of P (stretch pieces: m from the run's segment) or Q (counted runs, m = 0).
template <int W, class M>
__device__ __forceinline__ void write_runs(M bits, bool stretch, const Chunk<M> &c, uint32_t enter, uint32_t cont,
                                           uint32_t row_off0, MhlRec *__restrict__ out) {
  while (bits) {
    const int f = bm_ctz(bits);
    const M t = ~(bits >> f);
    const int e = t ? bm_ctz(t) : (int)(8 * sizeof(M)) - f;    // run length
    uint32_t m = 0;
    if (stretch) {
      const M lc = c.L & bm_below<M>(f);
      const int a = lc ? bm_msb(lc) + 1 : 0;                   // segment = bits [a, b) between the surrounding cuts
      const int end = f + e;
      const M hc = end < W ? (c.L >> end) : (M)0;
      const int b = hc ? end + bm_ctz(hc) : W;
      const M segmask = bm_below<M>(b) & ~bm_below<M>(a);
      m = (a == 0 ? enter : 0u) + (uint32_t)bm_popc(c.U & segmask) + (b == W ? cont : 0u);
    }
    MhlRec r;
    r.first = row_off0 + (uint32_t)f;
    r.last = r.first + (uint32_t)e - 1u;
    r.m = m;
    *out++ = r;
    bits &= ~(bm_below<M>(e) << f);
  }
}

__device__ __forceinline__ bool mhl_keep(uint32_t h, uint32_t oo_m, uint32_t oo_u, int32_t hmin, double max_oo) {
  const double frac = (double)oo_m / (double)((uint64_t)oo_m + oo_u);      // :178 (0/0 = NaN -> kept)
  return !((int)h < hmin || frac > max_oo);                                // :179
}



// Reads that fit one block of G lanes x 16*C bytes (every read of a short-read batch): 256/G reads per workgroup.
template <int G, int C>
__global__ __launch_bounds__(256) void k_mhl_rows(RowsArgs a) {
  using M = typename MaskOf<C>::T;
  constexpr int W = 16 * C;
  __shared__ uint32_t s_w[5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1);
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const bool valid = row < a.n;
  int64_t rs = 0, re = 0;
  if (valid) { rs = a.off[row]; re = rs + a.len[row]; }
  const int64_t g0 = ((rs >> 4) << 4) + (int64_t)sub * W;     // the read starts somewhere in lane 0's first 16 bytes
  const Chunk<M> c = mhl_chunk<C>(a.xm, g0, g0 < re, rs, re, a.lut);

  // members of the open segment to the left (enter) and to the right (cont) of this lane
  Seg pf = {c.L ? 1u : 0u, trail_members(c)}, sf = {c.L ? 1u : 0u, lead_members(c)};
  seg_scan_steps<G, 1>(pf, sf, sub);
  uint32_t enter = grp_up<G, 1>(pf.cnt), cont = grp_down<G, 1>(sf.cnt);
  if (sub == 0) enter = 0u;
  if (sub == G - 1) cont = 0u;
  const uint32_t h = grp_sum<G / 2>((uint32_t)bm_popc(c.U | c.L)), oo_m = grp_sum<G / 2>(c.oom), oo_u = grp_sum<G / 2>(c.oou);
  const uint32_t anyk = grp_or<G / 2>(c.K ? 1u : 0u);
  const bool keep = valid && mhl_keep(h, oo_m, oo_u, a.hmin, a.max_oo);
  // the current maximum is read from L2 (an L1 copy would stay 0 and every read would issue the atomic: 67 ms)
  if (keep && sub == 0 && h > __hip_atomic_load(a.max_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.max_h, h);
  const M P = keep ? span_bits<W>(c, enter, cont) : (M)0;
  const M Q = (keep && anyk) ? (c.V & ~c.K) : (M)0;
  const uint32_t nrec = run_count(P) + run_count(Q);

  // record slots: exclusive scan over the workgroup, one cursor atomic per workgroup
  const uint32_t inc = wave_scan_u32(nrec);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int w = 0; w < 4; w++) { const uint32_t t = s_w[w]; s_w[w] = acc; acc += t; }
    const uint32_t region = blockIdx.x & (MHL_REGIONS - 1), region_cap = a.rec_cap / MHL_REGIONS;
    unsigned long long base = 0;
    if (acc) base = atomicAdd(a.rec_cursor + region * MHL_CUR_STRIDE, (unsigned long long)acc);
    s_w[4] = (base + acc <= (unsigned long long)region_cap) ? region * region_cap + (uint32_t)base : 0xFFFFFFFFu;   // does not fit: count only
  }
  __syncthreads();
  const uint32_t base = s_w[4];
  const uint32_t my = base + s_w[wave] + inc - nrec;
  const uint32_t row_n = grp_sum<G / 2>(nrec);
  if (valid && sub == 0)
    a.rowinfo[row] = make_int4(keep ? (int32_t)h : -1, (int32_t)anyk, (int32_t)my, base == 0xFFFFFFFFu ? 0 : (int32_t)row_n);
  if (nrec && base != 0xFFFFFFFFu) {
    const uint32_t off0 = (uint32_t)(g0 - rs);               // row offset of the chunk's byte 0 (wraps for the first chunk)
    MhlRec *out = a.recs + my;
    write_runs<W>(P, true, c, enter, cont, off0, out);
    write_runs<W>(Q, false, c, enter, cont, off0, out + run_count(P));
  }
}

// Any read length: one wavefront per read, blocks of 64 x 32 bytes.  A backward sweep leaves, per block, the members
// that enter it from the right (and the read's totals); the forward sweep then has both sides and writes the records
// of each block (pass 2 looks records up by block, so a long read is never scanned whole).
__global__ __launch_bounds__(256) void k_mhl_rows_multi(RowsArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n) return;
  const int64_t rs = a.off[row], re = rs + a.len[row];
  const int64_t c0 = rs >> 5;
  const int64_t c1 = re > rs ? (re + 31) >> 5 : c0;
  const int64_t nblk = (c1 - c0 + 63) >> 6;
  const int64_t bi = (rs >> MHL_BLK_SHIFT) + 2 * row;        // this read's slots in blkrec / cont (>= nblk of them)
  uint32_t h = 0, oo_m = 0, oo_u = 0, anyk = 0;
  uint32_t from_right = 0;
  auto load_blk = [&](int64_t b) {                           // this lane's 32 bytes of block b (none outside [0, nblk))
    const int64_t ci = c0 + b * 64 + lane;
    return mhl_chunk_load<2>(a.xm, ci << 5, b >= 0 && b < nblk && ci < c1, rs, re);
  };
  ChunkRaw<2> raw = load_blk(nblk - 1);
  for (int64_t b = nblk - 1; b >= 0; b--) {
    const ChunkRaw<2> nxt = load_blk(b - 1);                 // in flight during this block's scans
    const Chunk<uint32_t> c = mhl_chunk_masks<2>(raw, a.lut);
    raw = nxt;
    h += __popc(c.U | c.L); oo_m += c.oom; oo_u += c.oou; anyk |= c.K ? 1u : 0u;
    Seg sf = {c.L ? 1u : 0u, lead_members(c)};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      Seg r;
      r.has = __shfl_down(sf.has, d, 64); r.cnt = __shfl_down(sf.cnt, d, 64);
      if (lane + d < 64) sf = seg_combine(r, sf);
    }
    if (lane == 0) __hip_atomic_store(a.cont + bi + b, from_right, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // to L2, read back below
    const uint32_t bh = __shfl(sf.has, 0, 64), bc = __shfl(sf.cnt, 0, 64);
    from_right = bh ? bc : bc + from_right;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    h += __shfl_xor(h, d, 64);
    oo_m += __shfl_xor(oo_m, d, 64);
    oo_u += __shfl_xor(oo_u, d, 64);
    anyk |= __shfl_xor(anyk, d, 64);
  }
  const bool keep = mhl_keep(h, oo_m, oo_u, a.hmin, a.max_oo);
  if (lane == 0) a.rowinfo[row] = make_int4(keep ? (int32_t)h : -1, (int32_t)anyk, 0, 0);
  if (keep && lane == 0 && h > __hip_atomic_load(a.max_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.max_h, h);
  if (!keep) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    // the stores above have left the wave (no cache maintenance needed:
                                                             // the loads below are agent-scope, i.e. served by L2)
  Seg carry = {0u, 0u};
  raw = load_blk(0);
  for (int64_t b = 0; b < nblk; b++) {
    const int64_t cidx = c0 + b * 64 + lane;
    const ChunkRaw<2> nxt = load_blk(b + 1);
    const Chunk<uint32_t> c = mhl_chunk_masks<2>(raw, a.lut);
    raw = nxt;
    Seg pf = {c.L ? 1u : 0u, trail_members(c)}, sf = {c.L ? 1u : 0u, lead_members(c)};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      Seg l, r;
      l.has = __shfl_up(pf.has, d, 64); l.cnt = __shfl_up(pf.cnt, d, 64);
      if (lane >= d) pf = seg_combine(l, pf);
      r.has = __shfl_down(sf.has, d, 64); r.cnt = __shfl_down(sf.cnt, d, 64);
      if (lane + d < 64) sf = seg_combine(r, sf);
    }
    Seg ex, exr;
    ex.has = __shfl_up(pf.has, 1, 64); ex.cnt = __shfl_up(pf.cnt, 1, 64);
    if (lane == 0) { ex.has = 0u; ex.cnt = 0u; }
    exr.has = __shfl_down(sf.has, 1, 64); exr.cnt = __shfl_down(sf.cnt, 1, 64);
    if (lane == 63) { exr.has = 0u; exr.cnt = 0u; }
    const uint32_t enter = seg_combine(carry, ex).cnt;
    Seg last;
    last.has = __shfl(pf.has, 63, 64); last.cnt = __shfl(pf.cnt, 63, 64);
    carry = seg_combine(carry, last);
    const uint32_t right = __hip_atomic_load(a.cont + bi + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t cont = exr.has ? exr.cnt : exr.cnt + right;
    const uint32_t P = span_bits<32>(c, enter, cont);
    const uint32_t Q = anyk ? (c.V & ~c.K) : 0u;
    const uint32_t nrec = run_count(P) + run_count(Q);
    uint32_t inc = nrec;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    const uint32_t total = __shfl(inc, 63, 64);
    const uint32_t region = blockIdx.x & (MHL_REGIONS - 1), region_cap = a.rec_cap / MHL_REGIONS;
    unsigned long long base = 0;
    if (lane == 0 && total) base = atomicAdd(a.rec_cursor + region * MHL_CUR_STRIDE, (unsigned long long)total);
    base = ((unsigned long long)__shfl((uint32_t)(base >> 32), 0, 64) << 32) | __shfl((uint32_t)base, 0, 64);
    const bool fits = base + total <= (unsigned long long)region_cap;
    base += (unsigned long long)region * region_cap;
    if (lane == 0) a.blkrec[bi + b] = make_uint2((uint32_t)base, fits ? total : 0u);
    if (nrec && fits) {
      const uint32_t off0 = (uint32_t)((cidx << 5) - rs);
      MhlRec *out = a.recs + (uint32_t)base + inc - nrec;
      write_runs<32>(P, true, c, enter, cont, off0, out);
      write_runs<32>(Q, false, c, enter, cont, off0, out + run_count(P));
    }
  }
}

// largest region usage -> *out (the host compares it with the region capacity)
__global__ void k_mhl_cursor_max(const unsigned long long *cur, unsigned long long *out) {
  unsigned long long v = cur[threadIdx.x * MHL_CUR_STRIDE];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(v >> 32), d, 64) << 32) | __shfl_xor((uint32_t)v, d, 64);
    v = o > v ? o : v;
  }
  if (threadIdx.x == 0) *out = v;
}

struct MhlArgs {
  RowCols c;                              // pass is always null here (no lower-casing in lMHL)
  const int4 *rowinfo;                    // see RowsArgs
  const uint2 *blkrec;
  const MhlRec *recs;
  uint32_t rec_cap;
  int multi;                              // records are kept per 2 KiB block of a row (k_mhl_rows_multi)
  const Tile *tiles;
  uint32_t ctx_mask, H;
  uint32_t *pool_key, *pool_cov;
  unsigned long long *pool_hs, *pool_nu, *pool_de;   // sum h, sum S(M), sum S(h) of the row's (pos,strand)
  uint32_t pool_cap;
  uint32_t slot_rows, ovf_base;          // a pool slot per tile, larger tiles behind them through the cursor (as in the CX report)
  uint32_t *cursor, *tile_nrow, *tile_base;
  // ultra-deep tiles are set aside and split over many workgroups (as in the CX kernel)
  int heavy_rows, heavy_chunk;
  uint32_t *heavy_count, *heavy_max, *heavy_list;
  uint32_t *heavy_cnt;                    // [heavy tile][16][T] counters
  unsigned long long *heavy_sums;         // [heavy tile][MHL_NSUM] difference arrays
  // tiles shared with other ranks of a sharded run: same two slabs, indexed by shared slot
  uint32_t *shared_cnt;
  unsigned long long *shared_sums;
#ifdef EPI_MHL_CHECK
  uint32_t *dbg;                          // diagnostic build: first out-of-range index {code, v0, v1, block, thread}
  int64_t n, nbytes, nblkrec;
#endif
};

#ifdef EPI_MHL_CHECK
#define MHL_CHECK(cond, code, v0, v1)                                                                               \
  if (!(cond)) {                                                                                                    \
    if (atomicCAS(a.dbg, 0u, (uint32_t)(code)) == 0u) {                                                             \
      a.dbg[1] = (uint32_t)(v0); a.dbg[2] = (uint32_t)(v1); a.dbg[3] = blockIdx.x; a.dbg[4] = threadIdx.x;          \
    }                                                                                                               \
    return;                                                                                                         \
  }
#else
#define MHL_CHECK(cond, code, v0, v1)
#endif

// (stray nibbles 3 / 4 / 8, whose counter slot IS the numerator / denominator / haplotype-size sum in the reference,
// :190, are flagged by bits 6-7 of the packed counter LUT: tile_common.hpp)

// ST = type of the LDS difference arrays: u64, or u32 when no tile (or heavy-tile chunk) of the batch can reach 2^31 in
// any of the three sums -- decided on the host from the largest haplotype size pass 1 saw; u32 entries wrap like signed
// numbers and are sign-extended where they leave LDS for the u64 slabs.
template <class ST> struct MhlLds {
  uint32_t *cnt;                          // [2][4][T] packed code counters (as the CX kernel)
  ST *sums;                               // [3][2][MHL_SLEN] padded difference arrays of sum S(M) (:193), sum h (:192), sum S(h) (:194)
};
__device__ __forceinline__ unsigned long long mhl_widen(unsigned long long v) { return v; }
__device__ __forceinline__ unsigned long long mhl_widen(uint32_t v) { return (unsigned long long)(long long)(int32_t)v; }
constexpr int MHL_DN = 0, MHL_DH = 2 * MHL_SLEN, MHL_DD = 4 * MHL_SLEN;

struct MhlSlice {
  RowSlice rs;
  int pos0;                               // tile position of byte 0 of this lane's first dword (may be -1..-3)
  int pf, pe;                             // tile positions [pf, pe) the slice covers
  int sidx;                               // 0 '+', 1 '-'
  uint32_t hs;                            // haplotype size of the read | bit 31: the read has skipped bytes (its counted
                                          // runs come as records)
  int32_t rel;                            // tile position of the read's byte 0 = start - pos0
  int32_t blk0, blk1;                     // multi: record blocks of the read that can reach into the tile
  uint32_t rb, rn;                        // else: the read's records
};

struct MhlRow {                           // what a lane prefetches of a candidate row
  RowVals v;
  int32_t hrow, skips;
  uint32_t rb, rn;
};

__device__ __forceinline__ MhlRow mhl_load_row(const MhlArgs &a, const Tile &td, int r) {
  MhlRow m;
  m.v = cx_load_row(a.c, td, r);
  m.hrow = -1; m.skips = 0; m.rb = 0; m.rn = 0;
  if (m.v.ok) {
    const int4 ri = a.rowinfo[r];
    m.hrow = ri.x; m.skips = ri.y; m.rb = (uint32_t)ri.z; m.rn = (uint32_t)ri.w;
    if (m.hrow < 0) m.v.ok = false;                       // read dropped by pass 1 (:179)
  }
  return m;
}

template <int G>
__device__ __forceinline__ MhlSlice mhl_slice_of(const MhlArgs &a, const MhlRow &row, const Tile &td, int sub, uint32_t *cnt) {
  MhlSlice m;
  m.rs = cx_slice_of<MHL_T, G, true>(a.c, row.v, td, sub, cnt);
  m.pos0 = 0; m.pf = 0; m.pe = 0; m.sidx = 0; m.hs = 0; m.rel = 0; m.blk0 = 0; m.blk1 = -1; m.rb = 0; m.rn = 0;
  if (m.rs.nd > 0) {
    m.hs = (uint32_t)row.hrow | (row.skips ? 0x80000000u : 0u);
    m.rb = row.rb; m.rn = (uint64_t)row.rb + row.rn <= a.rec_cap ? row.rn : 0u;   // pass 1 out of record space: the caller reruns
    const int32_t rel = (int32_t)((uint32_t)td.pos0 - (uint32_t)row.v.st);
    const int32_t lo = rel > 0 ? rel : 0;
    const int32_t hi = row.v.len < rel + MHL_T ? row.v.len : rel + MHL_T;
    const int64_t b0 = row.v.o + lo;
    const int32_t e_lo = (int32_t)b0 & 3;
    m.pf = lo - rel;
    m.pe = hi - rel;
    m.pos0 = m.pf - e_lo + 4 * sub;
    m.sidx = row.v.sd - 1;
    m.rel = -rel;
    if (a.multi) {
      const int64_t c0 = row.v.o >> 5;
      m.blk0 = (int32_t)((((row.v.o + lo) >> 5) - c0) >> 6);
      m.blk1 = (int32_t)((((row.v.o + hi - 1) >> 5) - c0) >> 6);
    }
  }
  return m;
}

// One dword of a row: the packed CX counters; stray nibbles 3/4/8 additionally bump one of the three sums at their
// position (+1 there, -1 after it, in the difference arrays).
template <int OFF, bool FIRST, class ST>
__device__ __forceinline__ void mhl_add_dword(uint32_t w, int k, const MhlSlice &m, const MhlLds<ST> &L) {
  // the packed counter LUT also flags the stray nibbles (bits 6-7 of its bytes: 1 = nibble 3, 2 = nibble 4, 3 = nibble 8)
  const uint32_t f4 = cx_add_dword<MHL_T, OFF, FIRST, true>(w, k == m.rs.nd - 1, m.rs);
  if (f4 == 0u) return;                                  // common case: nothing but the counters
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t fl = (f4 >> (8 * j + 6)) & 3u;
    if (!fl) continue;
    const int p = m.pos0 + OFF + j;
    ST *d = L.sums + (fl == 1u ? MHL_DN : fl == 2u ? MHL_DD : MHL_DH) + m.sidx * MHL_SLEN;
    atomicAdd(d + mhl_pad(p), (ST)1);
    atomicAdd(d + mhl_pad(p + 1), (ST)0 - (ST)1);
  }
}

template <int G, int U0, int U1, class ST>
__device__ __forceinline__ void mhl_add_range(const uint32_t (&w)[MHL_NU], int sub, const MhlSlice &cur, const MhlLds<ST> &L) {
  if constexpr (U0 < U1) {
    if (sub + U0 * G < cur.rs.nd) mhl_add_dword<4 * G * U0, U0 == 0>(w[U0], sub + U0 * G, cur, L);
    mhl_add_range<G, U0 + 1, U1>(w, sub, cur, L);
  }
}

// +v on tile positions [a, b) of one difference array
template <class ST>
__device__ __forceinline__ void mhl_interval(ST *d, int64_t a, int64_t b, unsigned long long v) {
  if (a < 0) a = 0;
  if (b > MHL_T) b = MHL_T;
  if (a < b) { atomicAdd(d + mhl_pad((int)a), (ST)v); atomicAdd(d + mhl_pad((int)b), (ST)0 - (ST)v); }
}

template <int G, int WG, class ST>
__device__ __forceinline__ void mhl_accumulate(const MhlArgs &a, const Tile &td, const MhlLds<ST> &L) {
  constexpr int R = 64 / G;
  constexpr int NW = WG / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1), grp = lane / G;
  int r = td.row_lo + wave * R + grp;
  MHL_CHECK(td.row_lo >= 0 && td.row_hi <= a.n && td.row_hi >= td.row_lo, 9, td.row_lo, td.row_hi)
  MhlRow row = mhl_load_row(a, td, r);
  for (int rbase = td.row_lo + wave * R; rbase < td.row_hi; rbase += NW * R) {
    const MhlSlice cur = mhl_slice_of<G>(a, row, td, sub, L.cnt);
    if (cur.rs.nd > 0) {
      MHL_CHECK(r >= 0 && r < a.n, 1, r, cur.rs.nd)
      MHL_CHECK(cur.sidx == 0 || cur.sidx == 1, 2, cur.sidx, r)
      MHL_CHECK(cur.pf >= 0 && cur.pf < MHL_T && cur.pe > cur.pf && cur.pe <= MHL_T, 3, cur.pf, cur.pe)
      MHL_CHECK(cur.rs.nd <= (MHL_T + 6) / 4 + 1, 4, cur.rs.nd, r)
      MHL_CHECK(reinterpret_cast<const uint8_t *>(cur.rs.src) >= a.c.xm &&
                reinterpret_cast<const uint8_t *>(cur.rs.src) - a.c.xm + 4 * (int64_t)(cur.rs.nd - sub) <= a.nbytes + 64, 5,
                reinterpret_cast<const uint8_t *>(cur.rs.src) - a.c.xm, cur.rs.nd)
#ifdef EPI_MHL_CHECK
      if (sub < cur.rs.nd) {
        const long rel0 = (cur.rs.dst[0] - L.cnt) - cur.sidx * 4 * MHL_T;      // first cell this lane adds to, within its strand's plane 0
        MHL_CHECK(rel0 >= -3 && rel0 + 4 * ((cur.rs.nd - 1 - sub) / G) * G < MHL_T + 3, 7, rel0, cur.rs.nd * 1000 + sub)
      }
#endif
    }
    uint32_t w[MHL_NU];
#pragma unroll
    for (int u = 0; u < MHL_NU; u++) w[u] = sub + u * G < cur.rs.nd ? cur.rs.src[u * G] : 0u;
    const int rcur = r;
    const int64_t ocur = row.v.o;

    r += NW * R;
    row = mhl_load_row(a, td, r);                          // the next step's columns are in flight during this step's adds
    if (cur.rs.nd > 0) {
      const uint32_t h = cur.hs & 0x7FFFFFFFu;
      const unsigned long long sh = mhl_lut(h, a.H);       // S(h), :194
      ST *dn = L.sums + MHL_DN + cur.sidx * MHL_SLEN;
      ST *dh = L.sums + MHL_DH + cur.sidx * MHL_SLEN;
      ST *dd = L.sums + MHL_DD + cur.sidx * MHL_SLEN;
      if (!(cur.hs >> 31) && sub == 0) {   // every byte of the slice is counted: one interval per sum
        mhl_interval(dh, cur.pf, cur.pe, (unsigned long long)h);
        mhl_interval(dd, cur.pf, cur.pe, sh);
      }
      auto add_rec = [&](const MhlRec &rec) {                 // a stretch piece or a counted run that may reach into the tile
        const int64_t ta = (int64_t)cur.rel + rec.first, tb = (int64_t)cur.rel + rec.last + 1;
        if (rec.m) mhl_interval(dn, ta, tb, mhl_lut(rec.m, a.H));
        else { mhl_interval(dh, ta, tb, (unsigned long long)h); mhl_interval(dd, ta, tb, sh); }
      };
      {
        for (uint32_t k = (uint32_t)sub; k < cur.rn; k += G) add_rec(a.recs[cur.rb + k]);
        if (a.multi) {                                         // long reads: records are kept per 2 KiB block
          const int64_t bi = (ocur >> MHL_BLK_SHIFT) + 2 * (int64_t)rcur;
          for (int32_t blk = cur.blk0; blk <= cur.blk1; blk++) {
            MHL_CHECK(bi + blk >= 0 && bi + blk < a.nblkrec, 8, bi + blk, blk)
            const uint2 br = a.blkrec[bi + blk];
            if ((uint64_t)br.x + br.y > a.rec_cap) continue;   // pass 1 ran out of record space: the caller reruns
            for (uint32_t k = (uint32_t)sub; k < br.y; k += G) add_rec(a.recs[br.x + k]);
          }
        }
      }
    }
    mhl_add_range<G, 0, MHL_NU>(w, sub, cur, L);
    for (int k = sub + MHL_NU * G; k < cur.rs.nd; k += G) {
      MhlSlice t = cur;
#pragma unroll
      for (int j = 0; j < 4; j++) t.rs.dst[j] = cur.rs.dst[j] + 4 * (k - sub);
      t.pos0 = cur.pos0 + 4 * (k - sub);
      mhl_add_dword<0, false>(cur.rs.src[k - sub], k, t, L);
    }
  }
}

// Rule, prefix sums of the difference arrays, ordered compaction of one tile (one position per thread).  The pool rows
// carry the three integer sums; the two divisions (:92-93) are done by k_mhl_gather, one row per lane.
// inclusive prefix sum over the 64 lanes with DPP moves (row_shr 1, 2, 4, 8 inside a row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3): no LDS traffic, unlike __shfl_up (ds_bpermute)
template <int WG, bool PK, class ST>
__device__ __forceinline__ void mhl_emit(const MhlArgs &a, int tile, const MhlLds<ST> &L, uint32_t *s_scan) {
  constexpr int T = MHL_T;
  constexpr int NW = WG / 64;
  constexpr int PER = T / 64;                             // positions per lane in the prefix-sum phase
  constexpr int PPT = T / WG;                             // consecutive positions per thread in the emit phase
  constexpr int NS = 2 * PPT;                             // (pos, strand) cells per thread, in key order
  static_assert(PPT == 1 || PPT == 2, "emit phase layout");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // in-place inclusive prefix sums of the six difference arrays: a wavefront takes arrays wave, wave + NW, ... (serial
  // over a lane's PER consecutive positions, one 64-lane scan of the lane totals)
  for (int ai = wave; ai < 6; ai += NW) {
    static_assert(PER == 8, "the padding of the difference arrays assumes 8 entries per lane");
    ST *arr = L.sums + ai * MHL_SLEN + lane * (PER + 1);
    ST x[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) x[j] = arr[j];
#pragma unroll
    for (int j = 1; j < PER; j++) x[j] += x[j - 1];
    const ST inc = mhl_wave_scan<ST>(x[PER - 1]);
    const ST ex = inc - x[PER - 1];
#pragma unroll
    for (int j = 0; j < PER; j++) arr[j] = x[j] + ex;
  }
  __syncthreads();
  uint32_t key[NS], cov[NS];
  unsigned long long hs[NS], nu[NS], de[NS];
  bool ok[NS];
  int nr = 0;
#pragma unroll
  for (int i = 0; i < NS; i++) {
    const int p = (int)threadIdx.x * PPT + (i >> 1), s = i & 1;
    uint32_t c[8];
    if constexpr (PK) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t w = L.cnt[(s * 4 + k) * T + p];
        c[2 * k] = w & 0xFFFFu;
        c[2 * k + 1] = w >> 16;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = L.cnt[(s * 8 + k) * T + p];
    }
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
    ok[i] = k != 0;
    key[i] = ((uint32_t)p << 4) | ((uint32_t)s << 3) | (uint32_t)k;
    cov[i] = cc;                                                             // :90
    hs[i] = L.sums[MHL_DH + s * MHL_SLEN + mhl_pad(p)];                      // :92 numerator
    nu[i] = L.sums[MHL_DN + s * MHL_SLEN + mhl_pad(p)];                      // :93 numerator
    de[i] = L.sums[MHL_DD + s * MHL_SLEN + mhl_pad(p)];                      // :93 denominator
    nr += k != 0;
  }
  // rows of the lower lanes from one ballot per cell (a lane has 0..NS rows), no shuffle scan
  uint32_t inc = (uint32_t)nr, wtot = 0;
#pragma unroll
  for (int i = 0; i < NS; i++) {
    const unsigned long long bi = __ballot(ok[i]);
    inc += __builtin_amdgcn_mbcnt_hi((uint32_t)(bi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bi, 0u));
    wtot += (uint32_t)__popcll(bi);
  }
  if (lane == 0) s_scan[wave] = wtot;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int w = 0; w < NW; w++) { const uint32_t t = s_scan[w]; s_scan[w] = acc; acc += t; }
    s_scan[NW] = acc;
    uint32_t base = 0;
    bool fits = true;
    if (acc) {
      if (acc <= a.slot_rows) base = (uint32_t)tile * a.slot_rows;        // no atomic: see cx_pool_reserve (cx_report.hip)
      else {
        const uint32_t o = atomicAdd(a.cursor, acc);
        fits = (uint64_t)a.ovf_base + o + acc <= a.pool_cap;
        base = a.ovf_base + o;
      }
    }
    s_scan[NW + 1] = base;
    s_scan[NW] = fits ? acc : 0xFFFFFFFFu;
    a.tile_nrow[tile] = acc;
    a.tile_base[tile] = base;
  }
  __syncthreads();
  const uint32_t total = s_scan[NW], base = s_scan[NW + 1];
  if (total != 0xFFFFFFFFu) {
    uint32_t w = base + inc - (uint32_t)nr + s_scan[wave];
#pragma unroll
    for (int i = 0; i < NS; i++) {
      if (ok[i]) {
        a.pool_key[w] = key[i];
        a.pool_cov[w] = cov[i];
        a.pool_hs[w] = hs[i];
        a.pool_nu[w] = nu[i];
        a.pool_de[w] = de[i];
        w++;
      }
    }
  }
}

constexpr int MHL_LDS_CNT = cx_lds_dwords<MHL_T, true>() + 2 * kCxGuard;

template <class ST>
__device__ __forceinline__ MhlLds<ST> mhl_lds(uint32_t *cnt, ST *sums) {
  MhlLds<ST> L;
  L.cnt = cnt;
  L.sums = sums;
  return L;
}

// three workgroups per CU (6 waves per SIMD, 80 VGPRs) for both sum types: with u32 sums (30 KiB of LDS) a fourth would
// fit, but at 64 VGPRs the kernel spills 33 of them and runs 26 ms instead of 13.7 on config 4 (u64: 17.5)
template <int WG, class ST> constexpr int mhl_waves_per_simd() {
  if (WG == 256) return sizeof(ST) == 4 ? 5 : 3;          // as many 256-thread workgroups as LDS holds: 5 x 30 KiB / 3 x 44 KiB
  return sizeof(ST) == 4 ? EPI_MHL_WPS : 6;
}

// adds a tile's LDS difference arrays to its u64 slab in HBM
template <int WG, class ST>
__device__ __forceinline__ void mhl_dump_sums(const ST *sums, unsigned long long *ds) {
  for (int i = threadIdx.x; i < MHL_NSUM; i += WG) { const ST v = sums[i]; if (v) atomicAdd(ds + i, mhl_widen(v)); }
}

template <int G, int WG, class ST>
__global__ __launch_bounds__(WG, (mhl_waves_per_simd<WG, ST>())) void k_mhl_tiles(MhlArgs a, int ntiles) {
  constexpr int T = MHL_T;
  constexpr int NW = WG / 64;
  __shared__ __attribute__((aligned(16))) uint32_t cnt_raw[MHL_LDS_CNT];
  __shared__ __attribute__((aligned(16))) ST sums[MHL_NSUM];
  __shared__ uint32_t s_scan[NW + 2];
  const MhlLds<ST> L = mhl_lds(cnt_raw + kCxGuard, sums);
  const int chunk = (ntiles + 7) >> 3;                   // XCD-aware tile order, as the CX kernel
  const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const Tile td = a.tiles[tile];                         // (in flight while LDS is cleared, 16 bytes per store)
  static_assert(MHL_LDS_CNT % 4 == 0 && (MHL_NSUM * sizeof(ST)) % 16 == 0, "LDS is cleared in 16-byte pieces");
  {
    uint4 *z = reinterpret_cast<uint4 *>(cnt_raw);
    for (int i = threadIdx.x; i < MHL_LDS_CNT / 4; i += WG) z[i] = make_uint4(0, 0, 0, 0);
    uint4 *zs = reinterpret_cast<uint4 *>(sums);
    for (int i = threadIdx.x; i < (int)(MHL_NSUM * sizeof(ST) / 16); i += WG) zs[i] = make_uint4(0, 0, 0, 0);
  }
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
  __syncthreads();
  mhl_accumulate<G, WG>(a, td, L);
  __syncthreads();
  if (td.slot >= 0) {                                    // shared with another rank: hand the raw sums over
    cx_dump_slab<T, WG, true>(L.cnt, reinterpret_cast<int32_t *>(a.shared_cnt + (int64_t)td.slot * (16 * T)));
    mhl_dump_sums<WG>(sums, a.shared_sums + (int64_t)td.slot * MHL_NSUM);
    if (threadIdx.x == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
    return;
  }
  mhl_emit<WG, true>(a, tile, L, s_scan);
}

// One chunk of the candidate rows of one heavy tile -> added into that tile's slab in HBM.
template <int G, int WG, class ST>
__global__ __launch_bounds__(WG, (mhl_waves_per_simd<WG, ST>())) void k_mhl_heavy(MhlArgs a) {
  constexpr int T = MHL_T;
  __shared__ __attribute__((aligned(16))) uint32_t cnt_raw[MHL_LDS_CNT];
  __shared__ __attribute__((aligned(16))) ST sums[MHL_NSUM];
  const MhlLds<ST> L = mhl_lds(cnt_raw + kCxGuard, sums);
  const int tile = (int)a.heavy_list[blockIdx.y];
  Tile td = a.tiles[tile];
  const int lo = td.row_lo + (int)blockIdx.x * a.heavy_chunk;
  if (lo >= td.row_hi) return;
  td.row_lo = lo;
  if (td.row_hi - lo > a.heavy_chunk) td.row_hi = lo + a.heavy_chunk;
  for (int i = threadIdx.x; i < MHL_LDS_CNT; i += WG) cnt_raw[i] = 0;
  for (int i = threadIdx.x; i < MHL_NSUM; i += WG) sums[i] = (ST)0;
  __syncthreads();
  mhl_accumulate<G, WG>(a, td, L);
  __syncthreads();
  uint32_t *dc = td.slot >= 0 ? a.shared_cnt + (int64_t)td.slot * (16 * T) : a.heavy_cnt + (int64_t)blockIdx.y * (16 * T);
  unsigned long long *ds = td.slot >= 0 ? a.shared_sums + (int64_t)td.slot * MHL_NSUM : a.heavy_sums + (int64_t)blockIdx.y * MHL_NSUM;
  cx_dump_slab<T, WG, true>(L.cnt, reinterpret_cast<int32_t *>(dc));
  mhl_dump_sums<WG>(sums, ds);
}

// Rule + rows of one tile whose sums sit in HBM slabs (u32 counters [16][T] + MHL_NSUM u64): heavy and shared tiles.
template <int WG>
__device__ __forceinline__ void mhl_emit_from_slab(const MhlArgs &a, int tile, const uint32_t *sc, const unsigned long long *ss) {
  constexpr int T = MHL_T;
  constexpr int NW = WG / 64;
  __shared__ __attribute__((aligned(16))) uint32_t cnt[16 * T];
  __shared__ __attribute__((aligned(16))) unsigned long long sums[MHL_NSUM];   // the slabs are always u64
  __shared__ uint32_t s_scan[NW + 2];
  for (int i = threadIdx.x; i < 16 * T; i += WG) cnt[i] = sc[i];
  for (int i = threadIdx.x; i < MHL_NSUM; i += WG) sums[i] = ss[i];
  __syncthreads();
  mhl_emit<WG, false>(a, tile, mhl_lds(cnt, sums), s_scan);
}

template <int WG>
__global__ __launch_bounds__(WG) void k_mhl_emit_heavy(MhlArgs a) {
  const int tile = (int)a.heavy_list[blockIdx.x];
  if (a.tiles[tile].slot >= 0) return;                   // emitted after the cross-rank reduce
  mhl_emit_from_slab<WG>(a, tile, a.heavy_cnt + (int64_t)blockIdx.x * (16 * MHL_T), a.heavy_sums + (int64_t)blockIdx.x * MHL_NSUM);
}

// Emits the shared tiles this rank owns from the (already cross-rank reduced) slabs: one workgroup per slot.
template <int WG>
__global__ __launch_bounds__(WG) void k_mhl_emit_slab(MhlArgs a, const int32_t *__restrict__ owned,
                                                       const int32_t *__restrict__ slot_tile) {
  if (!owned[blockIdx.x]) return;
  const int tile = slot_tile[blockIdx.x];
  if (tile < 0) return;
  mhl_emit_from_slab<WG>(a, tile, a.shared_cnt + (int64_t)blockIdx.x * (16 * MHL_T), a.shared_sums + (int64_t)blockIdx.x * MHL_NSUM);
}

// one wavefront per tile: pool rows -> their place in the final table (see k_cx_gather)
__global__ __launch_bounds__(256) void k_mhl_gather(const Tile *__restrict__ tiles, const uint32_t *__restrict__ tile_out,
                                                     const uint32_t *__restrict__ tile_nrow, const uint32_t *__restrict__ tile_base,
                                                     int32_t ntiles, const uint32_t *__restrict__ pool_key,
                                                     const uint32_t *__restrict__ pool_cov, const unsigned long long *__restrict__ pool_hs,
                                                     const unsigned long long *__restrict__ pool_nu,
                                                     const unsigned long long *__restrict__ pool_de, int32_t *__restrict__ o_rname,
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
  // two consecutive rows per lane and instruction (8- / 16-byte loads and stores at any dword / qword address), an odd
  // last row on its own
  struct __attribute__((packed, aligned(4))) U2 { uint32_t x, y; };
  struct __attribute__((packed, aligned(8))) Q2 { unsigned long long x, y; };
  struct __attribute__((packed, aligned(8))) D2 { double x, y; };
  const uint32_t n2 = n & ~1u;
  const uint32_t rn = (uint32_t)td.rname, p0 = (uint32_t)td.pos0;
  for (uint32_t i = 2u * lane; i < n2; i += 128) {
    const U2 key = *reinterpret_cast<const U2 *>(pool_key + src0 + i);
    const U2 cov = *reinterpret_cast<const U2 *>(pool_cov + src0 + i);
    const Q2 hs = *reinterpret_cast<const Q2 *>(pool_hs + src0 + i);
    const Q2 nu = *reinterpret_cast<const Q2 *>(pool_nu + src0 + i);
    const Q2 de = *reinterpret_cast<const Q2 *>(pool_de + src0 + i);
    const uint32_t o = dst0 + i;
    *reinterpret_cast<U2 *>(o_rname + o) = U2{rn, rn};
    *reinterpret_cast<U2 *>(o_strand + o) = U2{1u + ((key.x >> 3) & 1u), 1u + ((key.y >> 3) & 1u)};
    *reinterpret_cast<U2 *>(o_pos + o) = U2{p0 + (key.x >> 4), p0 + (key.y >> 4)};
    *reinterpret_cast<U2 *>(o_ctx + o) = U2{key.x & 7u, key.y & 7u};
    *reinterpret_cast<U2 *>(o_cov + o) = cov;                                          // :90
    *reinterpret_cast<D2 *>(o_len + o) = D2{(double)hs.x / (double)(int)cov.x, (double)hs.y / (double)(int)cov.y};   // :92
    *reinterpret_cast<D2 *>(o_lmhl + o) = D2{(double)nu.x / (double)de.x, (double)nu.y / (double)de.y};             // :93
  }
  if (n2 < n && lane == 0) {
    const uint32_t i = n2;
    const uint32_t key = pool_key[src0 + i];
    const uint32_t o = dst0 + i;
    o_rname[o] = td.rname;
    o_strand[o] = 1 + (int32_t)((key >> 3) & 1u);
    o_pos[o] = (int32_t)(td.pos0 + (int64_t)(key >> 4));
    o_ctx[o] = (int32_t)(key & 7u);
    const uint32_t cov = pool_cov[src0 + i];
    o_cov[o] = (int32_t)cov;                                                          // :90
    o_len[o] = (double)pool_hs[src0 + i] / (double)(int)cov;                          // :92
    o_lmhl[o] = (double)pool_nu[src0 + i] / (double)pool_de[src0 + i];                // :93
  }
}

size_t mhl_pool_rows(const epi_batch *b) { return b->pool_cap < b->pool_cap2 ? b->pool_cap : b->pool_cap2; }

int ensure_mhl_pool(epi_batch *b, size_t rows) {
  if (rows > b->pool_cap || !b->pool_key.p) {
    EPI_TRY(b->pool_key.ensure(rows * 4));
    EPI_TRY(b->pool_a.ensure(rows * 4));
    EPI_TRY(b->pool_b.ensure(rows * 4));
    b->pool_cap = rows;
  }
  if (rows > b->pool_cap2 || !b->pool_d.p) {
    EPI_TRY(b->pool_d.ensure(rows * 8));
    EPI_TRY(b->pool_e.ensure(rows * 8));
    EPI_TRY(b->pool_f.ensure(rows * 8));
    b->pool_cap2 = rows;
  }
  return EPI_OK;
}

template <int WG, class ST>
static void launch_mhl_tiles_wg(int g, int nt, hipStream_t s, const MhlArgs &a) {
  const unsigned grid = (unsigned)(((nt + 7) / 8) * 8);
  switch (g) {
    case 8: hipLaunchKernelGGL((k_mhl_tiles<8, WG, ST>), dim3(grid), dim3(WG), 0, s, a, nt); break;
    case 16: hipLaunchKernelGGL((k_mhl_tiles<16, WG, ST>), dim3(grid), dim3(WG), 0, s, a, nt); break;
    case 32: hipLaunchKernelGGL((k_mhl_tiles<32, WG, ST>), dim3(grid), dim3(WG), 0, s, a, nt); break;
    default: hipLaunchKernelGGL((k_mhl_tiles<64, WG, ST>), dim3(grid), dim3(WG), 0, s, a, nt); break;
  }
}

template <class ST>
static void launch_mhl_tiles(int g, int nt, hipStream_t s, const MhlArgs &a) {
  const int wg_env = options().mhl_wg;                     // EPIHIP_MHL_WG=256/512 forces one (tests, A/B runs)
  const int wg = wg_env ? wg_env : (a.multi ? MHL_WG : MHL_WG_SHORT);
  if (wg == 256) launch_mhl_tiles_wg<256, ST>(g, nt, s, a); else launch_mhl_tiles_wg<512, ST>(g, nt, s, a);
}

template <class ST>
static void launch_mhl_heavy(int g, uint32_t nheavy, uint32_t nchunks, hipStream_t s, const MhlArgs &a) {
  const dim3 grid(nchunks, nheavy);
  switch (g) {
    case 8: hipLaunchKernelGGL((k_mhl_heavy<8, MHL_WG, ST>), grid, dim3(MHL_WG), 0, s, a); break;
    case 16: hipLaunchKernelGGL((k_mhl_heavy<16, MHL_WG, ST>), grid, dim3(MHL_WG), 0, s, a); break;
    case 32: hipLaunchKernelGGL((k_mhl_heavy<32, MHL_WG, ST>), grid, dim3(MHL_WG), 0, s, a); break;
    default: hipLaunchKernelGGL((k_mhl_heavy<64, MHL_WG, ST>), grid, dim3(MHL_WG), 0, s, a); break;
  }
  hipLaunchKernelGGL((k_mhl_emit_heavy<MHL_WG>), dim3(nheavy), dim3(MHL_WG), 0, s, a);
}

// lanes per row in the tile kernel (as pick_cx_group)
static int pick_mhl_tile_group(int32_t max_len) {
  { const int g = options().mhl_tile_group; if (g == 8 || g == 16 || g == 32 || g == 64) return g; }   // EPIHIP_MHL_TILE_GROUP
  const int slice = (max_len < MHL_T ? max_len : MHL_T) + 3;
  const int nd = (slice + 3) / 4;
  int g = 8;
  while (g < 64 && g * MHL_NU < nd) g <<= 1;
  return g;
}

// k_mhl_rows variant for the batch: G lanes per read x 16*C bytes per lane, the smallest G*16*C that holds the longest
// read wherever it starts inside its first 16 bytes; 0 = the batch has longer reads than 64 lanes cover (or
// EPIHIP_MHL_MULTI is set): k_mhl_rows_multi.  Returned as G*8 + C.
int pick_mhl_group(int32_t max_len) {
  if (options().mhl_multi) return 0;
  if (options().mhl_group_g > 0) {                                         // EPIHIP_MHL_GROUP="G,C" for A/B runs; must cover the reads
    const int g = options().mhl_group_g, c = options().mhl_group_c;
    if ((g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64) && c >= 2 && c <= 4 &&
        (int64_t)g * 16 * c >= (int64_t)max_len + 15)
      return g * 8 + c;
  }
  int best = 0;
  int64_t best_cap = 0;
  for (int g = 2; g <= 64; g <<= 1)
    for (int c = 2; c <= 4; c++) {
      const int64_t cap = (int64_t)g * 16 * c;
      if (cap < (int64_t)max_len + 15) continue;
      if (!best || cap < best_cap) { best = g * 8 + c; best_cap = cap; }    // ties: the earlier (fewer lanes) wins
    }
  return best;
}

// nibble -> MhlLut flags for one context set (rcpp_mhl_report.cpp:104-107, :176-177, :187)
MhlLut make_mhl_lut(uint32_t ctx_mask) {
  uint32_t w[4] = {0, 0, 0, 0};
  for (uint32_t code = 0; code < 16; code++) {
    const bool in = (ctx_mask >> code) & 1u;
    uint32_t f = 0;
    if (in) f |= code < 8 ? 1u : 2u;
    if (code == 11) f |= 4u;
    if (!in && ((0x00E4u >> code) & 1u)) f |= 8u;            // codes 2,5,6,7
    if (!in && ((0xE400u >> code) & 1u)) f |= 16u;           // codes 10,13,14,15
    w[code >> 2] |= f << (8 * (code & 3));
  }
  MhlLut l;
  l.lo0 = w[0]; l.lo1 = w[1]; l.hi0 = w[2]; l.hi1 = w[3];
  return l;
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

  {                                                        // short reads, one haplotype context: one pass over the bytes
    bool done = false;
    EPI_TRY(mhl_fused_report(b, ctx_mask, H, hmin, max_ooctx_meth_frac, s, nrow_out, &done));
    if (done) return EPI_OK;
  }

  RowStats st;
  int32_t nt = 0;
  EPI_TRY(build_tiles(b, s, kMhlTile, &st, &nt));
  b->last_ntiles = nt;
  if (nt == 0) {                                           // (a rank of a sharded run without rows still takes part in the exchange)
    b->last_kind = !b->shared_keys.empty() && b->d_mhl_cnt_slab ? 4 : 2; b->last_nrow = 0;
    return EPI_OK;
  }

  // pass 1 workspace: per-read info, record table, records (grown on demand like the row pool)
  const int gc = pick_mhl_group(st.max_len);
  const bool multi = gc == 0;
  const size_t nblkrec = multi ? (size_t)(b->nbytes >> MHL_BLK_SHIFT) + 2 * (size_t)b->n + 2 : 1;
  EPI_TRY(b->mhl_h.ensure((size_t)b->n * 16));
  EPI_TRY(b->mhl_blk.ensure(nblkrec * 8));
  if (multi) EPI_TRY(b->mhl_cont.ensure(nblkrec * 4));
  if (b->mhl_rec_cap == 0) {
    b->mhl_rec_cap = (size_t)b->nbytes / 48 + (size_t)b->n + 64 * MHL_REGIONS;
    EPI_TRY(b->mhl_m.ensure(b->mhl_rec_cap * sizeof(MhlRec)));
  }
  EPI_TRY(b->mhl_cur.ensure((size_t)MHL_REGIONS * MHL_CUR_STRIDE * 8));
  unsigned long long *rec_cursor = b->mhl_cur.as<unsigned long long>();
  unsigned long long *rec_max = reinterpret_cast<unsigned long long *>(b->misc.as<uint32_t>() + 12);   // misc[12..13]

  RowsArgs ra;
  ra.xm = b->xm; ra.off = b->off; ra.len = b->len; ra.n = b->n;
  ra.lut = make_mhl_lut(ctx_mask);
  ra.hmin = (int32_t)hmin; ra.max_oo = max_ooctx_meth_frac;
  ra.rowinfo = b->mhl_h.as<int4>();
  ra.blkrec = b->mhl_blk.as<uint2>();
  ra.rec_cursor = rec_cursor;
  ra.cont = multi ? b->mhl_cont.as<uint32_t>() : nullptr;
  ra.max_h = b->misc.as<uint32_t>() + 14;                    // misc[14]

  EPI_TRY(b->tile_nrow.ensure((size_t)nt * 4));
  EPI_TRY(b->tile_base.ensure((size_t)nt * 4));
  EPI_TRY(b->tile_out.ensure((size_t)(nt + 1) * 4));
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;

  MhlArgs a;
  a.c.xm = b->xm; a.c.off = b->off; a.c.len = b->len; a.c.start = b->start; a.c.strand = b->strand; a.c.pass = nullptr;
  a.rowinfo = b->mhl_h.as<int4>();
  a.blkrec = b->mhl_blk.as<uint2>();
  a.multi = multi ? 1 : 0;
#ifdef EPI_MHL_CHECK
  EPI_TRY(b->diag.ensure(256));
  a.dbg = b->diag.as<uint32_t>();
  a.n = b->n; a.nbytes = b->nbytes; a.nblkrec = (int64_t)nblkrec;
  EPI_HIP(hipMemsetAsync(a.dbg, 0, 32, s));
#endif
  a.tiles = b->tiles.as<Tile>();
  a.ctx_mask = ctx_mask; a.H = H;
  a.cursor = cursor;
  a.tile_nrow = b->tile_nrow.as<uint32_t>();
  a.tile_base = b->tile_base.as<uint32_t>();
  a.heavy_rows = 16384;
  if (options().heavy_rows > 0) a.heavy_rows = options().heavy_rows;   // test hook (EPIHIP_HEAVY_ROWS)
  if (a.heavy_rows > 32767) a.heavy_rows = 32767;          // k_mhl_tiles' packed u16 counters: a base adds at most 2 (the
                                                           // CX kernels cap at 16384, cx_report.hip)
  const int heavy_rows_base = a.heavy_rows;
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
  // row pool = a slot per tile + an overflow region (see epi_batch_cx_report_dev): CpG haplotypes give ~7 % of the
  // (pos,strand) cells of a tile a row; the slot doubles for the next call when 1/8 of the rows outgrew it
  if (!b->mhl_slot) b->mhl_slot = MHL_T / 8;
  uint32_t slot = b->mhl_slot > 2u * MHL_T ? 2u * MHL_T : b->mhl_slot;
  if (options().mhl_slot >= 0 && options().mhl_slot <= 2 * MHL_T) slot = (uint32_t)options().mhl_slot;   // test hook (EPIHIP_MHL_SLOT)
  while (slot && (unsigned long long)nt * slot > 0xC0000000ull) slot >>= 1;   // row indices are u32
  size_t ovf_base = (size_t)nt * slot;
  for (;;) {
    const size_t ovf = (ovf_base >> 4) > 65536 ? (ovf_base >> 4) : 65536;
    if (mhl_pool_rows(b) >= ovf_base + ovf + headroom) break;
    const int rc = ensure_mhl_pool(b, ovf_base + ovf + headroom);
    if (rc == EPI_OK) break;
    b->pool_cap = 0; b->pool_cap2 = 0;                     // (a failed growth has released the old buffers)
    if (!slot) return rc;
    slot = 0;                                              // the slots do not fit in device memory: every tile through the cursor
    ovf_base = 0;
  }
  a.slot_rows = slot;
  a.ovf_base = (uint32_t)ovf_base;
  b->mhl_last_slot = slot;
  b->mhl_last_ovf = (uint32_t)ovf_base;
  b->mhl_ctx_mask = ctx_mask;
  uint32_t used_total[2] = {0, 0};
  const int tg = pick_mhl_tile_group(st.max_len);
  for (int attempt = 0; attempt < 3; attempt++) {
    // pass 1: per-read haplotype size and stretch records
    ra.recs = b->mhl_m.as<MhlRec>();
    ra.rec_cap = (uint32_t)b->mhl_rec_cap;
    EPI_HIP(hipMemsetAsync(rec_cursor, 0, (size_t)MHL_REGIONS * MHL_CUR_STRIDE * 8, s));
    EPI_HIP(hipMemsetAsync(ra.max_h, 0, 4, s));
    prof_begin("mhl_rows", s);
    if (multi) {
      hipLaunchKernelGGL(k_mhl_rows_multi, dim3((unsigned)((b->n + 3) / 4)), dim3(256), 0, s, ra);
    } else {
      const int g = gc >> 3;
      const unsigned nb = (unsigned)((b->n * g + 255) / 256);
#define EPI_LAUNCH(GG)                                                                                       \
  case GG * 8 + 2: hipLaunchKernelGGL((k_mhl_rows<GG, 2>), dim3(nb), dim3(256), 0, s, ra); break;            \
  case GG * 8 + 3: hipLaunchKernelGGL((k_mhl_rows<GG, 3>), dim3(nb), dim3(256), 0, s, ra); break;            \
  case GG * 8 + 4: hipLaunchKernelGGL((k_mhl_rows<GG, 4>), dim3(nb), dim3(256), 0, s, ra); break;
      switch (gc) {
        EPI_LAUNCH(2) EPI_LAUNCH(4) EPI_LAUNCH(8) EPI_LAUNCH(16) EPI_LAUNCH(32) EPI_LAUNCH(64)
        default: return fail(EPI_ERR_ARG, "bad group size");
      }
#undef EPI_LAUNCH
    }
    prof_end("mhl_rows", s);
    hipLaunchKernelGGL(k_mhl_cursor_max, dim3(1), dim3(MHL_REGIONS), 0, s, rec_cursor, rec_max);
    EPI_HIP(hipGetLastError());
    uint32_t p1[3];                                        // {fullest record region (u64), largest haplotype size}
    EPI_TRY(read_scalars(b, s, rec_max, 12, p1));
    const unsigned long long rec_used = ((unsigned long long)p1[1] << 32) | p1[0];
    if (rec_used > b->mhl_rec_cap / MHL_REGIONS) {         // record space ran out: the need is known now, redo pass 1
      if (attempt == 2) return fail(EPI_ERR_STATE, "stretch record overflow after regrow");
      const unsigned long long want = (rec_used + rec_used / 16 + 64) * MHL_REGIONS;
      if (want > 0xFFFFFFF0ull) return fail(EPI_ERR_NOMEM, "too many methylated stretches in one batch (%llu)", want);
      b->mhl_rec_cap = (size_t)want;
      EPI_TRY(b->mhl_m.ensure(b->mhl_rec_cap * sizeof(MhlRec)));
      continue;
    }
    // u32 LDS sums if no position of a tile (or heavy-tile chunk) can reach 2^31: rows x (the largest value a read
    // can add: S(h) >= h, S(M) <= S(h) for M <= h; + 1 for a stray nibble at the position)
    uint32_t hcap = p1[2] > 65535u ? 65535u : p1[2];
    if (hcap >= H) hcap = H;
    const unsigned long long vmax = nrS(hcap) > 1 ? nrS(hcap) : 1;
    const unsigned long long narrow_rows = ((1ull << 31) - 1) / (vmax + 1);
    bool narrow = narrow_rows >= 512;
    if (options().mhl_sums) narrow = narrow && options().mhl_sums == 32;   // EPIHIP_MHL_SUMS=64 forces the wide kernel
    a.heavy_rows = heavy_rows_base;
    if (narrow && (unsigned long long)a.heavy_rows > narrow_rows) a.heavy_rows = (int)narrow_rows;
    a.heavy_chunk = a.heavy_rows / 4 > 64 ? a.heavy_rows / 4 : 64;

    // pass 2
    a.recs = ra.recs;
    a.rec_cap = ra.rec_cap;
    a.pool_key = b->pool_key.as<uint32_t>();
    a.pool_cov = b->pool_a.as<uint32_t>();
    a.pool_hs = b->pool_d.as<unsigned long long>();
    a.pool_nu = b->pool_e.as<unsigned long long>();
    a.pool_de = b->pool_f.as<unsigned long long>();
    a.pool_cap = (uint32_t)(mhl_pool_rows(b) > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : mhl_pool_rows(b));
    if (attempt > 0) {                                   // (the tile-index pass zeroed them for the first attempt)
      EPI_HIP(hipMemsetAsync(cursor, 0, 12, s));         // cursor, total, heavy count
      EPI_HIP(hipMemsetAsync(a.heavy_max, 0, 4, s));
    }
    prof_begin("mhl_tiles", s);
    if (narrow) launch_mhl_tiles<uint32_t>(tg, nt, s, a); else launch_mhl_tiles<unsigned long long>(tg, nt, s, a);
    prof_end("mhl_tiles", s);
    EPI_HIP(hipGetLastError());
    EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
#ifdef EPI_MHL_CHECK
    {
      uint32_t d[8];
      EPI_HIP(hipMemcpy(d, a.dbg, 32, hipMemcpyDeviceToHost));
      if (d[0]) return fail(EPI_ERR_STATE, "lMHL index check %u failed: v0=%d v1=%d block=%u thread=%u (n=%lld nt=%d attempt=%d)", d[0],
                            (int)d[1], (int)d[2], d[3], d[4], (long long)b->n, nt, attempt);
    }
#endif
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
      if (narrow) launch_mhl_heavy<uint32_t>(tg, nheavy, nchunks, s, a); else launch_mhl_heavy<unsigned long long>(tg, nheavy, nchunks, s, a);
      prof_end("mhl_heavy", s);
      EPI_HIP(hipGetLastError());
      EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
      EPI_TRY(read_scalars(b, s, cursor, 8, host));
    }
    used_total[0] = host[0];
    used_total[1] = host[1];
    if (ovf_base + used_total[0] + headroom <= a.pool_cap) break;
    if (attempt == 2) return fail(EPI_ERR_STATE, "row pool overflow after regrow");
    EPI_TRY(ensure_mhl_pool(b, ovf_base + used_total[0] + (used_total[0] >> 4) + 1024 + headroom));
    if (nshared > 0) {   // the rerun adds into the shared slabs again
      EPI_HIP(hipMemsetAsync(a.shared_cnt, 0, (size_t)nshared * 16 * MHL_T * 4, s));
      EPI_HIP(hipMemsetAsync(a.shared_sums, 0, (size_t)nshared * MHL_NSUM * 8, s));
    }
  }
  if (used_total[0] > used_total[1] / 8 && b->mhl_slot < 2u * MHL_T) b->mhl_slot *= 2;   // too many tiles outgrew their slot
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
  b->mhl_shared_fused = false;
  return EPI_OK;
}

int epi_mhl_fused_tile_positions(void) { return MHLF_T; }

int epi_batch_mhl_fused_ok(epi_batch *b, const char *ctx, void *stream, int32_t *ok_out) {
  if (!b || !ctx || !ok_out) return fail(EPI_ERR_ARG, "epi_batch_mhl_fused_ok: NULL argument");
  *ok_out = 0;
  EPI_HIP(hipSetDevice(b->eng->device));
  uint32_t ctx_mask = 0;
  for (const unsigned char *c = reinterpret_cast<const unsigned char *>(ctx); *c; c++) ctx_mask |= 1u << ctx_to_idx(*c);
  if (b->n == 0) { RowStats st; memset(&st, 0, sizeof(st)); *ok_out = mhl_fused_eligible(b, ctx_mask, st) ? 1 : 0; return EPI_OK; }
  EPI_TRY(fetch_row_stats(b, pick_stream(b, stream)));
  *ok_out = (!b->h_stats.bad_len && mhl_fused_eligible(b, ctx_mask, b->h_stats)) ? 1 : 0;
  return EPI_OK;
}

int epi_batch_mhl_set_shared_fused(epi_batch *b, const int64_t *h_keys, const int32_t *h_owned, int32_t nshared,
                                   int32_t *d_cnt_slab, int64_t *d_sum_slab) {
  EPI_TRY(epi_batch_mhl_set_shared(b, h_keys, h_owned, nshared, d_cnt_slab, d_sum_slab));
  b->mhl_shared_fused = nshared > 0;
  return EPI_OK;
}

// Second half of a sharded lMHL report: the slabs have been sum-reduced across ranks.
int epi_batch_mhl_finish_shared(epi_batch *b, void *stream, int64_t *nrow_out) {
  if (!b || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_mhl_finish_shared: NULL argument");
  if (b->last_kind != 4 && b->last_kind != 5) return fail(EPI_ERR_STATE, "epi_batch_mhl_finish_shared without a sharded epi_batch_mhl_report_dev");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  if (b->last_kind == 5) return mhl_fused_finish_shared(b, s, nrow_out);   // the one-pass kernel's slabs
  const int32_t nt = b->last_ntiles;
  if (nt == 0) { b->last_kind = 2; b->last_nrow = 0; *nrow_out = 0; return EPI_OK; }   // this rank holds no rows: owns no tile
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
  a.pool_hs = b->pool_d.as<unsigned long long>();
  a.pool_nu = b->pool_e.as<unsigned long long>();
  a.pool_de = b->pool_f.as<unsigned long long>();
  a.pool_cap = (uint32_t)(mhl_pool_rows(b) > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : mhl_pool_rows(b));
  a.slot_rows = b->mhl_last_slot;
  a.ovf_base = b->mhl_last_ovf;
  hipLaunchKernelGGL((k_mhl_emit_slab<MHL_WG>), dim3((unsigned)b->shared_keys.size()), dim3(MHL_WG), 0, s, a,
                     b->d_shared_owned.as<int32_t>(), b->d_slot_tile.as<int32_t>());
  EPI_HIP(hipGetLastError());
  EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
  uint32_t ut[2] = {0, 0};
  EPI_TRY(read_scalars(b, s, cursor, 8, ut));
  if ((size_t)a.ovf_base + ut[0] > a.pool_cap) return fail(EPI_ERR_STATE, "row pool overflow in sharded lMHL report");
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
  prof_begin("gather", s);
  hipLaunchKernelGGL(k_mhl_gather, dim3(nb), dim3(256), 0, s, b->tiles.as<Tile>(), b->tile_out.as<uint32_t>(),
                     b->tile_nrow.as<uint32_t>(), b->tile_base.as<uint32_t>(), b->last_ntiles, b->pool_key.as<uint32_t>(),
                     b->pool_a.as<uint32_t>(), b->pool_d.as<unsigned long long>(), b->pool_e.as<unsigned long long>(),
                     b->pool_f.as<unsigned long long>(), d_icols[0], d_icols[1],
                     d_icols[2], d_icols[3], d_icols[4], d_dcols[0], d_dcols[1]);
  prof_end("gather", s);
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
  CopyPart parts[7];
  for (int i = 0; i < 5; i++) parts[i] = {h_icols[i], ic[i], (size_t)nrow * 4};
  for (int i = 0; i < 2; i++) parts[5 + i] = {h_dcols[i], dc[i], (size_t)nrow * 8};
  return copy_parts_to_host(b->eng, parts, 7, s);
}

}  // extern "C"
