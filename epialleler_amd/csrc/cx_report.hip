// rcpp_cx_report (src/rcpp_cx_report.cpp:34-159) on the GPU, optionally with rcpp_threshold_reads
// (src/rcpp_threshold_reads.cpp:15-74) fused in, i.e. all of generateCytosineReport() in ONE pass over the packed bytes.
//
// The reference walks the sorted reads and, per base, emplaces pos -> int[32] into an ordered map, flushing the map
// through the majority-context rule (spit_results, :58-85).  For sorted input the flush timing does not change the
// result: the output is the per-(rname,pos,strand) counter table pushed through the rule, in (rname, pos, '+' before
// '-') order.  A row (pos, strand, k) is emitted iff context k is reported and n_k = M_k + m_k > cov/2 (integer half,
// strict): the counters of one position are disjoint and sum to at most cov, so n_k > cov/2 already excludes '.', and
// every other context, from winning (:64-71).  Per (pos,strand) the table therefore needs only
//   * (n, M) of each REPORTED context (one for a CG report, three for CX): calls of the context in either case, and
//     the methylated ones of reads that passed (a failed read is lower-cased, :118,122); unmeth = n - M, and
//   * cov = bases of rows covering the position, minus skipped codes ('+'/'-'/filler, :123), plus nibble 9 once more
//     (it IS the reference's coverage slot, :126-127).
//
// A workgroup owns one tile of T positions on an absolute grid (tiles.hip; T = 2048 for single-context reports,
// 1024 otherwise); its candidate rows are a contiguous range.
//  * Loads are POSITION-aligned: lane `sub` of the G lanes of a row loads 16-byte chunks that cover tile positions
//    16c..16c+15 (global_load_dwordx4 at any byte alignment; the hardware allows it), so the four bases of a dword
//    land in ONE LDS cell.
//  * Counters are u8, four positions per dword, (n | M) of a context side by side in one u64: ONE ds_add_u64 per
//    dword of xm and reported context (issued only by lanes whose dword holds a call) instead of one LDS atomic per
//    base.  LEAN kernel (batches in which no position is covered by more than 255 rows, RowStats::deep == 0 -- any WGS
//    data): the u8 counters cannot overflow, the emit reads them directly; 24 KiB of LDS, 256-thread workgroups, six
//    per CU.  General kernel (pile-ups): the u8 counters are folded into u16 pairs every 255 rows (a row adds at most
//    1 per position and counter); 40 KiB, 512-thread workgroups.
//  * Lane shape: G lanes x NU chunks of 16 bytes per row, the smallest G * NU that holds the longest row
//    (pick_cx_shape; PE150: 4 x 5, 16 rows per wavefront step).  The kernel is bound by integer VALU issue (plain
//    2-operand ops 2 cycles per wavefront on gfx950, perm / compare / 3-operand / DPP / 64-bit ops 4), so idle chunk
//    slots and per-step set-up are what the shape minimises.
//  * Coverage is a difference array (+1 at the first, -1 behind the last position of a row: two LDS atomics per row,
//    both strands packed into one dword); skipped / doubled codes go to a second u8 array that is folded into it.
//  * Fused thresholding (FUSED = true; the default generateCytosineReport call: report.context == threshold.context,
//    one context): the G lanes load the WHOLE row (reads of up to ~5 kb), ONE v_perm LUT lookup per dword yields the
//    four thresholding classes (2-bit fields: in context, methylated, out-of-context methylated / unmethylated) and
//    the skip / double flags; field-wise adds, v_sad_u8 and DPP group sums give the read's totals, a table filled with
//    the reference's own IEEE divisions (:43-70) turns them into pass / fail, and the same LUT bytes then give the
//    calls.  Rows that reach into two tiles are decided twice (same result); the xm bytes come from HBM once (the
//    second visit is an L2 hit thanks to the XCD-aware tile order).
// After a barrier the workgroup prefix-sums the coverage array, lists the cells with any call (ballot ranks), applies
// the rule and writes (key, meth, unmeth) in position order into the tile's slot of the row pool; k_cx_gather places
// the pool rows in the final table.  Ultra-deep tiles are split over many workgroups through a slab in HBM; tiles
// shared with other ranks of a sharded run hand over the same slab for the RCCL all-reduce.
#include "common.hpp"
#include "tile_common.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace epi {

#ifndef EPI_CX_WG                             // (EPI_CX_WG, EPI_CX_T1: timing builds vary them; the product uses these values)
#define EPI_CX_WG 512
#endif
#ifndef EPI_CX_T1
#define EPI_CX_T1 2048
#endif
constexpr int CX_WG = EPI_CX_WG;              // threads per tile workgroup (8 wavefronts)
constexpr int CX_T1 = EPI_CX_T1;              // tile positions of single-context reports
constexpr int CX_CH = 16;                     // bytes (= positions) of one position-aligned load
constexpr int CX_FLUSH_ROWS = 255;            // u8 -> u16 fold interval: a row adds at most 1 per position and counter
constexpr int CX_SLAB_COV = 12;               // slab planes [16][T]: 2*(strand*NP + p) + {0: n, 1: M}; 12, 13: coverage
                                              // difference array of '+', '-'; 14, 15 unused

struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };   // 16 bytes at any alignment: global_load_dwordx4

// Timing builds only (`make timing ABLATE=n` -> libepihip_t<n>.so, wrong results by design; the product library is
// always built with 0): 1 aligned instead of position-aligned loads, 2 no call atomics, 4 no thresholding phase,
// 8 loads only, 16 no emit, 32 no coverage atomics
#ifndef EPI_CX_ABLATE
#define EPI_CX_ABLATE 0
#endif

struct Cx2Args {
  RowCols c;                              // c.pass: external pass vector or null (all TRUE); ignored when fused
  int64_t xm_cap;                         // readable bytes behind c.xm
  const Tile *tiles;
  uint32_t fill4;                         // 4 x a code without any flag in either LUT (stands in for bytes outside a row)
  ClassLut lut_r;                         // report LUT: byte = [n0 M0 n1 M1 n2 M2 skip dbl] flags of a code (n: a call of
                                          // plane p's context in either case, M: a methylated one)
  ClassLut lut_rx, lut_sx;                // ... and both in the two-lookup form the kernels read (lut16_xor_form)
  ClassLut lut_s;                         // fused: bits 0,2,4,6 = in context / methylated / out-of-context methylated /
                                          // unmethylated; 1,3 = skipped / doubled as is; 5,7 = the same when lower-cased
  ThrParams thr;
  const uint32_t *thr_tab;                // fused thresholding without divisions: [n] = least passing n_m for n_m + n_u = n
                                          // (low half), largest passing o_m for o_m + o_u = n (high half); k_thr_table
  uint32_t ctx_of_plane;                  // byte p = context code (2, 6, 7) of plane p
  int32_t *pass_out;                      // fused: pass flag of every row (may be null)
  uint32_t *pool_key, *pool_meth, *pool_unmeth;
  uint32_t pool_cap;
  uint32_t *cursor;                       // rows handed out of the overflow region so far (may exceed its size)
  uint32_t slot_rows, ovf_base;           // tile t owns pool rows [t * slot_rows, +slot_rows); larger tiles take
                                          // rows from [ovf_base, pool_cap) through the cursor
  uint32_t *tile_nrow, *tile_base;
  int32_t *slab;                          // shared-tile slabs [slot][16][T]
  // ultra-deep tiles (amplicon pile-ups) are set aside by k_cx_tiles and split over many workgroups
  int heavy_rows;                         // a tile with more candidate rows than this is "heavy"
  int heavy_chunk;                        // rows per work item of k_cx_heavy
  int heavy_first;                        // the heavy kernels work on heavy_list[heavy_first + blockIdx.y]; entries at or behind
                                          // *heavy_count do not exist (fixed-grid launches queued before the count is known)
  uint32_t *heavy_count, *heavy_max;      // number of heavy tiles, largest candidate-row count among them
  uint32_t *heavy_list;                   // their tile ids (order of discovery)
  int32_t *heavy_slab;                    // [heavy tile][16][T] summed over the work items
  // tiles in which a position may be covered by more than 255 rows (u8 counters): the LEAN kernel lists them and the
  // general kernel -- launched right behind it with a fixed grid, no host round trip -- works through the list
  uint32_t *deep_count, *deep_list;       // LEAN: where to list them (null: RowStats says there are none in this batch)
  const uint32_t *tile_list, *tile_list_count;   // general kernel behind a lean one: the list to work through
  uint32_t *dbg;                          // check build only (EPI_CHECK): first index violation; null in the product
  int64_t nrows;                          // rows of the batch (check build)
  int walk;                               // walking lean kernel: consecutive tiles per workgroup
};

// LEAN: no position of the batch is covered by more than 255 rows (RowStats::deep == 0, tiles.hip), so the u8 counters
// cannot overflow however many rows a tile has: no u16 copy, no folds, the emit reads the u8 counters.
// PAD > 0 (a WALKING workgroup, lean kernel only): the arrays hold W = T + PAD positions -- the tile and the reach of its
// own rows into the next tile.  The workgroup works through consecutive tiles; after a tile's emit the PAD positions
// behind it are shifted to the front and become the next tile's opening state, so that a row is analysed ONCE, by the
// tile it starts in (without it 15 % of the PE150 row visits of a 2048-position tile are second visits of rows that
// reach into the next tile: each a full load + thresholding decision).
template <int T, int NP, bool LEAN = false, int PAD = 0> struct Cx2Lds {
  static_assert(PAD == 0 || (LEAN && PAD % 16 == 0 && PAD <= T), "the window is a lean-kernel feature");
  static constexpr int TILE = T, W = T + PAD;    // positions of the tile / held in LDS
  static constexpr int Q = W / 4;
  static constexpr int N_NARROW = 2 * NP * Q;    // u64: [strand][plane][Q], low dword = n of 4 positions (u8), high = M
  static constexpr int N_CORR = 2 * Q;           // u64: [strand][Q], low dword = skipped, high = doubled (u8 x 4)
  static constexpr int N_WIDE = LEAN ? 16 : 2 * NP * T;   // u32: [strand][plane][T] = n | M << 16 (LEAN: scan scratch only)
  static constexpr int N_COV = W;                // u32: coverage difference array, '+' in the low half, '-' in the high half
                                                 // (a change "behind the window" is simply not recorded)
  unsigned long long *narrow, *corr;
  uint32_t *wide, *cov;
  uint32_t tid;                                  // threadIdx.x (a walking workgroup reads it anew for every tile, see k_cx_tiles)
};

// 16-entry byte LUT lookup of the four codes of a dword in two v_perm_b32 (X: the table in lut16_xor_form, common.hpp);
// low8 = 0 or, for a lower-cased read, 0x08080808 (every byte takes the codes-8..15 half: c | 8, rcpp_cx_report.cpp:118,122)
__device__ __forceinline__ uint32_t cx2_lut(uint32_t w, const ClassLut &X, uint32_t low8) {
  const uint32_t sel = (w & 0x0F0F0F0Fu) | low8;
  return __builtin_amdgcn_perm(X.lo1, X.lo0, sel) ^ __builtin_amdgcn_perm(X.hi1, X.hi0, sel ^ 0x08080808u);
}

template <int G>
__device__ __forceinline__ uint32_t cx2_group_sum(uint32_t v) {     // over the G lanes of a row, every lane gets it
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);                  // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);                  // quad_perm [2,3,0,1]
  if (G >= 8) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);     // row_half_mirror
  if (G >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);    // row_mirror
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// rcpp_threshold_reads.cpp:43-70 on the four class counts of a read
__device__ __forceinline__ int cx2_threshold(uint32_t n_m, uint32_t n_u, uint32_t o_m, uint32_t o_u, const ThrParams &prm) {
  int res = 0;
  if (n_m != 0) {                                           // :43
    const unsigned n_all = n_m + n_u;
    if (!(n_all < prm.min_n_ctx)) {                         // :50
      const double frac = (double)n_m / (double)n_all;      // :52
      if (!(frac < prm.min_ctx_meth_frac)) {
        res = 1;
        if (o_m > 0) {                                      // :59
          const unsigned o_all = o_m + o_u;
          const double ofrac = (double)o_m / (double)o_all; // :66
          if (ofrac > prm.max_ooctx_meth_frac) res = 0;
        }
      }
    }
  }
  return res;
}

// The same decision from a table that k_thr_table fills by evaluating exactly the expressions above (IEEE
// divisions, once per possible n instead of twice per read and tile visit): both comparisons are monotone in the
// numerator, so per denominator n there is a least passing n_m and a largest passing o_m.
__device__ __forceinline__ int cx2_threshold_tab(uint32_t n_m, uint32_t n_u, uint32_t o_m, uint32_t o_u, const ThrParams &prm,
                                                 const uint32_t *__restrict__ tab) {
  const uint32_t n_all = n_m + n_u, o_all = o_m + o_u;
  const uint32_t least = tab[n_all] & 0xFFFFu, most = tab[o_all] >> 16;
  return n_m != 0u && !(n_all < prm.min_n_ctx) && n_m >= least && (o_m == 0u || o_m <= most);
}

__global__ __launch_bounds__(256) void k_thr_table(ThrParams prm, int32_t nmax, uint32_t *__restrict__ tab) {
  const int32_t n = (int32_t)(blockIdx.x * 256 + threadIdx.x);
  if (n > nmax) return;
  // least m in [1, n] with !((double)m / n < min_frac) (0xFFFF: none); the predicate is false..false true..true
  uint32_t least = 0xFFFFu, most = 0u;
  if (n >= 1) {
    int32_t lo = 1, hi = n + 1;                              // first true in [lo, hi)
    while (lo < hi) {
      const int32_t m = (lo + hi) >> 1;
      const double frac = (double)(uint32_t)m / (double)(uint32_t)n;
      if (!(frac < prm.min_ctx_meth_frac)) hi = m; else lo = m + 1;
    }
    if (lo <= n) least = (uint32_t)lo;
    // largest m in [1, n] with !((double)m / n > max_oo) (0: none); true..true false..false
    lo = 1; hi = n + 1;                                      // first false in [lo, hi)
    while (lo < hi) {
      const int32_t m = (lo + hi) >> 1;
      const double frac = (double)(uint32_t)m / (double)(uint32_t)n;
      if (frac > prm.max_ooctx_meth_frac) hi = m; else lo = m + 1;
    }
    most = (uint32_t)(lo - 1);
  }
  tab[n] = least | (most << 16);
}

// One visit of up to G * NU position-aligned 16-byte chunks of a row, starting at chunk cs: loads (all in flight
// before the first is used), then -- FUSED -- the thresholding decision, then the calls.  `fetch_next` runs between
// the loads and their first use (the caller fetches the next row's columns there).
struct Cx2Row {
  const uint8_t *base;                    // address of tile position 0 in the row's byte string (may lie outside the row)
  int32_t rel, len;                       // tile position of byte 0 (-Lmax < rel < T); bytes
  int32_t c0, clast;                      // first / last position-aligned chunk of the row (chunk c = positions 16c..16c+15)
  bool edge;                              // a chunk of the row reaches outside the buffer (first / last rows of a batch only)
  int sidx, ps;
};

// Byte masks (0xFF per byte) of the four dwords of a 16-byte chunk: bytes [lo, 16) resp. [0, hi) -- a 128-bit shift
// done as one 64-bit shift and two selects per half (the generic per-dword form cost ~35 VALU per mask)
__device__ __forceinline__ void cx2_mask_from(int lo /* 0..15 */, uint32_t (&m)[4]) {
  const unsigned long long x = ~0ull << ((8 * lo) & 63);
  const bool low = lo < 8;
  const unsigned long long a = low ? x : 0ull, b = low ? ~0ull : x;
  m[0] = (uint32_t)a; m[1] = (uint32_t)(a >> 32); m[2] = (uint32_t)b; m[3] = (uint32_t)(b >> 32);
}
__device__ __forceinline__ void cx2_mask_upto(int hi /* 1..16 */, uint32_t (&m)[4]) {
  const int sh = 8 * (16 - hi);                                     // 0..120
  const unsigned long long x = ~0ull >> (sh & 63);
  const bool high = sh < 64;
  const unsigned long long a = high ? ~0ull : x, b = high ? x : 0ull;
  m[0] = (uint32_t)a; m[1] = (uint32_t)(a >> 32); m[2] = (uint32_t)b; m[3] = (uint32_t)(b >> 32);
}

// MODE (fused only): 0 = the whole row is this one visit; a row of more chunks than the lane shape holds is worked on in
// slices, twice: 1 = add the slice's class totals to `carry` (nothing else), then -- the decision made by the caller --
// 2 = the calls of the slice with g.ps given.
template <int T, int G, int NU, int NP, bool FUSED, int MODE = 0, class LT, class F>
__device__ __forceinline__ void cx2_visit(const Cx2Args &a, Cx2Row &g, int32_t cs, int32_t cz, int sub, int rcur,
                                          const LT &L, F fetch_next, uint32_t *carry = nullptr) {
  constexpr int C = LT::W / CX_CH, Q = LT::Q;                       // chunks / u64 cells per strand of the window (= the tile unless the workgroup walks)
  const int32_t cb = cs + sub;                                      // this lane's chunks: cb + u*G
  const int32_t tl = cz - cb;                                       // chunk u is part of the visit iff u*G <= tl
  uint32_t w[NU][4];
  if (__builtin_expect(g.edge, 0)) {
#pragma unroll
    for (int u = 0; u < NU; u++) {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        uint32_t x = a.fill4;
        if (u * G <= tl) {
          const int64_t ad = (g.base - a.c.xm) + CX_CH * (int64_t)(cb + u * G) + 4 * d;
          x = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int64_t q = ad + j;
            const uint32_t byte = (q >= 0 && q < a.xm_cap) ? (uint32_t)a.c.xm[q] : (a.fill4 & 255u);
            x |= byte << (8 * j);
          }
        }
        w[u][d] = x;
      }
    }
  } else {
    const uint8_t *p = g.base + CX_CH * (int64_t)cb;
    {                                                     // (check build) every chunk this lane loads lies inside the buffer
      const int32_t uq = tl / G < NU - 1 ? tl / G : NU - 1;   // the lane's last chunk of THIS visit (a longer row goes on in the next slice)
      const int64_t a_lo = p - a.c.xm, a_hi = a_lo + (tl >= 0 ? CX_CH * (int64_t)(uq * G) + CX_CH : 0);
      (void)a_lo; (void)a_hi;
      if (!EPI_DEV_CHECK(a.dbg, tl < 0 || (a_lo >= 0 && a_hi <= a.xm_cap), 21, a_lo, a_hi)) return;
    }
    if (EPI_CX_ABLATE & 1) p = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)15);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      if (u * G <= tl) {
        const U4u v = *reinterpret_cast<const U4u *>(p + CX_CH * u * G);
        w[u][0] = v.x; w[u][1] = v.y; w[u][2] = v.z; w[u][3] = v.w;
      } else {
        w[u][0] = w[u][1] = w[u][2] = w[u][3] = a.fill4;
      }
    }
  }
  fetch_next();
  if (EPI_CX_ABLATE & 8) {
    uint32_t x = 0;
#pragma unroll
    for (int u = 0; u < NU; u++) x ^= w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
    if (x == 0x12345678u) atomicAdd(L.cov, 1u);
    return;
  }
  // Bytes of the row's first / last chunk that belong to neighbouring rows -> a code without flags.  The first chunk
  // is the first lane's first one; the last one can be any of the lane's chunks.
  {
    uint32_t mf[4], ml[4];
    cx2_mask_from(cb == g.c0 ? (g.rel & 15) : 0, mf);
    cx2_mask_upto(((g.rel + g.len - 1) & 15) + 1, ml);
#pragma unroll
    for (int d = 0; d < 4; d++) { asm("" : "+v"(mf[d])); asm("" : "+v"(ml[d])); }    // (once per visit, not per use)
#pragma unroll
    for (int d = 0; d < 4; d++) w[0][d] = (w[0][d] & mf[d]) | (a.fill4 & ~mf[d]);
    // rows of similar length have their last chunk at the same u: the other chunks of the wavefront skip the masking
    // (a wave-uniform branch; the integer VALU these kernels are made of is what bounds them)
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const bool last = cb + u * G == g.clast;
      if (__ballot(last) != 0ull) {
        const uint32_t keep = last ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t m = ml[d] | keep;
          w[u][d] = (w[u][d] & m) | (a.fill4 & ~m);
        }
      }
    }
  }

  unsigned long long *nar = L.narrow + g.sidx * NP * Q + 4 * cb;
  unsigned long long *cor = L.corr + g.sidx * Q + 4 * cb;
  if constexpr (FUSED) {
    static_assert(NP == 1, "fused thresholding: one reported context, the thresholding context");
    // One LUT lookup per dword, kept in place of the bytes.  Class totals of the whole row: 2-bit fields, three dwords
    // add field-wise (<= 3), split into even / odd 4-bit fields (<= 12 over three chunks), summed by v_sad_u8, two
    // counts per word over the group.
    uint32_t cls[4] = {0, 0, 0, 0}, fl = 0;
#pragma unroll
    for (int u = 0; u < NU; u++) {
#pragma unroll
      for (int d = 0; d < 4; d++) w[u][d] = (EPI_CX_ABLATE & 4) ? w[u][d] : cx2_lut(w[u][d], a.lut_sx, 0u);
      fl |= w[u][0] | w[u][1] | w[u][2] | w[u][3];
    }
    // The odd bits of a LUT byte are the flags of the skipped / doubled codes; they have to be masked off before three
    // dwords are added field-wise -- unless no byte of the wavefront's rows carries one (the usual case: WGS reads whose
    // mates meet), which the OR of the lookups tells.
    auto count = [&](auto masked) {
      constexpr bool MASKED = decltype(masked)::value;
      constexpr uint32_t EV = MASKED ? 0x55555555u : 0xFFFFFFFFu;
      uint32_t E = 0, O = 0;
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const uint32_t t = (w[u][0] & EV) + (w[u][1] & EV) + (w[u][2] & EV);
        const uint32_t d3 = w[u][3] & EV;
        E += (t & 0x33333333u) + (d3 & 0x33333333u);
        O += ((t >> 2) & 0x33333333u) + ((d3 >> 2) & 0x33333333u);
        if (u % 3 == 2 || u == NU - 1) {
          cls[0] = __builtin_amdgcn_sad_u8(E & 0x0F0F0F0Fu, 0u, cls[0]);         // in context (either case)
          cls[2] = __builtin_amdgcn_sad_u8((E >> 4) & 0x0F0F0F0Fu, 0u, cls[2]);  // out of context, methylated
          cls[1] = __builtin_amdgcn_sad_u8(O & 0x0F0F0F0Fu, 0u, cls[1]);         // in context, methylated
          cls[3] = __builtin_amdgcn_sad_u8((O >> 4) & 0x0F0F0F0Fu, 0u, cls[3]);  // out of context, unmethylated
          E = 0; O = 0;
        }
      }
    };
    if constexpr (MODE != 2) {
      if (__builtin_expect(__ballot((fl & 0xAAAAAAAAu) != 0u) == 0ull, 1)) count(std::false_type{});
      else count(std::true_type{});
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) carry[k] += cls[k];
      return;
    }
    if constexpr (MODE == 0) {
      const uint32_t s01 = cx2_group_sum<G>(cls[0] | (cls[1] << 16)), s23 = cx2_group_sum<G>(cls[2] | (cls[3] << 16));
      const uint32_t n_all = s01 & 0xFFFFu, n_m = s01 >> 16;
      g.ps = a.thr_tab ? cx2_threshold_tab(n_m, n_all - n_m, s23 & 0xFFFFu, s23 >> 16, a.thr, a.thr_tab)
                       : cx2_threshold(n_m, n_all - n_m, s23 & 0xFFFFu, s23 >> 16, a.thr);
      if (a.pass_out && sub == 0 && (uint32_t)g.rel < (uint32_t)T) a.pass_out[rcur] = g.ps;   // by the tile the row starts in
    }
    // calls: n = in-context bytes, M = methylated ones of a passing read (a failed read is lower-cased, :118)
    const uint32_t pm = g.ps ? 0x01010101u : 0u;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const bool inside = (uint32_t)(cb + u * G) < (uint32_t)C;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const uint32_t lo = w[u][d] & 0x01010101u, hi = (w[u][d] >> 2) & pm;
        if (lo != 0u && inside && !(EPI_CX_ABLATE & 2)) atomicAdd(nar + 4 * u * G + d, (unsigned long long)lo | ((unsigned long long)hi << 32));
      }
    }
    // skipped / doubled codes (rare in WGS reads, common where mates do not meet): flags 1,3 as is, 5,7 lower-cased
    const uint32_t fsel = g.ps ? 0x0A0A0A0Au : 0xA0A0A0A0u;
    if (__builtin_expect((fl & fsel) != 0u, 0)) {
      const int sh = g.ps ? 1 : 5;
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const bool inside = (uint32_t)(cb + u * G) < (uint32_t)C;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t sk = (w[u][d] >> sh) & 0x01010101u, db = (w[u][d] >> (sh + 2)) & 0x01010101u;
          if ((sk | db) != 0u && inside) atomicAdd(cor + 4 * u * G + d, (unsigned long long)sk | ((unsigned long long)db << 32));
        }
      }
    }
  } else if constexpr (NP == 1 && G >= 16) {
    // One reported context, pass flags given, reads of more than ~600 bytes (16 and more lanes per row; measured: -5 % at
    // 1.2 and 2.4 kb, -12 % at 10 kb, but +3 ... +10 % for the 4 x 5 shape of PE150 templates, which keeps the LUT below;
    // profiles/r04_cx_experiments.txt): no byte LUT at all on the common path.  The codes of two dwords become the
    // nibbles of one word; ((pk & 7..7) ^ k..k) + 7..7 leaves bit 3 of a nibble CLEAR iff the code is the context's in either
    // case (k or k | 8), and the case bit of a code is its bit 3 -- plain two-operand VALU and three-input bit operations
    // (2 cycles per wavefront on gfx950, where v_perm_b32 / v_and_or_b32 / compares take 4), and one test per PAIR of dwords
    // instead of one per dword.  The rare codes -- skipped (11) and doubled (9), as is or after lower-casing (c | 8, :118,122:
    // then also 3 and 1) -- are a masked compare of the same word: (c & 1101b) == 1001b, resp. (c & 0101b) == 0001b; a lane
    // that holds one takes the LUT path below for its chunks.
    const bool passing = g.ps != 0;
    const uint32_t k77 = (a.ctx_of_plane & 255u) * 0x11111111u;
    const uint32_t rm = passing ? 0xDDDDDDDDu : 0x55555555u, rv = passing ? 0x99999999u : 0x11111111u;
    uint32_t clean = 0xFFFFFFFFu;                                     // bit 3 of every nibble stays set while no rare code was seen
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const bool inside = (uint32_t)(cb + u * G) < (uint32_t)C;      // (always: the visit stays inside the window)
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const uint32_t pk = (w[u][2 * e] & 0x0F0F0F0Fu) | ((w[u][2 * e + 1] << 4) & 0xF0F0F0F0u);
        const uint32_t z = ((pk & 0x77777777u) ^ k77) + 0x77777777u;
        const uint32_t np = ~z & 0x88888888u;                        // calls of the context: bit 3 of a byte = first dword, bit 7 = second
        const uint32_t t = (pk & rm) ^ rv;
        clean &= ((t & 0x77777777u) + 0x77777777u) | t;
        if (np != 0u && inside && !(EPI_CX_ABLATE & 2)) {
          const uint32_t up = passing ? np & ~pk : 0u;               // methylated calls of a passing read (a failed one is lower-cased)
          const uint32_t n0 = (np >> 3) & 0x01010101u, n1 = (np >> 7) & 0x01010101u;
          if (n0 != 0u) atomicAdd(nar + 4 * u * G + 2 * e, (unsigned long long)n0 | ((unsigned long long)((up >> 3) & 0x01010101u) << 32));
          if (n1 != 0u) atomicAdd(nar + 4 * u * G + 2 * e + 1, (unsigned long long)n1 | ((unsigned long long)((up >> 7) & 0x01010101u) << 32));
        }
      }
    }
    if (__builtin_expect((~clean & 0x88888888u) != 0u, 0)) {
      const uint32_t pick0 = passing ? 0u : 0x08080808u;
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const bool inside = (uint32_t)(cb + u * G) < (uint32_t)C;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t f = cx2_lut(w[u][d], a.lut_rx, pick0);
          const uint32_t sk = (f >> 6) & 0x01010101u, db = (f >> 7) & 0x01010101u;
          if ((sk | db) != 0u && inside) atomicAdd(cor + 4 * u * G + d, (unsigned long long)sk | ((unsigned long long)db << 32));
        }
      }
    }
  } else {
    // calls of the reported contexts: one ds_add_u64 per dword and plane, only from lanes that hold a call
    const uint32_t pick0 = g.ps == 0 ? 0x08080808u : 0u;            // failed the threshold: lower-cased (:118)
    uint32_t fl = 0;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const bool inside = (uint32_t)(cb + u * G) < (uint32_t)C;      // (always: the visit stays inside the tile)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const uint32_t f = cx2_lut(w[u][d], a.lut_rx, pick0);
        w[u][d] = f;
        fl |= f;
#pragma unroll
        for (int p = 0; p < NP; p++) {
          const uint32_t lo = (f >> (2 * p)) & 0x01010101u, hi = (f >> (2 * p + 1)) & 0x01010101u;
          if (lo != 0u && inside && !(EPI_CX_ABLATE & 2))
            atomicAdd(nar + p * Q + 4 * u * G + d, (unsigned long long)lo | ((unsigned long long)hi << 32));
        }
      }
    }
    if (__builtin_expect((fl & 0xC0C0C0C0u) != 0u, 0)) {
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const bool inside = (uint32_t)(cb + u * G) < (uint32_t)C;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t sk = (w[u][d] >> 6) & 0x01010101u, db = (w[u][d] >> 7) & 0x01010101u;
          if ((sk | db) != 0u && inside) atomicAdd(cor + 4 * u * G + d, (unsigned long long)sk | ((unsigned long long)db << 32));
        }
      }
    }
  }
}

// Rows [row_lo, row_hi) of the tile into the u8 counters and the coverage array.  G lanes own a row (64/G rows per
// wavefront step); the next step's row columns are fetched while the current row's bytes are in flight.
template <int T, int G, int NU, int NP, bool FUSED, int WG, class LT>
__device__ __forceinline__ void cx2_rows(const Cx2Args &a, const Tile &td, int row_lo, int row_hi, const LT &L) {
  constexpr int R = 64 / G, NW = WG / 64, C = LT::W / CX_CH, TW = LT::W;
  const int lane = L.tid & 63, wave = L.tid >> 6;
  const int sub = lane & (G - 1), grp = lane / G;
  Tile tb = td;
  tb.row_hi = row_hi;
  int r = row_lo + wave * R + grp;
  RowVals v = cx_load_row(a.c, tb, r);
#ifdef EPI_CX_TOUCH
  // Timing builds: one dword per 128-byte line of the row this lane group will work on TWO steps from now is requested
  // (and dropped) a step ahead of its columns: the bytes are on their way from HBM into the L2 while the step in between
  // is worked on.  Rows lie back to back in xm, so the row two steps ahead starts about as far behind the next step's row
  // as that one behind the current row.
  uint32_t touched = 0;
#endif
  for (int rbase = row_lo + wave * R; rbase < row_hi; rbase += NW * R) {
    const int rcur = r;
#ifdef EPI_CX_TOUCH
    const int64_t o_cur = v.o;
#endif
    if (!EPI_DEV_CHECK(a.dbg, !v.ok || (rcur >= 0 && rcur < a.nrows && v.len >= 0 && (v.sd == 1 || v.sd == 2 || v.len == 0)), 22, rcur, v.sd)) return;
    r += NW * R;
    RowVals nv;
    bool fetched = false;
    auto fetch_next = [&]() { if (!fetched) { nv = cx_load_row(a.c, tb, r); fetched = true; } };
    // (a candidate row -- start within the longest row's reach in front of the tile -- that ends in front of the tile has
    //  nothing for it; its pass flag is the business of the tile it starts in)
    if (v.ok && v.len > 0 && (int32_t)((uint32_t)v.st - (uint32_t)td.pos0) + v.len > 0) {
      Cx2Row g;
      g.rel = (int32_t)((uint32_t)v.st - (uint32_t)td.pos0);
      g.len = v.len;
      g.c0 = g.rel >> 4;
      g.clast = (g.rel + v.len - 1) >> 4;
      g.base = a.c.xm + (v.o - g.rel);
      // the first / last chunk may start before / end behind the buffer (first and last rows of a batch only)
      g.edge = v.o < CX_CH || v.o + v.len + CX_CH > a.xm_cap;
      g.sidx = v.sd - 1;
      g.ps = v.ps;
      if constexpr (FUSED) {
        if (__builtin_expect(__ballot(g.clast - g.c0 >= G * NU) == 0ull, 1)) {
          cx2_visit<T, G, NU, NP, true>(a, g, g.c0, g.clast, sub, rcur, L, fetch_next);   // every row of the step fits the lane shape
        } else {
          // A row of this step is longer than the shape holds (the host picks the shape for the bulk of the rows, pick_cx_shape):
          // the whole step goes slice by slice, once for the class totals and -- every row decided -- once more for the calls.
          uint32_t cls[4] = {0, 0, 0, 0};
          for (int32_t cs = g.c0; __ballot(cs <= g.clast) != 0ull; cs += G * NU)
            cx2_visit<T, G, NU, NP, true, 1>(a, g, cs, g.clast, sub, rcur, L, fetch_next, cls);
          const uint32_t s01 = cx2_group_sum<G>(cls[0] | (cls[1] << 16)), s23 = cx2_group_sum<G>(cls[2] | (cls[3] << 16));
          const uint32_t n_all = s01 & 0xFFFFu, n_m = s01 >> 16;
          g.ps = a.thr_tab ? cx2_threshold_tab(n_m, n_all - n_m, s23 & 0xFFFFu, s23 >> 16, a.thr, a.thr_tab)
                           : cx2_threshold(n_m, n_all - n_m, s23 & 0xFFFFu, s23 >> 16, a.thr);
          if (a.pass_out && sub == 0 && (uint32_t)g.rel < (uint32_t)T) a.pass_out[rcur] = g.ps;
          // (calls only exist inside the tile: slices in front of it or behind it are not loaded again)
          const int32_t ca = g.c0 > 0 ? g.c0 : 0, cz = g.clast < C - 1 ? g.clast : C - 1;
          for (int32_t cs = ca; __ballot(cs <= cz) != 0ull; cs += G * NU)
            cx2_visit<T, G, NU, NP, true, 2>(a, g, cs, cz, sub, rcur, L, fetch_next);
        }
      } else {
        const int32_t ca = g.c0 > 0 ? g.c0 : 0, cz = g.clast < C - 1 ? g.clast : C - 1;   // the slice inside the tile
        if (__builtin_expect(__ballot(cz - ca >= G * NU) == 0ull, 1)) {
          cx2_visit<T, G, NU, NP, false>(a, g, ca, cz, sub, rcur, L, fetch_next);
        } else {                                            // longer than the lane shape holds: several visits
          for (int32_t cs = ca; __ballot(cs <= cz) != 0ull; cs += G * NU) cx2_visit<T, G, NU, NP, false>(a, g, cs, cz, sub, rcur, L, fetch_next);
        }
      }
      if (sub == 0) {                                                 // coverage: +1 on the row's positions inside the tile
        const int32_t ca = g.rel > 0 ? g.rel : 0, cb = g.rel + v.len < TW ? g.rel + v.len : TW;
        const uint32_t unit = g.sidx ? 65536u : 1u;
        if (ca < cb && !(EPI_CX_ABLATE & 32)) { atomicAdd(L.cov + ca, unit); if (cb < TW) atomicAdd(L.cov + cb, 0u - unit); }
      }
    } else if (FUSED && v.ok && sub == 0 && a.pass_out && (uint32_t)((uint32_t)v.st - (uint32_t)td.pos0) < (uint32_t)T) {
      a.pass_out[rcur] = 0;             // an empty read has no call of the context: fails (rcpp_threshold_reads.cpp:43)
    }
    fetch_next();
    v = nv;
#ifdef EPI_CX_TOUCH
    asm volatile("" :: "v"(touched));                       // (the previous request has come back by now; its value is of no interest)
    {
      int64_t t = v.o + (v.o - o_cur) + 128 * sub;
      if (!v.ok || t < 0) t = 0;
      if (t > a.xm_cap - 4) t = a.xm_cap - 4;
      touched = *reinterpret_cast<const uint32_t *>(a.c.xm + (t & ~(int64_t)3));
    }
#endif
  }
#ifdef EPI_CX_TOUCH
  asm volatile("" :: "v"(touched));
#endif
}

// u8 counters -> u16 pairs, skipped / doubled codes -> coverage difference array.  Every cell has one owner thread.
template <int T, int NP, bool LEAN, int WG, int PAD = 0>
__device__ __forceinline__ void cx2_flush(const Cx2Lds<T, NP, LEAN, PAD> &L) {
  constexpr int Q = Cx2Lds<T, NP, LEAN, PAD>::Q, TW = Cx2Lds<T, NP, LEAN, PAD>::W;
  if constexpr (!LEAN) {
    for (int i = L.tid; i < 2 * NP * Q; i += WG) {
      const unsigned long long v = L.narrow[i];
      if (v == 0ull) continue;
      L.narrow[i] = 0ull;
      const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
      const int sp = i / Q, q = i - sp * Q;
      uint4 *wd = reinterpret_cast<uint4 *>(L.wide + sp * T + 4 * q);
      uint4 c = *wd;
      c.x += (lo & 255u) | ((hi & 255u) << 16);
      c.y += ((lo >> 8) & 255u) | (((hi >> 8) & 255u) << 16);
      c.z += ((lo >> 16) & 255u) | (((hi >> 16) & 255u) << 16);
      c.w += (lo >> 24) | ((hi >> 24) << 16);
      *wd = c;
    }
  }
  for (int i = L.tid; i < 2 * Q; i += WG) {
    const unsigned long long v = L.corr[i];
    if (v == 0ull) continue;
    L.corr[i] = 0ull;
    const uint32_t sk = (uint32_t)v, db = (uint32_t)(v >> 32);
    const int s = i / Q, q = i - s * Q;
    const int32_t unit = s ? 65536 : 1;
    int32_t prev = 0;                                                 // coverage change of position 4q+j: doubled - skipped
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int32_t d = (int32_t)((db >> (8 * j)) & 255u) - (int32_t)((sk >> (8 * j)) & 255u);
      if (d != prev) atomicAdd(L.cov + 4 * q + j, (uint32_t)((d - prev) * unit));
      prev = d;
    }
    if (prev != 0 && 4 * q + 4 < TW) atomicAdd(L.cov + 4 * q + 4, (uint32_t)(-prev * unit));
  }
}

// Pool rows for a tile's n output rows.  Every tile has its own slot of slot_rows rows, so the common case needs no
// atomic: one cursor for all tiles is ~10^5 atomics on one address, served one by one at ~8 ns each.  Only tiles with
// more rows than a slot go to the cursor.
__device__ __forceinline__ uint32_t cx_pool_reserve(const Cx2Args &a, int tile, uint32_t n, bool *fits) {
  if (n <= a.slot_rows) { *fits = true; return (uint32_t)tile * a.slot_rows; }
  const uint32_t o = atomicAdd(a.cursor, n);
  *fits = (uint64_t)a.ovf_base + o + n <= a.pool_cap;
  return a.ovf_base + o;
}

// Where the emit phase reads a tile's sums from: LDS (u16 pairs, packed coverage) ...
template <int T, int NP> struct CxSrcLds {
  const uint32_t *wide, *cov;             // cov already prefix-summed
  __device__ __forceinline__ uint32_t any(int sd, int pos) const {
    uint32_t x = 0;
#pragma unroll
    for (int p = 0; p < NP; p++) x |= wide[(sd * NP + p) * T + pos];
    return x;
  }
  __device__ __forceinline__ void pair(int sd, int p, int pos, uint32_t *n, uint32_t *M) const {
    const uint32_t w = wide[(sd * NP + p) * T + pos];
    *n = w & 0xFFFFu; *M = w >> 16;
  }
  __device__ __forceinline__ uint32_t coverage(int sd, int pos) const { const uint32_t v = cov[pos]; return sd ? v >> 16 : v & 0xFFFFu; }
};
// ... the u8 counters themselves (LEAN, one context) ...
template <int QS> struct CxSrcU8 {         // QS = u64 cells per strand
  const uint32_t *narrow;                 // the u64 cells as dword pairs: [2 * cell] = n, [2 * cell + 1] = M of 4 positions
  const uint32_t *cov;                    // prefix-summed
  __device__ __forceinline__ uint32_t any(int sd, int pos) const {
    return (narrow[2 * (sd * QS + (pos >> 2))] >> (8 * (pos & 3))) & 255u;
  }
  __device__ __forceinline__ void pair(int sd, int, int pos, uint32_t *n, uint32_t *M) const {
    const uint2 c = *reinterpret_cast<const uint2 *>(narrow + 2 * (sd * QS + (pos >> 2)));
    *n = (c.x >> (8 * (pos & 3))) & 255u; *M = (c.y >> (8 * (pos & 3))) & 255u;
  }
  __device__ __forceinline__ uint32_t coverage(int sd, int pos) const { const uint32_t v = cov[pos]; return sd ? v >> 16 : v & 0xFFFFu; }
};
// ... or a slab in HBM (u32 planes; coverage prefix-summed into LDS per strand)
template <int T, int NP> struct CxSrcSlab {
  const int32_t *slab;
  const uint32_t *cov;                    // LDS [2][T], prefix-summed
  __device__ __forceinline__ uint32_t any(int sd, int pos) const {
    uint32_t x = 0;
#pragma unroll
    for (int p = 0; p < NP; p++) x |= (uint32_t)slab[(2 * (sd * NP + p)) * T + pos];
    return x;
  }
  __device__ __forceinline__ void pair(int sd, int p, int pos, uint32_t *n, uint32_t *M) const {
    *n = (uint32_t)slab[(2 * (sd * NP + p)) * T + pos]; *M = (uint32_t)slab[(2 * (sd * NP + p) + 1) * T + pos];
  }
  __device__ __forceinline__ uint32_t coverage(int sd, int pos) const { return cov[sd * T + pos]; }
};

// in-place inclusive prefix sum of arr[0..T) by the whole workgroup (T / WG consecutive entries per thread)
template <int T, int WG = CX_WG>
__device__ __forceinline__ void cx2_prefix(uint32_t *arr, uint32_t *s_scan, uint32_t tid = threadIdx.x) {
  constexpr int PPT = T / WG, NW = WG / 64;
  static_assert(PPT >= 1 && PPT <= 8, "prefix layout");
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t x[PPT];
#pragma unroll
  for (int j = 0; j < PPT; j++) x[j] = arr[tid * PPT + j];
#pragma unroll
  for (int j = 1; j < PPT; j++) x[j] += x[j - 1];
  const uint32_t inc = wave_scan_u32(x[PPT - 1]);
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  uint32_t before = inc - x[PPT - 1];
#pragma unroll
  for (int w = 0; w < NW; w++) before += w < wave ? s_scan[w] : 0u;
#pragma unroll
  for (int j = 0; j < PPT; j++) arr[tid * PPT + j] = x[j] + before;
  __syncthreads();
}

// Rule + ordered compaction of one tile into the row pool.  A wavefront owns T / 8 consecutive positions and walks
// them in blocks of 32: lanes 0-31 look at the '+' strand, lanes 32-63 at the '-' strand.  Pass 1 lists, in key order,
// the cells with any call of a reported context (a row needs n_k > cov/2 >= 0); pass 2 reads the candidates densely,
// one per lane, and applies the rule.  Ranks come from ballots and popcounts.
template <int T, int NP, int WG = CX_WG, class SRC>
__device__ __forceinline__ void cx2_emit(const Cx2Args &a, int tile, const SRC &src, uint32_t *s_scan, uint16_t *s_list, uint32_t tid = threadIdx.x) {
  constexpr int NW = WG / 64, PW = T / NW, IT = PW / 32;
  static_assert(PW % 32 == 0 && IT >= 1 && IT <= 16, "emit phase layout");   // IT = 4 (T = 1024) or 8 (2048)
  const int lane = tid & 63, wave = tid >> 6;
  const int l5 = lane & 31, sd = lane >> 5;
  const uint32_t below = (1u << l5) - 1u;
  uint16_t *list = s_list + wave * (2 * PW);
  uint32_t nc = 0;                                        // candidates of this wavefront (uniform)
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const uint32_t any = src.any(sd, wave * PW + i * 32 + l5);
    const unsigned long long bal = __ballot(any != 0u);
    const uint32_t blo = (uint32_t)bal, bhi = (uint32_t)(bal >> 32);
    // key order inside a block: position first, '+' before '-': lane l precedes lane 32 + l
    const uint32_t rank = (uint32_t)__popc(blo & below) + (uint32_t)__popc(bhi & below) + (sd ? (blo >> l5) & 1u : 0u);
    if (any != 0u) list[nc + rank] = (uint16_t)(i * 64 + lane);
    nc += (uint32_t)__popcll(bal);
  }
  uint32_t key[IT], me[IT], un[IT], off[IT];              // statically indexed (fully unrolled): stay in VGPRs
  bool ok[IT];
  uint32_t carry = 0;                                     // rows of this wavefront so far (uniform)
#pragma unroll
  for (int jj = 0; jj < IT; jj++) {
    ok[jj] = false; key[jj] = 0; me[jj] = 0; un[jj] = 0; off[jj] = 0;
    if ((uint32_t)(jj * 64) < nc) {
      const uint32_t j = (uint32_t)(jj * 64 + lane);
      bool good = false;
      if (j < nc) {
        const uint32_t id = list[j];
        const int pos = wave * PW + (int)(id >> 6) * 32 + (int)(id & 31u), st = (int)((id >> 5) & 1u);
        const uint32_t half = src.coverage(st, pos) >> 1;                  // :63
        uint32_t ctx = 0, m = 0, u = 0;
#pragma unroll
        for (int p = 0; p < NP; p++) {                                     // :65-71 (at most one context can exceed half)
          uint32_t nn, M;
          src.pair(st, p, pos, &nn, &M);
          if (nn > half && ctx == 0) { ctx = (a.ctx_of_plane >> (8 * p)) & 255u; m = M; u = nn - M; }
        }
        good = ctx != 0;
        key[jj] = ((uint32_t)pos << 4) | ((uint32_t)st << 3) | ctx;
        me[jj] = m;
        un[jj] = u;
      }
      const unsigned long long be = __ballot(good);
      off[jj] = carry + __builtin_amdgcn_mbcnt_hi((uint32_t)(be >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)be, 0u));
      carry += (uint32_t)__popcll(be);
      ok[jj] = good;
    }
  }
  if (lane == 0) s_scan[wave] = carry;
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    for (int w = 0; w < NW; w++) { const uint32_t t = s_scan[w]; s_scan[w] = acc; acc += t; }
    s_scan[NW] = acc;
    uint32_t base = 0;
    bool fits = true;
    if (acc) base = cx_pool_reserve(a, tile, acc, &fits);
    s_scan[NW + 1] = base;
    s_scan[NW] = fits ? acc : 0xFFFFFFFFu;
    a.tile_nrow[tile] = acc;
    a.tile_base[tile] = base;
  }
  __syncthreads();
  const uint32_t total = s_scan[NW], base = s_scan[NW + 1];
  if (total != 0xFFFFFFFFu) {
    const uint32_t w0 = base + s_scan[wave];
#pragma unroll
    for (int jj = 0; jj < IT; jj++) {
      if (ok[jj]) {
        const uint32_t w = w0 + off[jj];
        if (!EPI_DEV_CHECK(a.dbg, w < a.pool_cap, 23, w, a.pool_cap)) continue;
        a.pool_key[w] = key[jj];
        a.pool_meth[w] = me[jj];
        a.pool_unmeth[w] = un[jj];
      }
    }
  }
}

// Adds a tile's (folded) LDS sums into its dense slab [16][T] in HBM (shared tiles, heavy tiles).  The coverage
// array goes over un-summed: difference arrays add across work items and ranks like everything else.
template <int T, int NP, bool LEAN, int WG, int PAD = 0>
__device__ __forceinline__ void cx2_dump_slab(const Cx2Lds<T, NP, LEAN, PAD> &L, int32_t *slab) {
  uint32_t *dst = reinterpret_cast<uint32_t *>(slab);
  if constexpr (LEAN) {
    constexpr int Q = Cx2Lds<T, NP, LEAN, PAD>::Q;
    for (int i = L.tid; i < 2 * NP * Q; i += WG) {
      const unsigned long long v = L.narrow[i];
      if (v == 0ull) continue;
      const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
      const int sp = i / Q, q = i - sp * Q;
      if (PAD > 0 && q >= T / 4) continue;                 // (behind the tile: the next tile's share of a walking workgroup's window)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t n = (lo >> (8 * j)) & 255u, M = (hi >> (8 * j)) & 255u;
        if (n) atomicAdd(dst + (2 * sp) * T + 4 * q + j, n);
        if (M) atomicAdd(dst + (2 * sp + 1) * T + 4 * q + j, M);
      }
    }
  } else {
    for (int i = L.tid; i < 2 * NP * T; i += WG) {
      const uint32_t v = L.wide[i];
      if (!v) continue;
      const int sp = i / T, p = i - sp * T;
      if (v & 0xFFFFu) atomicAdd(dst + (2 * sp) * T + p, v & 0xFFFFu);
      if (v >> 16) atomicAdd(dst + (2 * sp + 1) * T + p, v >> 16);
    }
  }
  for (int p = L.tid; p < T; p += WG) {
    const uint32_t v = L.cov[p];
    if (!v) continue;
    const int32_t lo = (int32_t)(int16_t)(v & 0xFFFFu);                // both halves are signed before the prefix sum
    const int32_t hi = ((int32_t)v - lo) >> 16;
    if (lo) atomicAdd(dst + CX_SLAB_COV * T + p, (uint32_t)lo);
    if (hi) atomicAdd(dst + (CX_SLAB_COV + 1) * T + p, (uint32_t)hi);
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one, each XCD has its own L2).
// Giving XCD x the contiguous tile range [x*chunk, (x+1)*chunk) makes the tiles that run together on
// an XCD genomic neighbours, so the rows two adjacent tiles both read are served from that L2.
// Bijective on [0, 8*chunk) >= ntiles; purely a speed choice (results do not depend on placement).
__device__ __forceinline__ int cx_tile_of_block(int b, int ntiles) {
  const int chunk = (ntiles + 7) >> 3;
  return (b & 7) * chunk + (b >> 3);
}

// workgroups per CU by LDS (the u8 arrays double as the emit phase's candidate lists) and the 2048-thread limit
template <int T, int NP, bool LEAN = false, int PAD = 0> constexpr int cx2_lds_bytes() {
  using LdsT = Cx2Lds<T, NP, LEAN, PAD>;
  return (LdsT::N_NARROW + LdsT::N_CORR) * 8 + (LdsT::N_WIDE + LdsT::N_COV) * 4;
}
#ifndef EPI_CX_WPS
#define EPI_CX_WPS 8
#endif
template <int T, int NP, int NU = 3, bool LEAN = false, int WG = CX_WG, int PAD = 0> constexpr int cx2_waves_per_simd() {
  const int by_lds = (160 * 1024) / (cx2_lds_bytes<T, NP, LEAN, PAD>() + 64), by_thr = 2048 / WG;
  int wgs = by_lds < by_thr ? by_lds : by_thr;
  if (NU >= 4 && wgs * WG > 1536) wgs = 1536 / WG;        // five chunks per lane in flight need ~80 VGPRs: 6 waves per SIMD
  if (wgs * WG / 256 > EPI_CX_WPS) wgs = EPI_CX_WPS * 256 / WG;
  return (wgs < 1 ? 1 : wgs) * WG / 256;
}

// LDS of a tile workgroup.  The emit phase's candidate lists (4 T bytes) and scan scratch reuse arrays that are dead by
// then: the u8 counters and `corr` (general kernel), or `corr` and the small `wide` stub (LEAN: the counters are read).
#define CX2_SHARED(T, NP, LEAN, PAD)                                                                             \
  using LdsT = Cx2Lds<T, NP, LEAN, PAD>;                                                                         \
  __shared__ __attribute__((aligned(16))) unsigned long long s_u8[LdsT::N_NARROW + LdsT::N_CORR];                \
  __shared__ __attribute__((aligned(16))) uint32_t s_wide[LdsT::N_WIDE];                                         \
  __shared__ __attribute__((aligned(16))) uint32_t s_cov[LdsT::N_COV];                                           \
  static_assert((LdsT::N_NARROW + LdsT::N_CORR) * 8 >= 4 * T + 64 && LdsT::N_CORR * 8 >= 4 * T, "the candidate lists and the scan scratch reuse dead arrays"); \
  uint32_t *s_scan = LEAN ? s_wide : reinterpret_cast<uint32_t *>(s_u8) + T;                                      \
  uint16_t *s_list = reinterpret_cast<uint16_t *>(LEAN ? s_u8 + LdsT::N_NARROW : s_u8);                          \
  (void)s_scan; (void)s_list;                                                                                     \
  LdsT L;                                                                                                         \
  L.narrow = s_u8; L.corr = s_u8 + LdsT::N_NARROW; L.wide = s_wide; L.cov = s_cov; L.tid = threadIdx.x;

template <int T, int NP, bool LEAN, int WG, int PAD = 0>
__device__ __forceinline__ void cx2_clear(const Cx2Lds<T, NP, LEAN, PAD> &L) {
  using LdsT = Cx2Lds<T, NP, LEAN, PAD>;
  uint4 *z = reinterpret_cast<uint4 *>(L.narrow);
  for (int i = L.tid; i < (LdsT::N_NARROW + LdsT::N_CORR) / 2; i += WG) z[i] = make_uint4(0, 0, 0, 0);
  if constexpr (!LEAN) {
    uint4 *y = reinterpret_cast<uint4 *>(L.wide);
    for (int i = L.tid; i < LdsT::N_WIDE / 4; i += WG) y[i] = make_uint4(0, 0, 0, 0);
  }
  uint4 *x = reinterpret_cast<uint4 *>(L.cov);
  for (int i = L.tid; i < LdsT::N_COV / 4; i += WG) x[i] = make_uint4(0, 0, 0, 0);
}

// rows [row_lo, row_hi) of a tile, folded every CX_FLUSH_ROWS rows; leaves everything in `wide` and `cov`
template <int T, int G, int NU, int NP, bool FUSED, bool LEAN, int WG, int PAD = 0>
__device__ __forceinline__ void cx2_accumulate(const Cx2Args &a, const Tile &td, int row_lo, int row_hi, const Cx2Lds<T, NP, LEAN, PAD> &L) {
  if constexpr (LEAN) {
    cx2_rows<T, G, NU, NP, FUSED, WG>(a, td, row_lo, row_hi, L);
  } else {
    for (int b0 = row_lo; b0 < row_hi; b0 += CX_FLUSH_ROWS) {
      if (b0 > row_lo) { __syncthreads(); cx2_flush<T, NP, LEAN, WG>(L); __syncthreads(); }
      cx2_rows<T, G, NU, NP, FUSED, WG>(a, td, b0, b0 + CX_FLUSH_ROWS < row_hi ? b0 + CX_FLUSH_ROWS : row_hi, L);
    }
  }
  __syncthreads();
  cx2_flush<T, NP, LEAN, WG, PAD>(L);
  __syncthreads();
}

// (the lean kernel runs 256-thread workgroups, six per CU: twice the row steps per wavefront and tile, so the fixed
//  work per tile weighs half as much, and 80 VGPRs for the five-chunk lane shapes)
#ifndef EPI_CX_LEAN_WG
#define EPI_CX_LEAN_WG 256                    // (timing builds vary it)
#endif
template <bool LEAN> constexpr int cx2_wg() { return LEAN ? EPI_CX_LEAN_WG : CX_WG; }

// A walking workgroup's step from one tile to the next: the PAD positions behind the tile become the front of the window.
// `narrow` moves as it is; `cov` -- prefix-summed in place over the tile by now -- hands on its running sum (the coverage
// entering the next tile) in the new first entry, followed by the un-summed differences behind the tile; `corr` was folded
// into `cov` by the flush and then used as the emit's candidate list: cleared.  Values travel through registers between
// two barriers (source and destination ranges of different threads overlap).
template <int T, int NP, int WG, int PAD>
__device__ __forceinline__ void cx2_shift(const Cx2Lds<T, NP, true, PAD> &L) {
  using LdsT = Cx2Lds<T, NP, true, PAD>;
  constexpr int Q = LdsT::Q, QT = T / 4, QP = PAD / 4, NN = 2 * NP * QP;      // u64 cells per strand and plane: window, tile, overhang
  constexpr int KN = (NN + WG - 1) / WG, KC = (PAD + WG - 1) / WG;
  unsigned long long nv[KN];
  uint32_t cv[KC];
#pragma unroll
  for (int k = 0; k < KN; k++) {
    const int i = (int)L.tid + k * WG;
    nv[k] = i < NN ? L.narrow[(i / QP) * Q + QT + (i % QP)] : 0ull;
  }
#pragma unroll
  for (int k = 0; k < KC; k++) {
    const int i = (int)L.tid + k * WG;
    cv[k] = i < PAD ? L.cov[T + i] + (i == 0 ? L.cov[T - 1] : 0u) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KN; k++) {
    const int i = (int)L.tid + k * WG;
    if (i < NN) L.narrow[(i / QP) * Q + (i % QP)] = nv[k];
  }
#pragma unroll
  for (int k = 0; k < KC; k++) {
    const int i = (int)L.tid + k * WG;
    if (i < PAD) L.cov[i] = cv[k];
  }
  // everything behind the carried part starts at zero (16-byte stores: QP, QT and PAD are multiples of 4)
  for (int i = L.tid; i < 2 * NP * (QT / 2); i += WG) {
    const int sp = i / (QT / 2), j = i - sp * (QT / 2);
    *reinterpret_cast<uint4 *>(L.narrow + sp * Q + QP + 2 * j) = make_uint4(0, 0, 0, 0);
  }
  for (int i = L.tid; i < T / 4; i += WG) *reinterpret_cast<uint4 *>(L.cov + PAD + 4 * i) = make_uint4(0, 0, 0, 0);
  for (int i = L.tid; i < LdsT::N_CORR / 2; i += WG) *reinterpret_cast<uint4 *>(L.corr + 2 * i) = make_uint4(0, 0, 0, 0);
}

// One tile: accumulate, then hand over (heavy / shared / deep tiles) or emit.  `fresh`: the arrays do not hold the previous
// tile's overhang (always, unless the workgroup walks): they are cleared and every candidate row of the tile is visited;
// otherwise only the rows that start in the tile, [row_from, row_hi).  Returns false when the tile was handed over without
// being accumulated (a walking workgroup then has nothing to carry on).
template <int T, int G, int NU, int NP, bool FUSED, bool LEAN, int PAD, class LdsT>
__device__ __forceinline__ bool cx2_tile(const Cx2Args &a, int tile, const Tile &td, bool fresh, int row_from, const LdsT &L, uint32_t *s_scan,
                                         uint16_t *s_list, int *s_flag) {
  constexpr int WG = cx2_wg<LEAN>();
  if (fresh) cx2_clear<T, NP, LEAN, WG, PAD>(L);
  if (td.row_hi - td.row_lo > a.heavy_rows) {
    // one workgroup would crawl through this pile-up alone: k_cx_heavy splits it by row chunks instead
    if (L.tid == 0) {
      const uint32_t h = atomicAdd(a.heavy_count, 1u);
      a.heavy_list[h] = (uint32_t)tile;
      atomicMax(a.heavy_max, (uint32_t)(td.row_hi - td.row_lo));
      a.tile_nrow[tile] = 0;
      a.tile_base[tile] = 0;
    }
    return false;
  }
  if constexpr (LEAN) {
    // u8 counters hold 255 rows per position.  Few enough candidate rows: safe.  Else the sorted starts decide: rows
    // covering a position p all start before the end of the first of them, so if row x + 255 starts at or behind the end
    // of row x for every candidate x, no position of the tile is covered by more than 255 rows (tiles.hip: k_row_stats
    // asks the same of the whole batch).  A tile that fails goes to the general kernel's list.  (The overhang a walking
    // workgroup carries into the tile comes from candidate rows of the tile as well: the criterion covers it.)
    if (a.deep_list && td.row_hi - td.row_lo > CX_FLUSH_ROWS) {
      if (L.tid == 0) *s_flag = 0;
      __syncthreads();
      bool deep = false;
      for (int x = td.row_lo + (int)L.tid; x + CX_FLUSH_ROWS < td.row_hi; x += WG)
        deep |= (int64_t)a.c.start[x + CX_FLUSH_ROWS] < (int64_t)a.c.start[x] + a.c.len[x];
      if (deep) *s_flag = 1;
      __syncthreads();
      if (*s_flag) {
        if (L.tid == 0) { a.deep_list[atomicAdd(a.deep_count, 1u)] = (uint32_t)tile; a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
        return false;
      }
    }
  }
  __syncthreads();
  cx2_accumulate<T, G, NU, NP, FUSED, LEAN, WG, PAD>(a, td, row_from, td.row_hi, L);
  if (td.slot >= 0) {
    // shared with another rank: hand the raw sums over (a carried-in coverage is part of the first difference)
    cx2_dump_slab<T, NP, LEAN, WG, PAD>(L, a.slab + (int64_t)td.slot * (kCxPlanes * T));
    if (L.tid == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
    if constexpr (PAD > 0) { __syncthreads(); cx2_prefix<T, WG>(L.cov, s_scan, L.tid); }    // (the walk goes on from the summed coverage)
    return true;
  }
  if (EPI_CX_ABLATE & 16) { if (L.tid == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; } return true; }
  // (a list-free emit -- every thread ruling on its own 8 cells, two barriers instead of four -- measured 1-2 % slower)
  cx2_prefix<T, WG>(L.cov, s_scan, L.tid);
  if constexpr (LEAN) {
    static_assert(NP == 1, "the u8 counters serve single-context reports");
    CxSrcU8<LdsT::Q> src;
    src.narrow = reinterpret_cast<const uint32_t *>(L.narrow); src.cov = L.cov;
    cx2_emit<T, NP, WG>(a, tile, src, s_scan, s_list, L.tid);
  } else {
    CxSrcLds<T, NP> src;
    src.wide = L.wide; src.cov = L.cov;
    cx2_emit<T, NP, WG>(a, tile, src, s_scan, s_list, L.tid);
  }
  return true;
}

template <int T, int G, int NU, int NP, bool FUSED, bool LEAN, int PAD = 0>
__global__ __launch_bounds__(cx2_wg<LEAN>(), (cx2_waves_per_simd<T, NP, NU, LEAN, cx2_wg<LEAN>(), PAD>())) void k_cx_tiles(Cx2Args a, int ntiles) {
  CX2_SHARED(T, NP, LEAN, PAD)
  __shared__ int s_flag;
#ifdef EPI_CX_LDS_PAD                                      // timing builds: LDS that nobody uses (workgroups per CU by LDS)
  __shared__ uint32_t s_pad[EPI_CX_LDS_PAD / 4];
  if (a.xm_cap == -12345) s_pad[threadIdx.x] = 1;
#endif
  if constexpr (!LEAN) {
    if (a.tile_list) {                                    // behind a lean launch: the tiles it listed, however many (fixed grid)
      const uint32_t n = *a.tile_list_count;
      for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int tile = (int)a.tile_list[i];
        const Tile td = a.tiles[tile];
        cx2_tile<T, G, NU, NP, FUSED, LEAN, 0>(a, tile, td, true, td.row_lo, L, s_scan, s_list, &s_flag);
        __syncthreads();                                  // (the next tile clears the arrays the emit just read)
      }
      return;
    }
  }
  if constexpr (PAD > 0) {
    // A WALKING workgroup: a.walk consecutive entries of the tile table.  While the next entry is the genomic neighbour of
    // the one just finished, its opening state is the shifted overhang and only the rows that start in it -- the rows
    // behind the previous entry's last row -- are visited; the first tile of a walk, and one behind a gap, a change of
    // reference sequence or a tile that was handed over (heavy, deep), starts from cleared arrays and visits all its
    // candidate rows, like a workgroup that does not walk.
    const int walk = a.walk;
    const int nruns = (ntiles + walk - 1) / walk;
    const int run = cx_tile_of_block(blockIdx.x, nruns);
    if (run >= nruns) return;
    const int t0 = run * walk, t1 = t0 + walk < ntiles ? t0 + walk : ntiles;
    // (the tile table through a constant-address-space pointer: it was written by the index kernel before this one and is
    //  read-only here, and this way a descriptor is a scalar load -- the next tile's is requested a whole tile ahead)
    typedef const Tile __attribute__((address_space(4))) *TileK;
    const TileK tk = (TileK)(uintptr_t)a.tiles;
    bool carry = false;
    int64_t prev_pos0 = 0;
    int32_t prev_rname = 0, prev_row_hi = 0;
    auto tile_at = [&](int i) { Tile t; t.pos0 = tk[i].pos0; t.rname = tk[i].rname; t.row_lo = tk[i].row_lo; t.row_hi = tk[i].row_hi; t.slot = tk[i].slot; return t; };
    Tile td = tile_at(t0);
    for (int tile = t0; tile < t1; tile++) {
      const Tile nxt = tile_at(tile + 1 < t1 ? tile + 1 : tile);
      // The thread index is read anew for every tile (it passes through an empty asm): everything derived from it
      // (lane-dependent LDS addresses, list slots of the emit) is otherwise computed once in front of the loop and kept in
      // vector registers throughout (21 of them spilled to scratch, measured; this kernel does not survive scratch)
      { uint32_t t = threadIdx.x; asm volatile("" : "+v"(t)); L.tid = t; }
      const bool cont = carry && td.rname == prev_rname && td.pos0 == prev_pos0 + T;
      const bool done = cx2_tile<T, G, NU, NP, FUSED, LEAN, PAD>(a, tile, td, !cont, cont ? prev_row_hi : td.row_lo, L, s_scan, s_list, &s_flag);
      carry = done;
      if (tile + 1 < t1) {
        __syncthreads();                                  // (the emit has read the counters; a handed-over tile leaves them untouched)
        if (done) { cx2_shift<T, NP, cx2_wg<LEAN>(), PAD>(L); __syncthreads(); }
      }
      prev_pos0 = td.pos0; prev_rname = td.rname; prev_row_hi = td.row_hi;
      td = nxt;
    }
    return;
  } else {
    const int tile = cx_tile_of_block(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const Tile td = a.tiles[tile];                        // (in flight while the counters are cleared)
    cx2_tile<T, G, NU, NP, FUSED, LEAN, 0>(a, tile, td, true, td.row_lo, L, s_scan, s_list, &s_flag);
  }
}

// One chunk of the candidate rows of one heavy tile: LDS sums as usual, then added into the tile's slab in HBM (or
// straight into its shared slab slot when other ranks contribute too).
template <int T, int G, int NU, int NP, bool FUSED>
__global__ __launch_bounds__(CX_WG, (cx2_waves_per_simd<T, NP, NU>())) void k_cx_heavy(Cx2Args a) {
  CX2_SHARED(T, NP, false, 0)
  const uint32_t hi_idx = (uint32_t)a.heavy_first + blockIdx.y;
  if (hi_idx >= *a.heavy_count) return;
  const int tile = (int)a.heavy_list[hi_idx];
  const Tile td = a.tiles[tile];
  for (int lo = td.row_lo + (int)blockIdx.x * a.heavy_chunk; lo < td.row_hi; lo += (int)gridDim.x * a.heavy_chunk) {
    const int hi = td.row_hi - lo > a.heavy_chunk ? lo + a.heavy_chunk : td.row_hi;
    cx2_clear<T, NP, false, CX_WG>(L);
    __syncthreads();
    cx2_accumulate<T, G, NU, NP, FUSED, false, CX_WG>(a, td, lo, hi, L);
    cx2_dump_slab<T, NP, false, CX_WG>(L, td.slot >= 0 ? a.slab + (int64_t)td.slot * (kCxPlanes * T)
                                          : a.heavy_slab + (int64_t)blockIdx.y * (kCxPlanes * T));
    __syncthreads();
  }
}

// Rule + rows of a tile whose sums sit in a slab: coverage difference arrays -> LDS, prefix sums, emit from HBM.
template <int T, int NP>
__device__ __forceinline__ void cx2_emit_from_slab(const Cx2Args &a, int tile, const int32_t *slab) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cov[2 * T];
  __shared__ uint32_t s_scan[CX_WG / 64 + 2];
  __shared__ uint16_t s_list[2 * T];
  for (int i = threadIdx.x; i < 2 * T; i += CX_WG) s_cov[i] = (uint32_t)slab[CX_SLAB_COV * T + i];
  __syncthreads();
  cx2_prefix<T>(s_cov, s_scan);
  cx2_prefix<T>(s_cov + T, s_scan);
  CxSrcSlab<T, NP> src;
  src.slab = slab; src.cov = s_cov;
  cx2_emit<T, NP>(a, tile, src, s_scan, s_list);
}

template <int T, int NP>
__global__ __launch_bounds__(CX_WG) void k_cx_emit_heavy(Cx2Args a) {
  const uint32_t hi_idx = (uint32_t)a.heavy_first + blockIdx.x;
  if (hi_idx >= *a.heavy_count) return;
  const int tile = (int)a.heavy_list[hi_idx];
  if (a.tiles[tile].slot >= 0) return;                     // emitted after the cross-rank reduce
  cx2_emit_from_slab<T, NP>(a, tile, a.heavy_slab + (int64_t)blockIdx.x * (kCxPlanes * T));
}

// Emits the shared tiles this rank owns from the (already cross-rank reduced) slab: one workgroup per shared slot.
template <int T, int NP>
__global__ __launch_bounds__(CX_WG) void k_cx_emit_slab(Cx2Args a, const int32_t *__restrict__ owned,
                                                         const int32_t *__restrict__ slot_tile) {
  if (!owned[blockIdx.x]) return;
  const int tile = slot_tile[blockIdx.x];
  if (tile < 0) return;
  cx2_emit_from_slab<T, NP>(a, tile, a.slab + (int64_t)a.tiles[tile].slot * (kCxPlanes * T));
}

// ---- reports of two or three contexts (CxG, CX): one u16-pair LDS atomic per base --------------------------------------
// With several reported contexts nearly every dword of a read holds a call, so the u8 scheme above would issue an LDS
// atomic per dword and context; here a base is one ds_add_u32 into packed u16 pairs [strand][('.',other),(H,h),(X,x),
// (Z,z)][T] with the byte order rotated per lane so that the 32 lanes of a half-wave hit 32 different banks
// (tile_common.hpp: cx_add_dword).  After the rows the counters are rewritten in place into what the common emit
// reads: plane 0 = coverage, context planes = n | M << 16.
#ifndef EPI_CXP_T
#define EPI_CXP_T 1024                        // (timing builds vary it)
#endif
constexpr int CXP_T = EPI_CXP_T;

template <int G, int U0, int U1>
__device__ __forceinline__ void cxp_add_range(const uint32_t (&w)[CX_NU], const RowSlice &cur) {
  if constexpr (U0 < U1) {
    if (U0 * G <= cur.tl) cx_add_dword<CXP_T, 4 * G * U0, U0 == 0, true>(w[U0], cur.tl == U0 * G, cur);
    cxp_add_range<G, U0 + 1, U1>(w, cur);
  }
}

// G lanes own a row (64/G rows per wavefront step); a lane keeps CX_NU dword loads of the row's in-tile slice in flight
// and the next step's row columns are fetched with them.
template <int G>
__device__ __forceinline__ void cxp_accumulate(const RowCols &c, const Tile &td, uint32_t *cnt) {
  constexpr int R = 64 / G, NW = CX_WG / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1), grp = lane / G;
  int r = td.row_lo + wave * R + grp;
  RowSlice cur = cx_row_slice<CXP_T, G, true>(c, td, r, sub, cnt);
  for (int rbase = td.row_lo + wave * R; rbase < td.row_hi; rbase += NW * R) {
    uint32_t w[CX_NU];                                    // every load of the slice is in flight before the first is used
#pragma unroll
    for (int u = 0; u < CX_NU; u++) w[u] = u * G <= cur.tl ? cur.src[u * G] : 0u;
    r += NW * R;
    const RowSlice nxt = cx_row_slice<CXP_T, G, true>(c, td, r, sub, cnt);
    cxp_add_range<G, 0, CX_NU>(w, cur);
    for (int k = sub + CX_NU * G; k < cur.nd; k += G) {   // slices longer than CX_NU*G dwords
      RowSlice t = cur;
#pragma unroll
      for (int j = 0; j < 4; j++) t.dst[j] = cur.dst[j] + 4 * (k - sub);
      cx_add_dword<CXP_T, 0, false, true>(cur.src[k - sub], k == cur.nd - 1, t);
    }
    cur = nxt;
  }
}

// packed pairs -> plane 0: coverage (:126-127: every counted base, nibble 9 twice), planes 1-3: n | M << 16 of H, X, Z
__device__ __forceinline__ void cxp_convert(uint32_t *cnt) {
  constexpr int T = CXP_T;
  for (int i = threadIdx.x; i < 2 * T; i += CX_WG) {
    uint32_t *c0 = cnt + (i / T) * 4 * T + (i % T);
    const uint32_t d0 = c0[0], dh = c0[T], dx = c0[2 * T], dz = c0[3 * T];
    const uint32_t nh = (dh & 0xFFFFu) + (dh >> 16), nx = (dx & 0xFFFFu) + (dx >> 16), nz = (dz & 0xFFFFu) + (dz >> 16);
    c0[0] = (d0 & 0xFFFFu) + (d0 >> 16) + nh + nx + nz;
    c0[T] = nh | (dh << 16);
    c0[2 * T] = nx | (dx << 16);
    c0[3 * T] = nz | (dz << 16);
  }
}

struct CxSrcPk {                          // the converted counters as the emit phase's source
  const uint32_t *cnt;
  int pl[3];                              // counter plane (1 = H, 2 = X, 3 = Z) of reported plane p, 0 = no such plane
  __device__ __forceinline__ uint32_t any(int sd, int pos) const {
    uint32_t x = 0;
#pragma unroll
    for (int p = 0; p < 3; p++) if (pl[p]) x |= cnt[(sd * 4 + pl[p]) * CXP_T + pos];
    return x;
  }
  __device__ __forceinline__ void pair(int sd, int p, int pos, uint32_t *n, uint32_t *M) const {
    const uint32_t w = pl[p] ? cnt[(sd * 4 + pl[p]) * CXP_T + pos] : 0u;
    *n = w & 0xFFFFu; *M = w >> 16;
  }
  __device__ __forceinline__ uint32_t coverage(int sd, int pos) const { return cnt[sd * 4 * CXP_T + pos]; }
};

__device__ __forceinline__ CxSrcPk cxp_source(const Cx2Args &a, const uint32_t *cnt) {
  CxSrcPk s;
  s.cnt = cnt;
#pragma unroll
  for (int p = 0; p < 3; p++) { const uint32_t k = (a.ctx_of_plane >> (8 * p)) & 255u; s.pl[p] = k == 2u ? 1 : k == 6u ? 2 : k == 7u ? 3 : 0; }
  return s;
}

// converted counters -> slab [16][T] in the common format ((n, M) per strand and reported plane, coverage as differences)
__device__ __forceinline__ void cxp_dump_slab(const Cx2Args &a, const uint32_t *cnt, int np, int32_t *slab) {
  constexpr int T = CXP_T;
  uint32_t *dst = reinterpret_cast<uint32_t *>(slab);
  const CxSrcPk src = cxp_source(a, cnt);
  for (int i = threadIdx.x; i < 2 * T; i += CX_WG) {
    const int sd = i / T, pos = i % T;
#pragma unroll
    for (int p = 0; p < 3; p++) {                          // (static plane index: the table stays in registers)
      if (p >= np) break;
      uint32_t n, M;
      src.pair(sd, p, pos, &n, &M);
      if (n) atomicAdd(dst + (2 * (sd * np + p)) * T + pos, n);
      if (M) atomicAdd(dst + (2 * (sd * np + p) + 1) * T + pos, M);
    }
    const uint32_t d = src.coverage(sd, pos) - (pos ? src.coverage(sd, pos - 1) : 0u);
    if (d) atomicAdd(dst + (CX_SLAB_COV + sd) * T + pos, d);
  }
}

#define CXP_SHARED                                                                                      \
  __shared__ __attribute__((aligned(16))) uint32_t cnt_raw[8 * CXP_T + 2 * kCxGuard];                  \
  __shared__ uint32_t s_scan[CX_WG / 64 + 2];                                                         \
  __shared__ uint16_t s_list[2 * CXP_T];                                                               \
  uint32_t *cnt = cnt_raw + kCxGuard;

template <int G>
__global__ __launch_bounds__(CX_WG, 8) void k_cxp_tiles(Cx2Args a, int ntiles, int np) {
  CXP_SHARED
  const int tile = cx_tile_of_block(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const Tile td = a.tiles[tile];                          // (in flight while the counters are cleared)
  uint4 *z = reinterpret_cast<uint4 *>(cnt_raw);
  for (int i = threadIdx.x; i < (8 * CXP_T + 2 * kCxGuard) / 4; i += CX_WG) z[i] = make_uint4(0, 0, 0, 0);
  if (td.row_hi - td.row_lo > a.heavy_rows) {
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
  cxp_accumulate<G>(a.c, td, cnt);
  __syncthreads();
  cxp_convert(cnt);
  __syncthreads();
  if (td.slot >= 0) {                                     // shared with another rank: hand the sums over
    cxp_dump_slab(a, cnt, np, a.slab + (int64_t)td.slot * (kCxPlanes * CXP_T));
    if (threadIdx.x == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
    return;
  }
  if (EPI_CX_ABLATE & 16) { if (threadIdx.x == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; } return; }   // timing builds: no emit
  cx2_emit<CXP_T, 3>(a, tile, cxp_source(a, cnt), s_scan, s_list);
}

template <int G>
__global__ __launch_bounds__(CX_WG, 8) void k_cxp_heavy(Cx2Args a, int np) {
  CXP_SHARED
  (void)s_scan; (void)s_list;
  const uint32_t hi_idx = (uint32_t)a.heavy_first + blockIdx.y;
  if (hi_idx >= *a.heavy_count) return;
  const int tile = (int)a.heavy_list[hi_idx];
  const Tile whole = a.tiles[tile];
  for (int lo = whole.row_lo + (int)blockIdx.x * a.heavy_chunk; lo < whole.row_hi; lo += (int)gridDim.x * a.heavy_chunk) {
    Tile td = whole;
    td.row_lo = lo;
    if (td.row_hi - lo > a.heavy_chunk) td.row_hi = lo + a.heavy_chunk;
    uint4 *z = reinterpret_cast<uint4 *>(cnt_raw);
    for (int i = threadIdx.x; i < (8 * CXP_T + 2 * kCxGuard) / 4; i += CX_WG) z[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    cxp_accumulate<G>(a.c, td, cnt);
    __syncthreads();
    cxp_convert(cnt);
    __syncthreads();
    cxp_dump_slab(a, cnt, np, td.slot >= 0 ? a.slab + (int64_t)td.slot * (kCxPlanes * CXP_T)
                                            : a.heavy_slab + (int64_t)blockIdx.y * (kCxPlanes * CXP_T));
    __syncthreads();
  }
}

struct __attribute__((packed, aligned(4))) CxU4u { uint32_t x, y, z, w; };   // 16 bytes at any dword address

// One wavefront per tile copies the tile's rows from the pool to their place in the final table
// (offset = exclusive scan of the tile row counts) and decodes them: contiguous reads, contiguous writes.
__global__ __launch_bounds__(256) void k_cx_gather(const Tile *__restrict__ tiles, const uint32_t *__restrict__ tile_out,
                                                    const uint32_t *__restrict__ tile_nrow, const uint32_t *__restrict__ tile_base,
                                                    int32_t ntiles, const uint32_t *__restrict__ pool_key,
                                                    const uint32_t *__restrict__ pool_meth, const uint32_t *__restrict__ pool_unmeth,
                                                    int32_t *__restrict__ o_rname, int32_t *__restrict__ o_strand,
                                                    int32_t *__restrict__ o_pos, int32_t *__restrict__ o_ctx,
                                                    int32_t *__restrict__ o_meth, int32_t *__restrict__ o_unmeth) {
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const uint32_t n = tile_nrow[tile];
  if (n == 0) return;
  const int lane = threadIdx.x & 63;
  const Tile td = tiles[tile];
  const uint32_t src0 = tile_base[tile], dst0 = tile_out[tile];
  // four consecutive rows per lane and instruction (16-byte loads and stores, any alignment: 1 KiB per wavefront
  // instruction instead of 256 bytes), the last few rows of the tile one by one
  const uint32_t n4 = n & ~3u;
  for (uint32_t i = 4u * lane; i < n4; i += 256) {
    const CxU4u key = *reinterpret_cast<const CxU4u *>(pool_key + src0 + i);
    const CxU4u me = *reinterpret_cast<const CxU4u *>(pool_meth + src0 + i);
    const CxU4u un = *reinterpret_cast<const CxU4u *>(pool_unmeth + src0 + i);
    const uint32_t o = dst0 + i;
    const uint32_t rn = (uint32_t)td.rname;
    const uint32_t p0 = (uint32_t)td.pos0;
    *reinterpret_cast<CxU4u *>(o_rname + o) = CxU4u{rn, rn, rn, rn};
    *reinterpret_cast<CxU4u *>(o_strand + o) = CxU4u{1u + ((key.x >> 3) & 1u), 1u + ((key.y >> 3) & 1u), 1u + ((key.z >> 3) & 1u), 1u + ((key.w >> 3) & 1u)};
    *reinterpret_cast<CxU4u *>(o_pos + o) = CxU4u{p0 + (key.x >> 4), p0 + (key.y >> 4), p0 + (key.z >> 4), p0 + (key.w >> 4)};
    *reinterpret_cast<CxU4u *>(o_ctx + o) = CxU4u{key.x & 7u, key.y & 7u, key.z & 7u, key.w & 7u};
    *reinterpret_cast<CxU4u *>(o_meth + o) = me;
    *reinterpret_cast<CxU4u *>(o_unmeth + o) = un;
  }
  for (uint32_t i = n4 + lane; i < n; i += 64) {
    const uint32_t key = pool_key[src0 + i];
    const uint32_t o = dst0 + i;
    o_rname[o] = td.rname;
    o_strand[o] = 1 + (int32_t)((key >> 3) & 1u);
    o_pos[o] = (int32_t)(td.pos0 + (int64_t)(key >> 4));
    o_ctx[o] = (int32_t)(key & 7u);
    o_meth[o] = (int32_t)pool_meth[src0 + i];
    o_unmeth[o] = (int32_t)pool_unmeth[src0 + i];
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------

// Tile size: 2048 positions for a single reported context (40 KiB of LDS, four workgroups per CU, fewer rows that
// reach into two tiles, fuller rounds), 1024 with two or three contexts (their counters take 32 / 44 KiB).
static int cx_tile_for(int np) { return np <= 1 ? CX_T1 : CXP_T; }

// Lanes per row and 16-byte chunks per lane.  Whole rows (fused thresholding) must fit one visit of G * NU chunks
// wherever they start inside their first chunk; so must the slice of a row inside a tile (at most T / 16 + 1 chunks).
// The smallest G * NU that holds them wins: idle chunk slots cost the same VALU as used ones, and fewer lanes per row
// are more rows per wavefront step (PE150: 20 chunks = 4 lanes x 5, against 8 x 3 = 24 slots).  Four to six chunks per
// lane need ~80 VGPRs, which only the lean kernel's 256-thread workgroups have.  Returned as G * 8 + NU.
// Lane shape (G lanes per row, NU chunks per lane) of the single-context tile kernels, as G * 8 + NU: the one with the least
// expected work per row over the batch's length histogram (RowStats::len_hist).  A row of g * nu chunks or fewer is one visit
// and costs ~ g * nu chunk slots + its share of the step's set-up (~ 2 g); a longer row is worked on in slices -- with fused
// thresholding twice, totals then calls -- and takes the 64 / g rows of its wavefront step along.  So the bulk of the rows
// picks the shape and a tail of long templates pays for itself: one 1 kb template among 100 000 PE150 ones used to widen the
// shape for all of them (config 2 with such a tail: 2.0 ms instead of 0.76, scratch/outlier_cost.py).
static int pick_cx_shape(const RowStats &st, int T, bool fused, bool lean) {
#ifdef EPI_CX_FORCE_SHAPE                                  // timing builds only: (G, NU) = (EPI_CX_FORCE_SHAPE / 8, % 8)
  if (fused) return EPI_CX_FORCE_SHAPE;
#endif
  const int max_chunks = (int)(((int64_t)st.max_len + 2 * (CX_CH - 1)) / CX_CH);
  const int tile_chunks = T / CX_CH;                       // (a slice of an un-thresholded report is clipped to the tile's position-aligned chunks)
  double total = 0;
  for (int k = 0; k < kLenBinCount; k++) total += st.len_hist[k];
  int best = 64 * 8 + 3;
  double best_cost = -1;
  for (int g = (fused || lean) ? 4 : 8; g <= 64; g <<= 1)
    for (int nu = 3; nu <= (lean ? 6 : 3); nu++) {
      const int cap = g * nu;
      double extra = 0;
      for (int k = 0; k < kLenBinCount; k++) {
        if (!st.len_hist[k]) continue;
        int ch = kLenBins[k] < max_chunks ? kLenBins[k] : max_chunks;
        if (!fused && ch > tile_chunks) ch = tile_chunks;
        const int slices = (ch + cap - 1) / cap;
        if (slices > 1) extra += (total > 0 ? st.len_hist[k] / total : 1.0) * (fused ? 2 * slices - 1 : slices - 1);
      }
      if (total == 0) {                                    // (no histogram: by the longest row, as rounds 1-3 did)
        int ch = max_chunks;
        if (!fused && ch > tile_chunks) ch = tile_chunks;
        if (cap < ch) continue;
      }
      const double cost = (cap + 2.0 * g) * (1.0 + (64 / g) * extra);
      if (best_cost < 0 || cost < best_cost) { best = g * 8 + nu; best_cost = cost; }   // ties: the earlier (fewer lanes per row) wins
    }
  return best;
}
// Fused thresholding decides a row from its whole length on every visit: for batches whose rows are (nearly all) short.  A
// tail of long rows goes slice by slice (cx2_rows); a long-read batch takes the per-read kernel + the un-thresholded report.
static bool cx_fused_fits(const RowStats &st) {
  if (st.max_len >= 65000) return false;                   // (class totals of a row travel as 16-bit halves)
  double total = 0, longer = 0;
  for (int k = 0; k < kLenBinCount; k++) { total += st.len_hist[k]; if (kLenBins[k] > 64 * 3) longer += st.len_hist[k]; }
  if (total == 0) return ((int64_t)st.max_len + 2 * (CX_CH - 1)) / CX_CH <= 64 * 3;
  return longer <= 0.01 * total;
}

constexpr unsigned CX_HEAVY_CAP = 8, CX_HEAVY_GRID = 128;   // ultra-deep tiles finished without a host round trip, work items each
constexpr unsigned CX_LIST_GRID = 512;        // workgroups of the general kernel behind a lean launch (they loop over the list)

// (timing builds) A walking workgroup's window holds CX_PAD positions behind its tile: rows of up to CX_PAD - 15 bytes (a row's last
// position-aligned chunk may end 15 positions behind its last byte), i.e. PE150 templates; four lanes per row.
[[maybe_unused]] constexpr int CX_PAD = 320;

template <int T, int NU, int NP, bool FUSED, bool LEAN>
static void launch_cx_tiles(int g, int nt, hipStream_t s, const Cx2Args &a) {
#ifdef EPI_CX_WALK_BUILD                                  // timing builds only (measured slower, profiles/r04_cx_experiments.txt): not in the product
  if constexpr (LEAN && NP == 1 && NU <= 5) {
    if (a.walk > 0 && g == 4 && !a.tile_list) {
      const int nruns = (nt + a.walk - 1) / a.walk;
      hipLaunchKernelGGL((k_cx_tiles<T, 4, NU, NP, FUSED, true, CX_PAD>), dim3((unsigned)(((nruns + 7) / 8) * 8)), dim3(cx2_wg<true>()), 0, s, a, nt);
      return;
    }
  }
#endif
  const unsigned nb = a.tile_list ? CX_LIST_GRID : (unsigned)(((nt + 7) / 8) * 8);
  switch (g) {
    case 4: if constexpr (FUSED || LEAN) { hipLaunchKernelGGL((k_cx_tiles<T, 4, NU, NP, FUSED, LEAN>), dim3(nb), dim3(cx2_wg<LEAN>()), 0, s, a, nt); } break;
    case 8: hipLaunchKernelGGL((k_cx_tiles<T, 8, NU, NP, FUSED, LEAN>), dim3(nb), dim3(cx2_wg<LEAN>()), 0, s, a, nt); break;
    case 16: hipLaunchKernelGGL((k_cx_tiles<T, 16, NU, NP, FUSED, LEAN>), dim3(nb), dim3(cx2_wg<LEAN>()), 0, s, a, nt); break;
    case 32: hipLaunchKernelGGL((k_cx_tiles<T, 32, NU, NP, FUSED, LEAN>), dim3(nb), dim3(cx2_wg<LEAN>()), 0, s, a, nt); break;
    default: hipLaunchKernelGGL((k_cx_tiles<T, 64, NU, NP, FUSED, LEAN>), dim3(nb), dim3(cx2_wg<LEAN>()), 0, s, a, nt); break;
  }
}

template <int T, int NU, int NP, bool FUSED>
static void launch_cx_g(bool heavy, bool lean, int g, int nt, dim3 grid, hipStream_t s, const Cx2Args &a) {
  if (!heavy) {
    if (lean) launch_cx_tiles<T, NU, NP, FUSED, true>(g, nt, s, a);
    else launch_cx_tiles<T, NU, NP, FUSED, false>(g, nt, s, a);
    return;
  }
  switch (g) {
    case 4: if constexpr (FUSED) { hipLaunchKernelGGL((k_cx_heavy<T, 4, NU, NP, FUSED>), grid, dim3(CX_WG), 0, s, a); } break;
    case 8: hipLaunchKernelGGL((k_cx_heavy<T, 8, NU, NP, FUSED>), grid, dim3(CX_WG), 0, s, a); break;
    case 16: hipLaunchKernelGGL((k_cx_heavy<T, 16, NU, NP, FUSED>), grid, dim3(CX_WG), 0, s, a); break;
    case 32: hipLaunchKernelGGL((k_cx_heavy<T, 32, NU, NP, FUSED>), grid, dim3(CX_WG), 0, s, a); break;
    default: hipLaunchKernelGGL((k_cx_heavy<T, 64, NU, NP, FUSED>), grid, dim3(CX_WG), 0, s, a); break;
  }
  hipLaunchKernelGGL((k_cx_emit_heavy<T, NP>), dim3(grid.y), dim3(CX_WG), 0, s, a);
}

// lanes per row of the packed-pair kernel: enough that CX_NU dwords per lane cover the longest in-tile slice
static int pick_cxp_group(int32_t max_len) {
  const int slice = (max_len < CXP_T ? max_len : CXP_T) + 3;
  const int nd = (slice + 3) / 4;
  int g = 8;
  while (g < 64 && g * CX_NU < nd) g <<= 1;
  return g;
}
// ... for the bulk of the rows (cxp_accumulate walks the rest of a longer slice dword by dword): the smallest group that
// holds all but 0.5 % of the rows in one pass, never wider than the longest row needs
static int pick_cxp_group(const RowStats &st) {
  const int by_max = pick_cxp_group(st.max_len);
  double total = 0;
  for (int k = 0; k < kLenBinCount; k++) total += st.len_hist[k];
  if (total == 0) return by_max;
  for (int g = 8; g < by_max; g <<= 1) {
    double longer = 0;
    for (int k = 0; k < kLenBinCount; k++) {
      const int64_t len_k = (int64_t)kLenBins[k] * 16 - 15;                 // the longest row of bin k
      const int64_t slice = (len_k < CXP_T ? len_k : CXP_T) + 3;
      if ((slice + 3) / 4 > (int64_t)g * CX_NU) longer += st.len_hist[k];
    }
    if (longer <= 0.005 * total) return g;
  }
  return by_max;
}

static void launch_cxp(bool heavy, int np, int g, int nt, dim3 grid, hipStream_t s, const Cx2Args &a) {
  if (!heavy) {
    const unsigned nb = (unsigned)(((nt + 7) / 8) * 8);
    switch (g) {
      case 8: hipLaunchKernelGGL((k_cxp_tiles<8>), dim3(nb), dim3(CX_WG), 0, s, a, nt, np); break;
      case 16: hipLaunchKernelGGL((k_cxp_tiles<16>), dim3(nb), dim3(CX_WG), 0, s, a, nt, np); break;
      case 32: hipLaunchKernelGGL((k_cxp_tiles<32>), dim3(nb), dim3(CX_WG), 0, s, a, nt, np); break;
      default: hipLaunchKernelGGL((k_cxp_tiles<64>), dim3(nb), dim3(CX_WG), 0, s, a, nt, np); break;
    }
    return;
  }
  switch (g) {
    case 8: hipLaunchKernelGGL((k_cxp_heavy<8>), grid, dim3(CX_WG), 0, s, a, np); break;
    case 16: hipLaunchKernelGGL((k_cxp_heavy<16>), grid, dim3(CX_WG), 0, s, a, np); break;
    case 32: hipLaunchKernelGGL((k_cxp_heavy<32>), grid, dim3(CX_WG), 0, s, a, np); break;
    default: hipLaunchKernelGGL((k_cxp_heavy<64>), grid, dim3(CX_WG), 0, s, a, np); break;
  }
  if (np == 2) hipLaunchKernelGGL((k_cx_emit_heavy<CXP_T, 2>), dim3(grid.y), dim3(CX_WG), 0, s, a);
  else hipLaunchKernelGGL((k_cx_emit_heavy<CXP_T, 3>), dim3(grid.y), dim3(CX_WG), 0, s, a);
}

static void launch_cx(bool heavy, int np, bool fused, bool lean, int shape, int nt, dim3 grid, hipStream_t s, const Cx2Args &a) {
  const int g = shape >> 3;
  if (np > 1) { launch_cxp(heavy, np, g, nt, grid, s, a); return; }   // several contexts: one packed-pair atomic per base
  if (!heavy && lean && (shape & 7) > 3) {                 // one context, 2048-position tiles, four to six chunks per lane
    switch (shape & 7) {
      case 4: if (fused) launch_cx_tiles<CX_T1, 4, 1, true, true>(g, nt, s, a); else launch_cx_tiles<CX_T1, 4, 1, false, true>(g, nt, s, a); break;
      case 5: if (fused) launch_cx_tiles<CX_T1, 5, 1, true, true>(g, nt, s, a); else launch_cx_tiles<CX_T1, 5, 1, false, true>(g, nt, s, a); break;
      default: if (fused) launch_cx_tiles<CX_T1, 6, 1, true, true>(g, nt, s, a); else launch_cx_tiles<CX_T1, 6, 1, false, true>(g, nt, s, a); break;
    }
    return;
  }
  if (fused) launch_cx_g<CX_T1, 3, 1, true>(heavy, lean, g, nt, grid, s, a);
  else launch_cx_g<CX_T1, 3, 1, false>(heavy, lean, g, nt, grid, s, a);
}

static void launch_cx_emit_slab(int np, int nshared, hipStream_t s, const Cx2Args &a, const int32_t *owned, const int32_t *slot_tile) {
  if (np == 1) hipLaunchKernelGGL((k_cx_emit_slab<CX_T1, 1>), dim3((unsigned)nshared), dim3(CX_WG), 0, s, a, owned, slot_tile);
  else if (np == 2) hipLaunchKernelGGL((k_cx_emit_slab<1024, 2>), dim3((unsigned)nshared), dim3(CX_WG), 0, s, a, owned, slot_tile);
  else hipLaunchKernelGGL((k_cx_emit_slab<1024, 3>), dim3((unsigned)nshared), dim3(CX_WG), 0, s, a, owned, slot_tile);
}

static int ensure_pool(epi_batch *b, size_t rows) {
  if (rows <= b->pool_cap && b->pool_key.p) return EPI_OK;
  EPI_TRY(b->pool_key.ensure(rows * 4));
  EPI_TRY(b->pool_a.ensure(rows * 4));
  EPI_TRY(b->pool_b.ensure(rows * 4));
  b->pool_cap = rows;
  return EPI_OK;
}

// Report LUT for the context string (rcpp_cx_report.cpp:88-91): planes in the rule's order H, X, Z.  Returns the
// number of reported contexts (0: nothing can be reported).
static int make_report_lut(uint32_t ctx_mask, ClassLut *lut, uint32_t *ctx_of_plane) {
  uint32_t f[16] = {0};
  int np = 0;
  *ctx_of_plane = 0;
  for (uint32_t k : {2u, 6u, 7u}) {
    if (!((ctx_mask >> k) & 1u)) continue;
    f[k] |= 3u << (2 * np);                                // methylated: a call (n) and M of plane np
    f[k + 8] |= 1u << (2 * np);                            // unmethylated (or lower-cased): a call only
    *ctx_of_plane |= k << (8 * np);
    np++;
  }
  f[11] |= 0x40u;                                          // skipped (:123)
  f[9] |= 0x80u;                                           // counts twice in the coverage (:126-127)
  auto pack = [&](int b0) { return f[b0] | (f[b0 + 1] << 8) | (f[b0 + 2] << 16) | (f[b0 + 3] << 24); };
  lut->lo0 = pack(0); lut->lo1 = pack(4); lut->hi0 = pack(8); lut->hi1 = pack(12);
  return np;
}

static uint32_t lut_byte(const ClassLut &l, int code) {
  const uint32_t w = code < 4 ? l.lo0 : code < 8 ? l.lo1 : code < 12 ? l.hi0 : l.hi1;
  return (w >> (8 * (code & 3))) & 255u;
}

struct CxThreshold {                      // fused thresholding request (null = use the pass vector)
  const char *cls[4];
  ThrParams prm;
};

// The fused kernel's LUT.  Fusable: the report has one context k, the thresholding classes are that context
// (ctx_meth = {k}, ctx_unmeth = {k | 8}) and no class string repeats a letter.  Also picks the stand-in code for bytes
// outside a row (no bit set in this LUT).
static bool make_fused_lut(const CxThreshold &t, uint32_t ctx_of_plane, int np, ClassLut *lut, uint32_t *fill4) {
  if (np != 1) return false;
  const uint32_t k = ctx_of_plane & 255u;
  unsigned w[4][16] = {{0}};
  for (int c = 0; c < 4; c++)
    if (t.cls[c]) for (const unsigned char *p = reinterpret_cast<const unsigned char *>(t.cls[c]); *p; p++) w[c][ctx_to_idx(*p)]++;
  for (int c = 0; c < 4; c++) for (int i = 0; i < 16; i++) if (w[c][i] > 1) return false;
  for (uint32_t i = 0; i < 16; i++) {
    if ((w[0][i] != 0) != (i == k)) return false;          // ctx_meth is exactly the reported context's methylated code
    if ((w[1][i] != 0) != (i == (k | 8u))) return false;   // ctx_unmeth its unmethylated one
  }
  uint32_t f[16];
  for (uint32_t i = 0; i < 16; i++) {
    f[i] = 0;
    if (i == k || i == (k | 8u)) f[i] |= 0x01u;            // in context, either case: n_m + n_u, and a call
    if (i == k) f[i] |= 0x04u;                             // methylated: n_m, and M for a passing read
    if (w[2][i]) f[i] |= 0x10u;
    if (w[3][i]) f[i] |= 0x40u;
    if (i == 11) f[i] |= 0x02u;                            // skipped as is (:123) ...
    if (i == 9) f[i] |= 0x08u;                             // counts twice in the coverage (:126-127)
    if ((i | 8u) == 11) f[i] |= 0x20u;                     // ... and when the read is lower-cased (c | 8, :122)
    if ((i | 8u) == 9) f[i] |= 0x80u;
  }
  int fill = -1;
  for (int c : {12, 8, 0, 4}) if (f[c] == 0) { fill = c; break; }
  if (fill < 0) return false;
  auto pack = [&](int b0) { return f[b0] | (f[b0 + 1] << 8) | (f[b0 + 2] << 16) | (f[b0 + 3] << 24); };
  lut->lo0 = pack(0); lut->lo1 = pack(4); lut->hi0 = pack(8); lut->hi1 = pack(12);
  *fill4 = 0x01010101u * (uint32_t)fill;
  return true;
}

// The CX report on a resident batch.  thr != null: thresholding fused into the tile kernel when the batch allows it
// (the report's one context is the thresholding context, reads of at most ~5 kb), else a separate pass of the
// per-read kernel first.
static int cx_report_impl(epi_batch *b, const int32_t *d_pass, const CxThreshold *thr, int32_t *d_pass_out, const char *ctx,
                          hipStream_t s, int64_t *nrow_out) {
  *nrow_out = 0;
  b->last_kind = 0;
  uint32_t ctx_mask = 0;                                   // rcpp_cx_report.cpp:88-91
  for (const unsigned char *c = reinterpret_cast<const unsigned char *>(ctx); *c; c++) ctx_mask |= 1u << ctx_to_idx(*c);

  Cx2Args a;
  memset(&a, 0, sizeof(a));
  const int np = make_report_lut(ctx_mask, &a.lut_r, &a.ctx_of_plane);
  a.lut_rx = lut16_xor_form(a.lut_r);
  const int T = cx_tile_for(np);
  RowStats st;
  int32_t nt = 0;
  bool nt_hinted = false;
  EPI_TRY(build_tiles(b, s, T, &st, &nt, &nt_hinted));
  b->last_ntiles = nt;
  b->last_tile = T;
  a.fill4 = 0x0C0C0C0Cu;                                   // '.': never a call, never skipped, also when lower-cased
  bool fused = false;
  if (thr) {
    uint32_t fill4 = 0;
    if (nt > 0 && np > 0 && cx_fused_fits(st) && make_fused_lut(*thr, a.ctx_of_plane, np, &a.lut_s, &fill4)) {
      fused = true;
      a.lut_sx = lut16_xor_form(a.lut_s);
      a.thr = thr->prm;
      a.fill4 = fill4;
      a.pass_out = d_pass_out;
      // decisions from a table over the possible class totals (filled on the device with the reference's own
      // expressions); kept while the thresholds do not change
      // (field by field: the caller's struct may carry padding; doubles bitwise, so that a NaN threshold compares equal to itself)
      const bool same = b->thr_tab_len == st.max_len && b->thr_tab_prm.min_n_ctx == thr->prm.min_n_ctx &&
                        memcmp(&b->thr_tab_prm.min_ctx_meth_frac, &thr->prm.min_ctx_meth_frac, sizeof(double)) == 0 &&
                        memcmp(&b->thr_tab_prm.max_ooctx_meth_frac, &thr->prm.max_ooctx_meth_frac, sizeof(double)) == 0;
      if (!same) {
        EPI_TRY(b->thr_tab.ensure((size_t)(st.max_len + 1) * 4));
        hipLaunchKernelGGL(k_thr_table, dim3((unsigned)(st.max_len / 256 + 1)), dim3(256), 0, s, thr->prm, st.max_len, b->thr_tab.as<uint32_t>());
        EPI_HIP(hipGetLastError());
        b->thr_tab_len = st.max_len;
        memset(&b->thr_tab_prm, 0, sizeof(ThrParams));
        b->thr_tab_prm.min_n_ctx = thr->prm.min_n_ctx;
        b->thr_tab_prm.min_ctx_meth_frac = thr->prm.min_ctx_meth_frac;
        b->thr_tab_prm.max_ooctx_meth_frac = thr->prm.max_ooctx_meth_frac;
      }
      a.thr_tab = b->thr_tab.as<uint32_t>();
    } else if (b->n > 0) {
      // not fusable: the per-read kernel decides first (into the caller's buffer, or a scratch column)
      int32_t *dst = d_pass_out;
      if (!dst) { EPI_TRY(b->pass_tmp.ensure((size_t)b->n * 4)); dst = b->pass_tmp.as<int32_t>(); }
      EPI_TRY(epi_batch_threshold_reads_dev(b, thr->cls[0], thr->cls[1], thr->cls[2] ? thr->cls[2] : "", thr->cls[3] ? thr->cls[3] : "",
                                            thr->prm.min_n_ctx, thr->prm.min_ctx_meth_frac, thr->prm.max_ooctx_meth_frac, dst, s));
      d_pass = dst;
    }
  }
  if (nt == 0 || np == 0) {                                // no rows, or a context string without H/X/Z: an empty table
    // (a rank of a sharded run without rows still takes part in the exchange: its second half returns the empty table)
    b->last_kind = !b->shared_keys.empty() && b->d_slab ? 3 : 1; b->last_nrow = 0; b->last_ntiles = 0;
    return EPI_OK;
  }

  EPI_TRY(b->tile_nrow.ensure((size_t)nt * 4));
  EPI_TRY(b->tile_base.ensure((size_t)nt * 4));
  EPI_TRY(b->tile_out.ensure((size_t)(nt + 1) * 4));
  // Row pool = one slot per tile + an overflow region behind the slots (cx_pool_reserve).  The slot size starts
  // at a typical density of reported cytosines (CpG ~6 % of the (pos,strand) cells of a tile, all contexts ~40 %)
  // and doubles for the next call when more than 1/8 of the rows went through the overflow cursor; an overflow of
  // the region itself is detected below and costs one rerun with the exact size.
  const int32_t nshared = (int32_t)b->shared_keys.size();
  const size_t headroom = nshared > 0 ? (size_t)nshared * 2 * T : 0;   // shared tiles are emitted later into the same pool
  uint32_t &slot_state = (ctx_mask & ~(1u << 7)) ? b->cx_slot_wide : b->cx_slot_cg;
  if (!slot_state) slot_state = (ctx_mask & ~(1u << 7)) ? (uint32_t)(3 * T) / 4 : (uint32_t)T / 8;
  uint32_t slot = slot_state > (uint32_t)(2 * T) ? (uint32_t)(2 * T) : slot_state;
  if (options().cx_slot >= 0 && options().cx_slot <= 2 * T) slot = (uint32_t)options().cx_slot;   // test hook (EPIHIP_CX_SLOT)
  while (slot && (unsigned long long)nt * slot > 0xC0000000ull) slot >>= 1;   // row indices are u32
  size_t ovf_base = (size_t)nt * slot;
  for (;;) {
    const size_t ovf = (ovf_base >> 4) > 65536 ? (ovf_base >> 4) : 65536;
    if (b->pool_cap >= ovf_base + ovf + headroom) break;
    const int rc = ensure_pool(b, ovf_base + ovf + headroom);
    if (rc == EPI_OK) break;
    b->pool_cap = 0;                                       // (a failed growth has released the old buffers)
    if (!slot) return rc;
    slot = 0;                                              // the slots do not fit in device memory: every tile through the
    ovf_base = 0;                                          // cursor, the pool sized by the rows actually produced
  }
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;           // misc[1] = pool cursor, misc[2] = nrow total
  // Single-context reports run the LEAN kernel (u8 counters, no folds).  Where a position may be covered by more than
  // 255 rows -- RowStats::deep says whether that can happen anywhere in the batch -- the lean kernel decides tile by tile
  // and lists the deep ones for the general kernel, which is queued right behind it (no host round trip in between).
  bool lean = np == 1;
  if (!options().cx_lean) lean = false;                    // test hook (EPIHIP_CX_LEAN=0): the general kernel for every tile
  const bool per_tile = lean && st.deep != 0;
  // lanes per row * 8 + chunks per lane; the heavy-tile kernel is the general one (three chunks per lane)
  const int grp = np > 1 ? pick_cxp_group(st) * 8 : pick_cx_shape(st, T, fused, lean);
  const int grp_heavy = np > 1 ? grp : pick_cx_shape(st, T, fused, false);

  // Walking workgroups (timing builds, EPI_CX_WALK_BUILD + EPIHIP_CX_WALK=K; lean kernel, rows of up to CX_PAD - 15 bytes = the
  // four-lane shapes): every row analysed once, by the tile it starts in.  Bit-exact, but slower than one tile per
  // workgroup at every K on config 2 (the second visits it removes are L2 hits and the kernel is bound by its memory
  // pipeline and by the number of resident workgroups, not by row visits): profiles/r04_cx_experiments.txt.
  a.walk = 0;
#ifdef EPI_CX_WALK_BUILD
  if (lean && np == 1 && (grp >> 3) == 4 && (grp & 7) <= 5 && (int64_t)st.max_len + (CX_CH - 1) <= CX_PAD && options().cx_walk > 0)
    a.walk = options().cx_walk > 64 ? 64 : options().cx_walk;
#endif
  a.c.xm = b->xm; a.c.off = b->off; a.c.len = b->len; a.c.start = b->start; a.c.strand = b->strand; a.c.pass = fused ? nullptr : d_pass;
  a.xm_cap = (b->nbytes + 15) / 16 * 16;                   // (both batch constructors guarantee this much)
  a.tiles = b->tiles.as<Tile>();
  a.cursor = cursor;
  a.tile_nrow = b->tile_nrow.as<uint32_t>();
  a.tile_base = b->tile_base.as<uint32_t>();
  a.slab = b->d_slab;
  a.heavy_rows = 16384;
  if (options().heavy_rows > 0) a.heavy_rows = options().heavy_rows;   // test hook (EPIHIP_HEAVY_ROWS)
  if (a.heavy_rows > 16384) a.heavy_rows = 16384;          // u16 pairs and the packed coverage halves: a base adds at most 2
  a.heavy_chunk = a.heavy_rows / 4 > 64 ? a.heavy_rows / 4 : 64;
  if (a.heavy_chunk > 256) a.heavy_chunk = 256;            // (a 20 000-row pile-up is then 80 work items, not 5)
  EPI_TRY(b->heavy_list.ensure((size_t)nt * 4));
  a.heavy_list = b->heavy_list.as<uint32_t>();
  a.heavy_count = b->misc.as<uint32_t>() + 3;             // misc[3] = heavy tiles, misc[8] = their largest row count
  a.heavy_max = b->misc.as<uint32_t>() + 8;
  a.heavy_slab = nullptr;
  if (per_tile) {
    EPI_TRY(b->deep_list.ensure((size_t)nt * 4));
    a.deep_list = b->deep_list.as<uint32_t>();
    a.deep_count = b->misc.as<uint32_t>() + 4;            // misc[4] = tiles handed from the lean to the general kernel
  }
  a.slot_rows = slot;
  a.ovf_base = (uint32_t)ovf_base;
  b->cx_last_slot = slot;
  b->cx_last_ovf = (uint32_t)ovf_base;
  b->cx_last_np = np;
  b->cx_last_ctx_of_plane = a.ctx_of_plane;
  EPI_TRY(check_grid(((int64_t)nt + 7) / 8 * 8, np == 1 && lean ? cx2_wg<true>() : CX_WG, "CX tile kernel"));
  a.nrows = b->n;
#ifdef EPI_CHECK
  EPI_TRY(b->diag.ensure(256));
  a.dbg = b->diag.as<uint32_t>();
  EPI_HIP(hipMemsetAsync(a.dbg, 0, 32, s));
#endif
  uint32_t used_total[2] = {0, 0};
  bool host_heavy = false;                                 // ultra-deep tiles had to be finished after the synchronisation
  b->cx_deferred = false;
  for (int attempt = 0; attempt < 2; attempt++) {
    a.pool_key = b->pool_key.as<uint32_t>();
    a.pool_meth = b->pool_a.as<uint32_t>();
    a.pool_unmeth = b->pool_b.as<uint32_t>();
    a.pool_cap = (uint32_t)(b->pool_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : b->pool_cap);
    if (attempt > 0) {                                   // (the tile-index pass zeroed them for the first attempt)
      EPI_HIP(hipMemsetAsync(cursor, 0, 16, s));         // cursor, total, heavy count, deep count
      EPI_HIP(hipMemsetAsync(a.heavy_max, 0, 4, s));
    }
    prof_begin("cx_tiles", s);
    launch_cx(false, np, fused, lean, grp, nt, dim3(1), s, a);
    prof_end("cx_tiles", s);
    if (per_tile) {                                      // the tiles the lean kernel listed, by the general kernel
      Cx2Args g = a;
      g.tile_list = a.deep_list;
      g.tile_list_count = a.deep_count;
      g.deep_list = nullptr; g.deep_count = nullptr;
      prof_begin("cx_deep", s);
      launch_cx(false, np, fused, false, grp_heavy, nt, dim3(1), s, g);
      prof_end("cx_deep", s);
    }
    // Batches with deep positions may also hold ultra-deep tiles (set aside by the kernels above): the first CX_HEAVY_CAP of
    // them are split, reduced and emitted by fixed-grid launches queued right here -- no host round trip for one pile-up
    // in a WGS batch; a larger number is finished below, once the count is known.
    const bool chained = st.deep != 0;
    if (chained) {
      EPI_TRY(b->heavy_slab.ensure((size_t)CX_HEAVY_CAP * kCxPlanes * T * 4));
      a.heavy_slab = b->heavy_slab.as<int32_t>();
      a.heavy_first = 0;
      EPI_HIP(hipMemsetAsync(a.heavy_slab, 0, (size_t)CX_HEAVY_CAP * kCxPlanes * T * 4, s));
      prof_begin("cx_heavy", s);
      launch_cx(true, np, fused, lean, grp_heavy, nt, dim3(CX_HEAVY_GRID, CX_HEAVY_CAP), s, a);
      prof_end("cx_heavy", s);
    }
    EPI_HIP(hipGetLastError());
    if (b->cx_defer && nshared > 0 && attempt == 0 && nt_hinted && b->cx_noheavy_T == T && b->cx_noheavy_rows == a.heavy_rows) {
      // sharded report with one host synchronisation: nothing is read back here; epi_batch_cx_finish_shared checks the tile
      // count, the heavy-tile count and the pool at its own synchronisation (comm.hip reruns the first half into a scratch
      // slab if the pool turns out too small)
      b->cx_deferred = true;
      b->cx_def_hinted = nt_hinted;
      b->cx_def_heavy_done = chained ? CX_HEAVY_CAP : 0u;
      b->cx_def_headroom = headroom;
      b->last_kind = 3;
      return EPI_OK;
    }
    // row offsets of the tiles are queued right away; {rows handed out, total rows, heavy tiles} come back in one sync
    EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
    uint32_t host9[9];
    EPI_TRY(read_scalars(b, s, cursor - 1, 36, host9));    // misc[0..8]
    uint32_t *host = host9 + 1;
    if (nt_hinted && host9[0] != (uint32_t)nt) {
      for (int i = 0; i < 4; i++) b->tile_hint_T[i] = 0;
      return fail(EPI_ERR_STATE, "the rows of this batch changed since an earlier report (tile count %u, was %d)", host9[0], nt);
    }
    const uint32_t heavy_done = chained ? CX_HEAVY_CAP : 0u;
    if (host[2] > heavy_done) {
      host_heavy = true;
      b->cx_noheavy_T = 0;
      // ultra-deep tiles were set aside (and not finished above): split each over ceil(rows/chunk) workgroups, reduce in
      // HBM, emit, rescan
      const uint32_t nheavy = host[2] - heavy_done, nchunks = (host[7] + (uint32_t)a.heavy_chunk - 1) / (uint32_t)a.heavy_chunk;
      a.heavy_first = (int)heavy_done;
      EPI_TRY(b->heavy_slab.ensure((size_t)nheavy * kCxPlanes * T * 4));
      a.heavy_slab = b->heavy_slab.as<int32_t>();
      EPI_HIP(hipMemsetAsync(a.heavy_slab, 0, (size_t)nheavy * kCxPlanes * T * 4, s));
      prof_begin("cx_heavy", s);
      launch_cx(true, np, fused, lean, grp_heavy, nt, dim3(nchunks, nheavy), s, a);
      prof_end("cx_heavy", s);
      EPI_HIP(hipGetLastError());
      EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
      EPI_TRY(read_scalars(b, s, cursor, 8, host));
    }
    used_total[0] = host[0];
    used_total[1] = host[1];
#ifdef EPI_CHECK
    {
      uint32_t d[8];
      EPI_HIP(hipMemcpy(d, a.dbg, 32, hipMemcpyDeviceToHost));
      if (d[0]) return fail(EPI_ERR_STATE, "CX index check %u failed: v0=%lld v1=%d block=%u thread=%u (n=%lld nt=%d attempt=%d)", d[0],
                            (long long)(((uint64_t)d[5] << 32) | d[1]), (int)d[2], d[3], d[4], (long long)b->n, nt, attempt);
    }
#endif
    if (ovf_base + used_total[0] + headroom <= a.pool_cap) break;
    if (attempt == 1) return fail(EPI_ERR_STATE, "row pool overflow after regrow");
    EPI_TRY(ensure_pool(b, ovf_base + used_total[0] + (used_total[0] >> 4) + 1024 + headroom));   // exact need is known now: rerun once
    if (nshared > 0)   // the rerun adds into the slab again
      EPI_HIP(hipMemsetAsync(b->d_slab, 0, (size_t)nshared * kCxPlanes * T * 4, s));
  }
  if (used_total[0] > used_total[1] / 8 && slot_state < (uint32_t)(2 * T)) slot_state *= 2;   // too many tiles outgrew their slot
  // (no ultra-deep tile was left for the host to finish: a later sharded report on this batch may defer its synchronisation)
  if (!host_heavy) { b->cx_noheavy_T = T; b->cx_noheavy_rows = a.heavy_rows; }
  if (nshared > 0) { b->last_kind = 3; return EPI_OK; }     // caller continues with epi_batch_cx_finish_shared
  b->last_kind = 1;
  b->last_nrow = used_total[1];
  *nrow_out = used_total[1];
  return EPI_OK;
}

}  // namespace epi

using namespace epi;

extern "C" {

int epi_tile_positions(void) { return cx_tile_for(1); }

int epi_cx_tile_positions(const char *ctx) {
  uint32_t ctx_mask = 0, cop = 0;
  if (ctx) for (const unsigned char *c = reinterpret_cast<const unsigned char *>(ctx); *c; c++) ctx_mask |= 1u << ctx_to_idx(*c);
  ClassLut l;
  return cx_tile_for(make_report_lut(ctx_mask, &l, &cop));
}

int epi_batch_cx_report_dev(epi_batch *b, const int32_t *d_pass, const char *ctx, void *stream, int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_cx_report_dev: NULL argument");
  EPI_HIP(hipSetDevice(b->eng->device));
  return cx_report_impl(b, d_pass, nullptr, nullptr, ctx, pick_stream(b, stream), nrow_out);
}

int epi_batch_cytosine_report_dev(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                                  const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                                  double max_ooctx_meth_frac, const char *ctx, int32_t *d_pass_out, void *stream,
                                  int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out || !ctx_meth || !ctx_unmeth) return fail(EPI_ERR_ARG, "epi_batch_cytosine_report_dev: NULL argument");
  EPI_HIP(hipSetDevice(b->eng->device));
  CxThreshold t;
  t.cls[0] = ctx_meth; t.cls[1] = ctx_unmeth; t.cls[2] = ooctx_meth ? ooctx_meth : ""; t.cls[3] = ooctx_unmeth ? ooctx_unmeth : "";
  t.prm.min_n_ctx = min_n_ctx; t.prm.min_ctx_meth_frac = min_ctx_meth_frac; t.prm.max_ooctx_meth_frac = max_ooctx_meth_frac;
  return cx_report_impl(b, nullptr, &t, d_pass_out, ctx, pick_stream(b, stream), nrow_out);
}

// Second half of a sharded report: the slab has been sum-reduced across ranks;
// emit the shared tiles this rank owns, then order all rows.
int epi_batch_cx_finish_shared(epi_batch *b, const char *ctx, void *stream, int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_cx_finish_shared: NULL argument");
  if (b->last_kind != 3) return fail(EPI_ERR_STATE, "epi_batch_cx_finish_shared without a sharded epi_batch_cx_report_dev");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  const int32_t nt = b->last_ntiles;
  if (nt == 0) { b->last_kind = 1; b->last_nrow = 0; *nrow_out = 0; return EPI_OK; }   // this rank holds no rows: owns no tile
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;
  Cx2Args a;
  memset(&a, 0, sizeof(a));
  a.tiles = b->tiles.as<Tile>();
  a.ctx_of_plane = b->cx_last_ctx_of_plane;
  a.cursor = cursor;
  a.tile_nrow = b->tile_nrow.as<uint32_t>();
  a.tile_base = b->tile_base.as<uint32_t>();
  a.slab = b->d_slab;
  a.pool_key = b->pool_key.as<uint32_t>();
  a.pool_meth = b->pool_a.as<uint32_t>();
  a.pool_unmeth = b->pool_b.as<uint32_t>();
  a.pool_cap = (uint32_t)(b->pool_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : b->pool_cap);
  a.slot_rows = b->cx_last_slot;
  a.ovf_base = b->cx_last_ovf;
  launch_cx_emit_slab(b->cx_last_np, (int)b->shared_keys.size(), s, a, b->d_shared_owned.as<int32_t>(),
                      b->d_slot_tile.as<int32_t>());
  EPI_HIP(hipGetLastError());
  uint32_t *d_total = b->misc.as<uint32_t>() + 2;
  EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, d_total, b->scan_tmp, s));
  uint32_t ut[2] = {0, 0};                                  // {overflow rows handed out, total rows}: one sync
  if (b->cx_deferred) {
    // the first half read nothing back: its checks happen here, with the report's only synchronisation
    b->cx_deferred = false;
    uint32_t host9[9];
    EPI_TRY(read_scalars(b, s, cursor - 1, 36, host9));     // misc[0..8]
    if (b->cx_def_hinted && host9[0] != (uint32_t)nt) {
      for (int i = 0; i < 4; i++) b->tile_hint_T[i] = 0;
      return fail(EPI_ERR_STATE, "the rows of this batch changed since an earlier report (tile count %u, was %d)", host9[0], nt);
    }
    if (host9[3] > b->cx_def_heavy_done) {
      b->cx_noheavy_T = 0;
      return fail(EPI_ERR_STATE, "ultra-deep tiles appeared in a batch that had none (%u): the rows of this batch changed", host9[3]);
    }
    if ((size_t)a.ovf_base + host9[1] > a.pool_cap) {         // (the cursor has served the shared tiles' rows as well by now)
      // the pool was too small for the rows of this report: the caller reruns the first half (which grows it) into a
      // scratch slab and emits the shared tiles from the slab that has already been reduced
      b->last_kind = 0;
      return EPI_RETRY_POOL;
    }
    ut[0] = host9[1]; ut[1] = host9[2];
  } else {
    EPI_TRY(read_scalars(b, s, cursor, 8, ut));
  }
  // cannot overflow: the first half kept 2*kTile rows per shared tile free
  if ((size_t)a.ovf_base + ut[0] > a.pool_cap) return fail(EPI_ERR_STATE, "row pool overflow in sharded report");
  const uint32_t total = ut[1];
  b->last_kind = 1;
  b->last_nrow = total;
  *nrow_out = total;
  return EPI_OK;
}

int epi_batch_cx_set_shared(epi_batch *b, const int64_t *h_keys, const int32_t *h_owned, int32_t nshared,
                            int32_t *d_slab) {
  if (!b || nshared < 0 || (nshared > 0 && (!h_keys || !h_owned || !d_slab)))
    return fail(EPI_ERR_ARG, "epi_batch_cx_set_shared: bad arguments");
  for (int32_t i = 1; i < nshared; i++)
    if (h_keys[i - 1] >= h_keys[i]) return fail(EPI_ERR_ARG, "epi_batch_cx_set_shared: keys must be strictly increasing");
  EPI_HIP(hipSetDevice(b->eng->device));
  b->shared_keys.assign(h_keys, h_keys + nshared);
  b->shared_owned.assign(h_owned, h_owned + nshared);
  // a sharded run attaches the same keys for every report: skip the copies when the device already holds them
  if (nshared > 0 && b->dev_shared_keys == b->shared_keys && b->dev_shared_owned == b->shared_owned) {
    b->d_slab = d_slab;
    b->d_mhl_cnt_slab = nullptr;
    b->d_mhl_sum_slab = nullptr;
    return EPI_OK;
  }
  b->d_slab = nshared > 0 ? d_slab : nullptr;
  b->d_mhl_cnt_slab = nullptr;
  b->d_mhl_sum_slab = nullptr;
  if (nshared > 0) {
    EPI_TRY(b->d_shared_keys.ensure((size_t)nshared * 8));
    EPI_TRY(b->d_shared_owned.ensure((size_t)nshared * 4));
    EPI_HIP(hipMemcpy(b->d_shared_keys.p, h_keys, (size_t)nshared * 8, hipMemcpyHostToDevice));
    EPI_HIP(hipMemcpy(b->d_shared_owned.p, h_owned, (size_t)nshared * 4, hipMemcpyHostToDevice));
    b->dev_shared_keys = b->shared_keys;
    b->dev_shared_owned = b->shared_owned;
  }
  return EPI_OK;
}

int epi_batch_tile_key_range(epi_batch *b, void *stream, int64_t *first_key, int64_t *last_key) {
  return epi_batch_tile_key_range_for(b, cx_tile_for(1), stream, first_key, last_key);
}

int epi_batch_tile_key_range_for(epi_batch *b, int tile_positions, void *stream, int64_t *first_key, int64_t *last_key) {
  if (!b || !first_key || !last_key) return fail(EPI_ERR_ARG, "epi_batch_tile_key_range: NULL argument");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  *first_key = 0; *last_key = -1;              // empty range
  const int T = tile_positions;
  RowStats st;
  int32_t nt = 0;
  EPI_TRY(build_tiles(b, s, T, &st, &nt));
  if (nt == 0) return EPI_OK;
  Tile t0, t1;
  EPI_TRY(read_scalars(b, s, b->tiles.as<Tile>(), sizeof(Tile), &t0));
  EPI_TRY(read_scalars(b, s, b->tiles.as<Tile>() + (nt - 1), sizeof(Tile), &t1));
  auto key = [T](const Tile &t) { return ((int64_t)t.rname << 32) | (int64_t)(uint32_t)((t.pos0 + kPosBias) / T); };
  *first_key = key(t0);
  *last_key = key(t1);
  return EPI_OK;
}

int epi_batch_cx_fetch_dev(epi_batch *b, int32_t *const d_cols[6], void *stream) {
  if (!b || !d_cols) return fail(EPI_ERR_ARG, "epi_batch_cx_fetch_dev: NULL argument");
  if (b->last_kind != 1) return fail(EPI_ERR_STATE, "epi_batch_cx_fetch_dev: no finished CX report on this batch");
  if (b->last_nrow == 0) return EPI_OK;
  for (int i = 0; i < 6; i++) if (!d_cols[i]) return fail(EPI_ERR_ARG, "epi_batch_cx_fetch_dev: NULL column");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  const unsigned nb = (unsigned)((b->last_ntiles + 3) / 4);
  prof_begin("gather", s);
  hipLaunchKernelGGL(k_cx_gather, dim3(nb), dim3(256), 0, s, b->tiles.as<Tile>(), b->tile_out.as<uint32_t>(),
                     b->tile_nrow.as<uint32_t>(), b->tile_base.as<uint32_t>(), b->last_ntiles, b->pool_key.as<uint32_t>(),
                     b->pool_a.as<uint32_t>(), b->pool_b.as<uint32_t>(), d_cols[0], d_cols[1], d_cols[2], d_cols[3],
                     d_cols[4], d_cols[5]);
  prof_end("gather", s);
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}

int epi_batch_cx_fetch_host(epi_batch *b, int32_t *const h_cols[6], void *stream) {
  if (!b || !h_cols) return fail(EPI_ERR_ARG, "epi_batch_cx_fetch_host: NULL argument");
  if (b->last_kind != 1) return fail(EPI_ERR_STATE, "epi_batch_cx_fetch_host: no finished CX report on this batch");
  const int64_t nrow = b->last_nrow;
  if (nrow == 0) return EPI_OK;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  EPI_TRY(b->pool_c.ensure((size_t)nrow * 4 * 6));
  int32_t *d = b->pool_c.as<int32_t>();
  int32_t *cols[6];
  for (int i = 0; i < 6; i++) cols[i] = d + (int64_t)i * nrow;
  EPI_TRY(epi_batch_cx_fetch_dev(b, cols, s));
  CopyPart parts[6];
  for (int i = 0; i < 6; i++) parts[i] = {h_cols[i], cols[i], (size_t)nrow * 4};
  return copy_parts_to_host(b->eng, parts, 6, s);
}

}  // extern "C"
