// rcpp_cx_report (src/rcpp_cx_report.cpp:34-159) on the GPU.
//
// The reference walks the sorted reads and, per base, emplaces
// pos -> int[32] into an ordered map, flushing the map through the
// majority-context rule (spit_results, :58-85) whenever a read starts beyond
// everything seen so far.  For sorted input the flush timing does not change
// the result: the output is the per-(rname,pos,strand) counter table pushed
// through the rule, in (rname, pos, '+' before '-') order.
//
// Here a workgroup owns one tile of T (default 1024) consecutive positions (tiles.hip).
// Its candidate rows are a contiguous row range; a group of G lanes takes a
// row, streams the slice of the row that falls inside the tile with coalesced
// dword loads (all in flight before the first is used) and adds every base into
// counters in LDS with ds_add_u32: two u16 counters per dword ([strand][4 pairs][T],
// 32 KiB; tiles with more than 32767 candidate rows never get here, they are "heavy"),
// or plain u32 [strand][8][T]; lanes of a group walk consecutive positions, so their
// atomics spread over the banks.
// Only eight counters per (pos,strand) are ever read by the rule: '.', H, h,
// X, x, Z, z and "everything else that counts toward coverage" (U/u and any
// other nibble; nibble 9 counts twice because the reference's coverage slot is
// slot 9, :126-127).  Coverage is their sum, so ONE LDS atomic per base.
// After a barrier the same workgroup applies the rule -- candidates first (cells
// where a reported context has any count, listed per wavefront by ballot ranks),
// then the rule on the listed cells -- and writes (key, meth, unmeth) in position
// order into the tile's own slot of the row pool (no atomic; a tile with more rows
// than a slot takes them from an overflow region).  A scan over per-tile row counts
// then gives every tile its place in the final, ordered table (k_cx_gather).
#include "common.hpp"
#include "tile_common.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace epi {

constexpr int CX_WG = 512;                    // threads per tile workgroup (8 wavefronts)

struct CxArgs {
  RowCols c;
  const Tile *tiles;
  uint32_t ctx_mask;                      // bit k set: context k (2,6,7) is reported
  uint32_t *pool_key, *pool_meth, *pool_unmeth;
  uint32_t pool_cap;
  uint32_t *cursor;                       // rows handed out of the overflow region so far (may exceed its size)
  uint32_t slot_rows, ovf_base;           // tile t owns pool rows [t * slot_rows, +slot_rows); larger tiles take
                                          // rows from [ovf_base, pool_cap) through the cursor
  uint32_t *tile_nrow, *tile_base;
  int32_t *slab;                          // shared-tile counters [slot][16][T]
  int ablate;                             // timing experiments only (EPIHIP_CX_ABLATE): 1 skip accumulate, 2 skip emit, 4 loads only
  unsigned long long *diag;               // timing experiments only (EPIHIP_CX_DIAG): per-phase cycle sums of wave 0
  // ultra-deep tiles (amplicon pile-ups) are set aside by k_cx_tiles and split over many workgroups
  int heavy_rows;                         // a tile with more candidate rows than this is "heavy"
  int heavy_chunk;                        // rows per work item of k_cx_heavy
  uint32_t *heavy_count, *heavy_max;      // number of heavy tiles, largest candidate-row count among them
  uint32_t *heavy_list;                   // their tile ids (order of discovery)
  int32_t *heavy_slab;                    // [heavy tile][16][T] counters summed over the work items
};

// Adds the in-tile slices of the candidate rows into the LDS counters.  G lanes own one row
// (64/G rows per wavefront step); a lane keeps CX_NU dword loads of its row in flight and the next
// step's row columns are fetched with them.  (Deeper software pipelines -- bytes one step ahead and
// columns three on the u32 layout, later two or three whole steps in flight through unconditional
// buffer loads -- ran no faster: DESIGN.md 4.3.)
// dwords u = U0..U1-1 of a lane's row slice (dword index sub + u*G), all already loaded
template <int T, int G, int U0, int U1, bool PK>
__device__ __forceinline__ void cx_add_range(const uint32_t (&w)[CX_NU], int sub, const RowSlice &cur) {
  if constexpr (U0 < U1) {
    if (U0 * G <= cur.tl) cx_add_dword<T, 4 * G * U0, U0 == 0, PK>(w[U0], cur.tl == U0 * G, cur);
    cx_add_range<T, G, U0 + 1, U1, PK>(w, sub, cur);
  }
}

template <int T, int G, int WG, bool PK>
__device__ __forceinline__ void cx_accumulate(const CxArgs &a, const Tile &td, uint32_t *cnt) {
  constexpr int R = 64 / G;
  constexpr int NW = WG / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1), grp = lane / G;
  int r = td.row_lo + wave * R + grp;
  RowSlice cur = cx_row_slice<T, G, PK>(a.c, td, r, sub, cnt);
  for (int rbase = td.row_lo + wave * R; rbase < td.row_hi; rbase += NW * R) {
    uint32_t w[CX_NU];                                    // every load of the slice is in flight before the first is used
#pragma unroll
    for (int u = 0; u < CX_NU; u++) {
      w[u] = u * G <= cur.tl ? cur.src[u * G] : 0u;
    }
    r += NW * R;
    const RowSlice nxt = cx_row_slice<T, G, PK>(a.c, td, r, sub, cnt);
    if (a.ablate & 4) {                                   // timing experiment: the loads without the counting
      uint32_t x = 0;
#pragma unroll
      for (int u = 0; u < CX_NU; u++) x ^= w[u];
      if (x == 0x12345678u) atomicAdd(cnt, 1u);
      cur = nxt;
      continue;
    }
    cx_add_range<T, G, 0, CX_NU, PK>(w, sub, cur);
    for (int k = sub + CX_NU * G; k < cur.nd; k += G) {   // slices longer than CX_NU*G dwords (EPIHIP_CX_GROUP overrides)
      RowSlice t = cur;
#pragma unroll
      for (int j = 0; j < 4; j++) t.dst[j] = cur.dst[j] + 4 * (k - sub);
      cx_add_dword<T, 0, false, PK>(cur.src[k - sub], k == cur.nd - 1, t);
    }
    cur = nxt;
  }
}

// The reference's majority rule on one (pos,strand): returns context 2/6/7 or 0 (no row).
__device__ __forceinline__ int cx_rule(const uint32_t c[8], uint32_t ctx_mask, uint32_t *meth, uint32_t *unmeth) {
  const uint32_t nH = c[SLOT_H] + c[SLOT_h], nX = c[SLOT_X] + c[SLOT_x], nZ = c[SLOT_Z] + c[SLOT_z];
  const uint32_t cov = c[SLOT_DOT] + c[SLOT_OTHER] + nH + nX + nZ;
  if (cov == 0) return 0;                                 // :62
  const uint32_t half = cov >> 1;                         // :63
  int k;
  if (c[SLOT_DOT] > half) return 0;                       // :64
  else if (nH > half) { k = 2; *meth = c[SLOT_H]; *unmeth = c[SLOT_h]; }   // :65
  else if (nX > half) { k = 6; *meth = c[SLOT_X]; *unmeth = c[SLOT_x]; }   // :67
  else if (nZ > half) { k = 7; *meth = c[SLOT_Z]; *unmeth = c[SLOT_z]; }   // :69
  else return 0;                                          // :71
  return ((ctx_mask >> k) & 1u) ? k : 0;                  // :72
}

// Pool rows for a tile's n output rows.  Every tile has its own slot of slot_rows rows, so the common case needs no
// atomic: one cursor for all tiles is ~10^5 atomics on one address, served one by one at ~8 ns each, and a workgroup
// waited on its turn with 32 KiB of LDS in hand (0.13 ms of the 1.22 ms kernel on 10 M templates; with 512-position
// tiles the cursor alone set the kernel time).  Only tiles with more rows than a slot go to the cursor.
__device__ __forceinline__ uint32_t cx_pool_reserve(const CxArgs &a, int tile, uint32_t n, bool *fits) {
  if (n <= a.slot_rows) { *fits = true; return (uint32_t)tile * a.slot_rows; }
  const uint32_t o = atomicAdd(a.cursor, n);
  *fits = (uint64_t)a.ovf_base + o + n <= a.pool_cap;
  return a.ovf_base + o;
}

// Rule + ordered compaction of one tile's u32 counters [16][T] (LDS, or staged from a slab) into the row pool.
template <int T, int WG>
__device__ __forceinline__ void cx_emit(const CxArgs &a, int tile, const uint32_t *cnt, uint32_t *s_scan) {
  constexpr int PPT = T / WG;                          // consecutive positions per thread
  static_assert(PPT == 1 || PPT == 2 || PPT == 4, "emit phase layout");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = WG / 64;
  const int p0 = threadIdx.x * PPT;
  uint32_t key[2 * PPT], me[2 * PPT], un[2 * PPT];       // statically indexed (fully unrolled): stay in VGPRs
  bool ok[2 * PPT];
  int nr = 0;
#pragma unroll
  for (int q = 0; q < PPT; q++) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      uint32_t c[8];
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = cnt[(s * 8 + k) * T + p0 + q];
      uint32_t m = 0, u = 0;
      const int ctx = cx_rule(c, a.ctx_mask, &m, &u);
      ok[q * 2 + s] = ctx != 0;
      key[q * 2 + s] = ((uint32_t)(p0 + q) << 4) | ((uint32_t)s << 3) | (uint32_t)ctx;
      me[q * 2 + s] = m;
      un[q * 2 + s] = u;
      nr += ctx != 0;
    }
  }
  // block-wide exclusive scan of nr
  const uint32_t inc = wave_scan_u32((uint32_t)nr);
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
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
  const uint32_t ex = inc - (uint32_t)nr + s_scan[wave];
  if (total != 0xFFFFFFFFu) {
    uint32_t w = base + ex;
#pragma unroll
    for (int i = 0; i < 2 * PPT; i++) {
      if (ok[i]) {
        a.pool_key[w] = key[i];
        a.pool_meth[w] = me[i];
        a.pool_unmeth[w] = un[i];
        w++;
      }
    }
  }
}

// Emit for the packed counters.  A wavefront owns T / (WG/64) consecutive positions and walks them in blocks of 32:
// lanes 0-31 look at the '+' strand of the block, lanes 32-63 at the '-' strand (two conflict-free half-wave reads).
// Pass 1 only asks whether any reported context has a count at all -- a row needs n_k > cov/2 >= 0 -- and writes
// the (few) candidates, in key order, to a per-wave list; pass 2 reads the candidates densely, one per lane, and
// applies the rule.  Ranks come from ballots and popcounts (no shuffle scan); a CG report evaluates the rule for
// ~4 % of the (pos,strand) cells instead of all of them, a CX report for ~25 %.
template <int T, int WG>
__device__ __forceinline__ void cx_emit_packed(const CxArgs &a, int tile, const uint32_t *cnt, uint32_t *s_scan,
                                               uint16_t *s_list) {
  constexpr int NW = WG / 64, PW = T / NW, IT = PW / 32;
  static_assert(PW % 32 == 0 && IT >= 1 && IT <= 16, "emit phase layout");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l5 = lane & 31, sd = lane >> 5;
  const uint32_t below = (1u << l5) - 1u;
  uint16_t *list = s_list + wave * (2 * PW);
  const bool rH = (a.ctx_mask >> 2) & 1u, rX = (a.ctx_mask >> 6) & 1u, rZ = (a.ctx_mask >> 7) & 1u;
  const uint32_t *cs = cnt + sd * 4 * T + wave * PW + l5;
  uint32_t nc = 0;                                        // candidates of this wavefront (uniform)
#pragma unroll
  for (int i = 0; i < IT; i++) {
    uint32_t any = 0;
    if (rH) any |= cs[1 * T + i * 32];
    if (rX) any |= cs[2 * T + i * 32];
    if (rZ) any |= cs[3 * T + i * 32];
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
        const uint32_t *c0 = cnt + st * 4 * T + pos;
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t w = c0[k * T]; c[2 * k] = w & 0xFFFFu; c[2 * k + 1] = w >> 16; }
        uint32_t m = 0, u = 0;
        const int ctx = cx_rule(c, a.ctx_mask, &m, &u);
        good = ctx != 0;
        key[jj] = ((uint32_t)pos << 4) | ((uint32_t)st << 3) | (uint32_t)ctx;
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
  if (threadIdx.x == 0) {
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
        a.pool_key[w] = key[jj];
        a.pool_meth[w] = me[jj];
        a.pool_unmeth[w] = un[jj];
      }
    }
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

// As many workgroups per CU as LDS and the 2048-thread limit allow (default: packed counters, 32 KiB, four
// 512-thread workgroups); always 8 waves per SIMD, i.e. a VGPR budget of 64.
template <int T, int G, int WG, bool PK>
__global__ __launch_bounds__(WG, (cx_waves_per_simd<T, WG, PK>())) void k_cx_tiles(CxArgs a, int ntiles) {
  constexpr int NLDS = cx_lds_dwords<T, PK>() + 2 * kCxGuard;
  __shared__ __attribute__((aligned(16))) uint32_t cnt_raw[NLDS];
  __shared__ uint32_t s_scan[WG / 64 + 2];
  __shared__ uint16_t s_list[PK ? 2 * T : 2];              // candidate lists of the packed emit, 2 * T / (WG/64) per wavefront
  uint32_t *cnt = cnt_raw + kCxGuard;
  const int tile = cx_tile_of_block(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  if (a.diag) t0 = __builtin_amdgcn_s_memtime();
  const Tile td = a.tiles[tile];                          // (in flight while the counters are cleared)
  uint4 *z = reinterpret_cast<uint4 *>(cnt_raw);
  for (int i = threadIdx.x; i < NLDS / 4; i += WG) z[i] = make_uint4(0, 0, 0, 0);
  if (td.row_hi - td.row_lo > a.heavy_rows) {
    // one workgroup would crawl through this pile-up alone: k_cx_heavy splits it by row chunks instead
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
  if (a.diag) t1 = __builtin_amdgcn_s_memtime();
  if (!(a.ablate & 1)) cx_accumulate<T, G, WG, PK>(a, td, cnt);
  if (a.diag) t2 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (a.diag) t3 = __builtin_amdgcn_s_memtime();
  if (td.slot >= 0) {
    // shared with another rank (or split over several work items): hand the raw counters over
    cx_dump_slab<T, WG, PK>(cnt, a.slab + (int64_t)td.slot * (kCxPlanes * T));
    if (threadIdx.x == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
    return;
  }
  if (a.ablate & 2) { if (threadIdx.x == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; } return; }
  if constexpr (PK) cx_emit_packed<T, WG>(a, tile, cnt, s_scan, s_list);
  else cx_emit<T, WG>(a, tile, cnt, s_scan);
  if (a.diag && (threadIdx.x & 63) == 0) {      // diagnostic build only: where a wavefront's tile time goes
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    const int w = threadIdx.x >> 6;
    if (w == 0 || w == WG / 64 - 1) {
      unsigned long long *d = a.diag + (w == 0 ? 0 : 8);
      atomicAdd(d + 0, t1 - t0); atomicAdd(d + 1, t2 - t1); atomicAdd(d + 2, t3 - t2); atomicAdd(d + 3, t4 - t3);
      atomicAdd(d + 4, 1ull);
    }
  }
}

// One chunk of the candidate rows of one heavy tile: LDS histogram as usual, then added into the tile's
// dense counter slab in HBM (or straight into its shared slab slot when other ranks contribute too).
template <int T, int G, int WG, bool PK>
__global__ __launch_bounds__(WG, (cx_waves_per_simd<T, WG, PK>())) void k_cx_heavy(CxArgs a) {
  constexpr int NLDS = cx_lds_dwords<T, PK>() + 2 * kCxGuard;
  __shared__ __attribute__((aligned(16))) uint32_t cnt_raw[NLDS];
  uint32_t *cnt = cnt_raw + kCxGuard;
  const int tile = (int)a.heavy_list[blockIdx.y];
  Tile td = a.tiles[tile];
  const int lo = td.row_lo + (int)blockIdx.x * a.heavy_chunk;
  if (lo >= td.row_hi) return;
  td.row_lo = lo;
  if (td.row_hi - lo > a.heavy_chunk) td.row_hi = lo + a.heavy_chunk;
  uint4 *z = reinterpret_cast<uint4 *>(cnt_raw);
  for (int i = threadIdx.x; i < NLDS / 4; i += WG) z[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  cx_accumulate<T, G, WG, PK>(a, td, cnt);
  __syncthreads();
  cx_dump_slab<T, WG, PK>(cnt, td.slot >= 0 ? a.slab + (int64_t)td.slot * (kCxPlanes * T)
                                            : a.heavy_slab + (int64_t)blockIdx.y * (kCxPlanes * T));
}

// Majority rule + rows for the heavy tiles that are not shared with other ranks.
template <int T>
__global__ __launch_bounds__(CX_WG) void k_cx_emit_heavy(CxArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t cnt[kCxPlanes * T];
  __shared__ uint32_t s_scan[CX_WG / 64 + 2];
  const int tile = (int)a.heavy_list[blockIdx.x];
  const Tile td = a.tiles[tile];
  if (td.slot >= 0) return;                                // emitted after the cross-rank reduce
  const int32_t *src = a.heavy_slab + (int64_t)blockIdx.x * (kCxPlanes * T);
  for (int i = threadIdx.x; i < kCxPlanes * T; i += CX_WG) cnt[i] = (uint32_t)src[i];
  __syncthreads();
  cx_emit<T, CX_WG>(a, tile, cnt, s_scan);
}

// Emits the shared tiles this rank owns from the (already cross-rank reduced) slab.
template <int T>
__global__ __launch_bounds__(CX_WG) void k_cx_emit_slab(CxArgs a, const int32_t *__restrict__ owned,
                                                        const int32_t *__restrict__ slot_tile) {
  __shared__ __attribute__((aligned(16))) uint32_t cnt[kCxPlanes * T];
  __shared__ uint32_t s_scan[CX_WG / 64 + 2];
  if (!owned[blockIdx.x]) return;                          // one workgroup per shared slot
  const int tile = slot_tile[blockIdx.x];
  if (tile < 0) return;
  const Tile td = a.tiles[tile];
  const int32_t *src = a.slab + (int64_t)td.slot * (kCxPlanes * T);
  for (int i = threadIdx.x; i < kCxPlanes * T; i += CX_WG) cnt[i] = (uint32_t)src[i];
  __syncthreads();
  cx_emit<T, CX_WG>(a, tile, cnt, s_scan);
}

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
  for (uint32_t i = lane; i < n; i += 64) {
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

// lanes per row: enough that CX_NU dwords per lane cover the longest in-tile slice
static int pick_cx_group(int32_t max_len, int T) {
  const char *env = getenv("EPIHIP_CX_GROUP");
  if (env) { int g = atoi(env); if (g == 8 || g == 16 || g == 32 || g == 64) return g; }
  const int slice = (max_len < T ? max_len : T) + 3;
  const int nd = (slice + 3) / 4;
  int g = 8;
  while (g < 64 && g * CX_NU < nd) g <<= 1;   // the whole slice in one round of CX_NU loads per lane
  return g;
}

// Counter layout: packed u16 pairs (32 KiB of LDS per 1024-position tile: four 512-thread workgroups per CU) unless
// EPIHIP_CX_PACKED=0 (u32 counters, 64 KiB, two 1024-thread workgroups per CU: 1.52 vs 1.22 ms on config 2).
static bool cx_packed() {
  static int pk = -1;
  if (pk < 0) {
    pk = 1;
    if (const char *env = getenv("EPIHIP_CX_PACKED")) pk = atoi(env) != 0;
  }
  return pk != 0;
}

int cx_tile_positions() {
  static int t = 0;
  if (!t) {
    t = kTile;
    if (const char *env = getenv("EPIHIP_CX_TILE")) { const int v = atoi(env); if (v == 512 || v == 1024 || v == 2048) t = v; }
  }
  return t;
}

static int cx_workgroup_size() {
  static int wg = 0;
  if (!wg) {
    wg = cx_packed() ? 512 : 1024;              // 32 wavefronts per CU either way (4 x 8 or 2 x 16), measured fastest (profiles/)
    if (const char *env = getenv("EPIHIP_CX_WG")) { const int v = atoi(env); if (v == 256 || v == 512 || v == 1024) wg = v; }
  }
  return wg;
}

template <int T, int WG, bool PK>
static void launch_cx_tiles_g(int g, int nt, hipStream_t s, const CxArgs &a) {
  const unsigned grid = (unsigned)(((nt + 7) / 8) * 8);
  switch (g) {
    case 8: hipLaunchKernelGGL((k_cx_tiles<T, 8, WG, PK>), dim3(grid), dim3(WG), 0, s, a, nt); break;
    case 16: hipLaunchKernelGGL((k_cx_tiles<T, 16, WG, PK>), dim3(grid), dim3(WG), 0, s, a, nt); break;
    case 32: hipLaunchKernelGGL((k_cx_tiles<T, 32, WG, PK>), dim3(grid), dim3(WG), 0, s, a, nt); break;
    default: hipLaunchKernelGGL((k_cx_tiles<T, 64, WG, PK>), dim3(grid), dim3(WG), 0, s, a, nt); break;
  }
}

template <int T, int WG, bool PK>
static void launch_cx_heavy_g(int g, dim3 grid, hipStream_t s, const CxArgs &a) {
  switch (g) {
    case 8: hipLaunchKernelGGL((k_cx_heavy<T, 8, WG, PK>), grid, dim3(WG), 0, s, a); break;
    case 16: hipLaunchKernelGGL((k_cx_heavy<T, 16, WG, PK>), grid, dim3(WG), 0, s, a); break;
    case 32: hipLaunchKernelGGL((k_cx_heavy<T, 32, WG, PK>), grid, dim3(WG), 0, s, a); break;
    default: hipLaunchKernelGGL((k_cx_heavy<T, 64, WG, PK>), grid, dim3(WG), 0, s, a); break;
  }
}

// (T, WG, layout) combinations that are built: the defaults (2048/1024/packed, 1024/1024/u32) and the ones the
// EPIHIP_CX_* experiment switches reach.  heavy = false: k_cx_tiles over nt tiles; true: k_cx_heavy on `grid` + emit.
template <int T, int WG, bool PK>
static void launch_cx_variant(bool heavy, int g, int nt, dim3 grid, hipStream_t s, const CxArgs &a) {
  if (!heavy) { launch_cx_tiles_g<T, WG, PK>(g, nt, s, a); return; }
  launch_cx_heavy_g<T, WG, PK>(g, grid, s, a);
  hipLaunchKernelGGL((k_cx_emit_heavy<T>), dim3(grid.y), dim3(CX_WG), 0, s, a);
}

static void launch_cx(bool heavy, int T, int g, int nt, dim3 grid, hipStream_t s, const CxArgs &a) {
  const bool big = cx_workgroup_size() == 1024, pk = cx_packed();
  if (pk && cx_workgroup_size() == 256) {
    if (T == 512) launch_cx_variant<512, 256, true>(heavy, g, nt, grid, s, a); else launch_cx_variant<1024, 256, true>(heavy, g, nt, grid, s, a);
    return;
  }
  if (T == 512) { if (pk) launch_cx_variant<512, 512, true>(heavy, g, nt, grid, s, a); else launch_cx_variant<512, 512, false>(heavy, g, nt, grid, s, a); }
  else if (T == 2048) {
    if (pk) { if (big) launch_cx_variant<2048, 1024, true>(heavy, g, nt, grid, s, a); else launch_cx_variant<2048, 512, true>(heavy, g, nt, grid, s, a); }
    else launch_cx_variant<2048, 1024, false>(heavy, g, nt, grid, s, a);
  } else {
    if (pk) { if (big) launch_cx_variant<1024, 1024, true>(heavy, g, nt, grid, s, a); else launch_cx_variant<1024, 512, true>(heavy, g, nt, grid, s, a); }
    else { if (big) launch_cx_variant<1024, 1024, false>(heavy, g, nt, grid, s, a); else launch_cx_variant<1024, 512, false>(heavy, g, nt, grid, s, a); }
  }
}

static void launch_cx_emit_slab(int T, int nshared, hipStream_t s, const CxArgs &a, const int32_t *owned, const int32_t *slot_tile) {
  if (T == 512) hipLaunchKernelGGL((k_cx_emit_slab<512>), dim3((unsigned)nshared), dim3(CX_WG), 0, s, a, owned, slot_tile);
  else if (T == 2048) hipLaunchKernelGGL((k_cx_emit_slab<2048>), dim3((unsigned)nshared), dim3(CX_WG), 0, s, a, owned, slot_tile);
  else hipLaunchKernelGGL((k_cx_emit_slab<1024>), dim3((unsigned)nshared), dim3(CX_WG), 0, s, a, owned, slot_tile);
}

static int ensure_pool(epi_batch *b, size_t rows) {
  if (rows <= b->pool_cap && b->pool_key.p) return EPI_OK;
  EPI_TRY(b->pool_key.ensure(rows * 4));
  EPI_TRY(b->pool_a.ensure(rows * 4));
  EPI_TRY(b->pool_b.ensure(rows * 4));
  b->pool_cap = rows;
  return EPI_OK;
}

}  // namespace epi

using namespace epi;

extern "C" {

int epi_tile_positions(void) { return cx_tile_positions(); }

int epi_batch_cx_report_dev(epi_batch *b, const int32_t *d_pass, const char *ctx, void *stream, int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_cx_report_dev: NULL argument");
  *nrow_out = 0;
  b->last_kind = 0;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);

  uint32_t ctx_mask = 0;                                   // rcpp_cx_report.cpp:88-91
  for (const unsigned char *c = reinterpret_cast<const unsigned char *>(ctx); *c; c++) ctx_mask |= 1u << ctx_to_idx(*c);

  const int T = cx_tile_positions();
  RowStats st;
  int32_t nt = 0;
  EPI_TRY(build_tiles(b, s, T, &st, &nt));
  b->last_ntiles = nt;
  if (nt == 0) { b->last_kind = 1; b->last_nrow = 0; return EPI_OK; }

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
  if (const char *env = getenv("EPIHIP_CX_SLOT")) { const int v = atoi(env); if (v >= 0 && v <= 2 * T) slot = (uint32_t)v; }
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
  const int grp = pick_cx_group(st.max_len, T);
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;           // misc[1] = pool cursor, misc[2] = nrow total

  CxArgs a;
  a.c.xm = b->xm; a.c.off = b->off; a.c.start = b->start; a.c.strand = b->strand; a.c.pass = d_pass;
  a.tiles = b->tiles.as<Tile>();
  a.ctx_mask = ctx_mask;
  a.cursor = cursor;
  a.tile_nrow = b->tile_nrow.as<uint32_t>();
  a.tile_base = b->tile_base.as<uint32_t>();
  a.slab = b->d_slab;
  a.ablate = 0;
  if (const char *env = getenv("EPIHIP_CX_ABLATE")) a.ablate = atoi(env);
  a.diag = nullptr;
  if (getenv("EPIHIP_CX_DIAG")) {
    EPI_TRY(b->diag.ensure(256));
    a.diag = b->diag.as<unsigned long long>();
    EPI_HIP(hipMemsetAsync(a.diag, 0, 128, s));
  }
  a.heavy_rows = 16384;
  if (const char *env = getenv("EPIHIP_HEAVY_ROWS")) { const int v = atoi(env); if (v > 0) a.heavy_rows = v; }
  if (a.heavy_rows > 32767) a.heavy_rows = 32767;          // packed u16 counters: a base adds at most 2
  a.heavy_chunk = a.heavy_rows / 4 > 64 ? a.heavy_rows / 4 : 64;
  EPI_TRY(b->heavy_list.ensure((size_t)nt * 4));
  a.heavy_list = b->heavy_list.as<uint32_t>();
  a.heavy_count = b->misc.as<uint32_t>() + 3;             // misc[3] = heavy tiles, misc[8] = their largest row count
  a.heavy_max = b->misc.as<uint32_t>() + 8;
  a.heavy_slab = nullptr;
  a.slot_rows = slot;
  a.ovf_base = (uint32_t)ovf_base;
  b->cx_last_slot = slot;
  b->cx_last_ovf = (uint32_t)ovf_base;
  uint32_t used_total[2] = {0, 0};
  for (int attempt = 0; attempt < 2; attempt++) {
    a.pool_key = b->pool_key.as<uint32_t>();
    a.pool_meth = b->pool_a.as<uint32_t>();
    a.pool_unmeth = b->pool_b.as<uint32_t>();
    a.pool_cap = (uint32_t)(b->pool_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : b->pool_cap);
    if (attempt > 0) {                                   // (the tile-index pass zeroed them for the first attempt)
      EPI_HIP(hipMemsetAsync(cursor, 0, 12, s));         // cursor, total, heavy count
      EPI_HIP(hipMemsetAsync(a.heavy_max, 0, 4, s));
    }
    prof_begin("cx_tiles", s);
    launch_cx(false, T, grp, nt, dim3(1), s, a);
    prof_end("cx_tiles", s);
    EPI_HIP(hipGetLastError());
    // row offsets of the tiles are queued right away; {rows handed out, total rows, heavy tiles} come back in one sync
    EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
    uint32_t host[8];
    EPI_TRY(read_scalars(b, s, cursor, 32, host));         // misc[1..8]
    if (host[2] > 0) {
      // ultra-deep tiles were set aside: split each over ceil(rows/chunk) workgroups, reduce in HBM, emit, rescan
      const uint32_t nheavy = host[2], nchunks = (host[7] + (uint32_t)a.heavy_chunk - 1) / (uint32_t)a.heavy_chunk;
      EPI_TRY(b->heavy_slab.ensure((size_t)nheavy * kCxPlanes * T * 4));
      a.heavy_slab = b->heavy_slab.as<int32_t>();
      EPI_HIP(hipMemsetAsync(a.heavy_slab, 0, (size_t)nheavy * kCxPlanes * T * 4, s));
      prof_begin("cx_heavy", s);
      launch_cx(true, T, grp, nt, dim3(nchunks, nheavy), s, a);
      prof_end("cx_heavy", s);
      EPI_HIP(hipGetLastError());
      EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
      EPI_TRY(read_scalars(b, s, cursor, 8, host));
    }
    used_total[0] = host[0];
    used_total[1] = host[1];
    if (ovf_base + used_total[0] + headroom <= a.pool_cap) break;
    if (attempt == 1) return fail(EPI_ERR_STATE, "row pool overflow after regrow");
    EPI_TRY(ensure_pool(b, ovf_base + used_total[0] + (used_total[0] >> 4) + 1024 + headroom));   // exact need is known now: rerun once
    if (nshared > 0)   // the rerun adds into the slab again
      EPI_HIP(hipMemsetAsync(b->d_slab, 0, (size_t)nshared * kCxPlanes * T * 4, s));
  }
  if (a.diag) {
    unsigned long long h[16];
    EPI_HIP(hipMemcpy(h, a.diag, 128, hipMemcpyDeviceToHost));
    for (int k = 0; k < 2; k++)
      if (h[8 * k + 4])
        fprintf(stderr, "[cx diag wave %s] tiles %llu  cycles/tile: head+clear %.0f  accumulate %.0f  barrier %.0f  emit %.0f\n",
                k ? "last" : "0", h[8 * k + 4], (double)h[8 * k] / h[8 * k + 4], (double)h[8 * k + 1] / h[8 * k + 4],
                (double)h[8 * k + 2] / h[8 * k + 4], (double)h[8 * k + 3] / h[8 * k + 4]);
  }
  if (used_total[0] > used_total[1] / 8 && slot_state < (uint32_t)(2 * T)) slot_state *= 2;   // too many tiles outgrew their slot
  if (nshared > 0) { b->last_kind = 3; return EPI_OK; }     // caller continues with epi_batch_cx_finish_shared
  b->last_kind = 1;
  b->last_nrow = used_total[1];
  *nrow_out = used_total[1];
  return EPI_OK;
}

// Second half of a sharded report: the slab has been sum-reduced across ranks;
// emit the shared tiles this rank owns, then order all rows.
int epi_batch_cx_finish_shared(epi_batch *b, const char *ctx, void *stream, int64_t *nrow_out) {
  if (!b || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_cx_finish_shared: NULL argument");
  if (b->last_kind != 3) return fail(EPI_ERR_STATE, "epi_batch_cx_finish_shared without a sharded epi_batch_cx_report_dev");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  const int32_t nt = b->last_ntiles;
  uint32_t ctx_mask = 0;
  for (const unsigned char *c = reinterpret_cast<const unsigned char *>(ctx); *c; c++) ctx_mask |= 1u << ctx_to_idx(*c);
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;
  CxArgs a;
  memset(&a, 0, sizeof(a));
  a.tiles = b->tiles.as<Tile>();
  a.ctx_mask = ctx_mask;
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
  launch_cx_emit_slab(cx_tile_positions(), (int)b->shared_keys.size(), s, a, b->d_shared_owned.as<int32_t>(),
                      b->d_slot_tile.as<int32_t>());
  EPI_HIP(hipGetLastError());
  uint32_t *d_total = b->misc.as<uint32_t>() + 2;
  EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, d_total, b->scan_tmp, s));
  uint32_t ut[2] = {0, 0};                                  // {overflow rows handed out, total rows}: one sync
  EPI_TRY(read_scalars(b, s, cursor, 8, ut));
  // cannot overflow: epi_batch_cx_report_dev kept 2*kTile rows per shared tile free
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
  return epi_batch_tile_key_range_for(b, cx_tile_positions(), stream, first_key, last_key);
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
  hipLaunchKernelGGL(k_cx_gather, dim3(nb), dim3(256), 0, s, b->tiles.as<Tile>(), b->tile_out.as<uint32_t>(),
                     b->tile_nrow.as<uint32_t>(), b->tile_base.as<uint32_t>(), b->last_ntiles, b->pool_key.as<uint32_t>(),
                     b->pool_a.as<uint32_t>(), b->pool_b.as<uint32_t>(), d_cols[0], d_cols[1], d_cols[2], d_cols[3],
                     d_cols[4], d_cols[5]);
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
  for (int i = 0; i < 6; i++)
    EPI_HIP(hipMemcpyAsync(h_cols[i], cols[i], (size_t)nrow * 4, hipMemcpyDeviceToHost, s));
  EPI_HIP(hipStreamSynchronize(s));
  return EPI_OK;
}

}  // extern "C"
