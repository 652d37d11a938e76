// Tile index over the sorted rows.
//
// Positions are cut into tiles of T positions (kTile for CX, kMhlTile for lMHL) on an ABSOLUTE grid (so every GPU of a
// sharded run agrees on tile boundaries).  A tile (rname, t) exists iff some
// row of that rname could reach it: tile(start) <= t <= tile(start + Lmax - 1).
// Because rows are sorted by (rname,start), row x only has to create the tiles
// its predecessor did not already reach -- a purely local count -- and one
// exclusive scan over the rows turns the counts into tile slots; a tile's last
// candidate row is found the same way (row_tiles below).  This replaces
// the reference's "flush the map when start > max_pos" windowing
// (src/rcpp_cx_report.cpp:113) and the key sort of a sort+segmented-reduce
// scheme: the input order already is the sort.
#include "common.hpp"
#include <stdlib.h>
#include <string.h>

namespace epi {

// T is a power of two: tile index by shift (sh = log2 T)
__device__ __forceinline__ int64_t tile_of(int64_t pos, int32_t sh) { return (pos + kPosBias) >> sh; }

__global__ __launch_bounds__(256) void k_row_stats(const int32_t *__restrict__ start, const int32_t *__restrict__ rname,
                                                    const int32_t *__restrict__ strand, const int64_t *__restrict__ off,
                                                    int64_t n, RowStats *__restrict__ st, int32_t *__restrict__ len_out) {
  __shared__ uint32_t s_hist[kLenBinCount];
  if (threadIdx.x < kLenBinCount) s_hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int len = 0, unsorted = 0, bad_strand = 0, bad_len = 0, deep = 0;
  if (x < n) {
    const int64_t l = off[x + 1] - off[x];                  // (the constructors take rows back to back)
    const int32_t s0 = start[x];
    if (l < 0 || (int64_t)s0 + l > 0x7FFFFFFFLL) bad_len = 1; else len = (int)l;
    len_out[x] = len;                                       // the row lengths every later kernel reads (0 for a bad row: the first report raises the error)
    const int32_t sd = strand[x];
    if (sd != 1 && sd != 2 && l != 0) bad_strand = 1;       // (an empty row may carry strand 0: the placeholder template the
                                                            // reference pushes for a paired-end file without a usable pair,
                                                            // src/rcpp_read_bam.cpp:155; it has no bases to count)
    if (x > 0) {
      const int32_t r0 = rname[x - 1], r1 = rname[x];
      if (r1 < r0 || (r1 == r0 && s0 < start[x - 1])) unsorted = 1;
    }
    // Rows covering a position p are consecutive-ish in the sorted order: if x is the first of them, all of them start
    // before start[x] + len[x].  So when row x + 255 starts at or behind the end of row x (for every x), no position is
    // covered by more than 255 rows, and u8 counters per position cannot overflow (cx_report.hip, LEAN).
    if (len > 0 && x + 255 < n && rname[x + 255] == rname[x] && (int64_t)start[x + 255] < (int64_t)s0 + len) deep = 1;
    if (len > 0) {
      const int ch = (len + 30) >> 4;
      int k = 0;
      while (k < kLenBinCount - 1 && ch > kLenBins[k]) k++;
      atomicAdd(&s_hist[k], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < kLenBinCount && s_hist[threadIdx.x]) atomicAdd(&st->len_hist[threadIdx.x], s_hist[threadIdx.x]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    len = max(len, __shfl_xor(len, d, 64));
    unsorted |= __shfl_xor(unsorted, d, 64);
    bad_strand |= __shfl_xor(bad_strand, d, 64);
    bad_len |= __shfl_xor(bad_len, d, 64);
    deep |= __shfl_xor(deep, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    // uniform-length input: after the first few waves the cached maximum already covers `len`
    if (len > __hip_atomic_load(&st->max_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->max_len, len);   // read from L2
    if (unsorted) atomicOr(&st->unsorted, 1);
    if (bad_strand) atomicOr(&st->bad_strand, 1);
    if (bad_len) atomicOr(&st->bad_len, 1);
    if (deep) atomicOr(&st->deep, 1);
  }
}

// What row x contributes to the tile table, from its own and its predecessor's (rname, start) alone:
//  * the tiles it creates, (lo .. b]: those it can reach that its predecessor could not (see header comment);
//  * the tiles it closes, [hi_a .. hi_b]: row_hi of a tile (r,t) is the first row whose (rname, tile of start) lies
//    beyond (r,t).  If x is such a row for anything, it is so exactly for the tiles from its predecessor's start
//    tile up to the predecessor's reach (capped below x's own start tile within one rname) -- tiles the predecessor
//    itself reaches, so they exist, and they are the LAST tiles created before x: their slots count back from the
//    exclusive scan at x.  No search over rows is needed for either end of a tile's row range.
struct RowTiles {
  int64_t lo, b;          // creates tiles lo..b (b < lo: none)
  int64_t hi_a, hi_b, bp; // closes tiles hi_a..hi_b (hi_b < hi_a: none); bp = predecessor's last reachable tile
};

// (s, r) = start / rname of the row, (sp, rp) = of its predecessor (has_prev: the row is not the first of the table)
__device__ __forceinline__ RowTiles row_tiles(int64_t s, int32_t r, int64_t sp, int32_t rp, bool has_prev, int32_t lmax, int32_t sh) {
  RowTiles rt;
  const int64_t sx = tile_of(s, sh);
  rt.b = tile_of(s + lmax - 1, sh);
  rt.lo = sx;
  rt.hi_a = 0; rt.hi_b = -1; rt.bp = 0;
  if (has_prev) {
    const int64_t tp = tile_of(sp, sh);
    rt.bp = tile_of(sp + lmax - 1, sh);
    if (rp == r) {
      if (rt.bp + 1 > rt.lo) rt.lo = rt.bp + 1;
      if (sx != tp) { rt.hi_a = tp; rt.hi_b = rt.bp < sx - 1 ? rt.bp : sx - 1; }
    } else {
      rt.hi_a = tp; rt.hi_b = rt.bp;
    }
  }
  return rt;
}

// The tile table in two passes over (start, rname) with nothing stored per row in between: pass A (FILL = false)
// reduces the per-row tile counts of a block of TB_ROWS rows to one number; after a single-block scan of those
// (k_scan_bsums) pass B (FILL = true) recomputes the counts, scans them inside the block and writes the tiles.
// A thread owns TB_ITEMS CONSECUTIVE rows: four 16-byte loads bring their columns, a row's predecessor is the
// thread's previous row, and the scan in row order is a serial prefix inside the thread plus one wavefront scan of
// the thread totals (the first version strided the rows over the threads: 32 scalar loads and eight wavefront scans
// per thread, 51-67 us for 10 M rows against 30).
constexpr int TB_THREADS = 256, TB_ITEMS = 8, TB_ROWS = TB_THREADS * TB_ITEMS;

struct __attribute__((packed, aligned(4))) TileI4 { int32_t v[4]; };   // 16 bytes at int32 alignment (adopted columns may be views)

// VERIFY (with FILL): `bsum` is the scanned block-sum array remembered from an earlier call on this batch and tile
// size (the count pass, the scan and the host round trip are skipped); every block checks its own total against it and
// the table's capacity `cap` is the remembered tile count.  misc[0] (zeroed before the launch) ends up as that count,
// or as 0xFFFFFFFF when any block disagrees -- the rows were changed under the batch -- which the caller reports.
template <bool FILL, bool VERIFY = false>
__global__ __launch_bounds__(TB_THREADS) void k_tile_pass(const int32_t *__restrict__ start, const int32_t *__restrict__ rname,
                                                           int64_t n, int32_t lmax, int32_t sh, uint32_t *__restrict__ bsum,
                                                           Tile *__restrict__ tiles, const int64_t *__restrict__ shared_keys,
                                                           int32_t nshared, int32_t *__restrict__ slot_tile,
                                                           uint32_t *__restrict__ misc, int64_t cap) {
  // (cap = entries of `tiles`: the count may be one remembered from an earlier call -- a batch that was changed since
  //  must not write behind the table before the host notices)
  __shared__ uint32_t s_tot[TB_THREADS / 64];
  if (FILL && blockIdx.x == 0 && threadIdx.x == 0) {       // the report kernels' counters start at zero (saves three memsets)
    misc[1] = 0; misc[2] = 0; misc[3] = 0; misc[8] = 0;    // pool cursor, output rows, heavy tiles, largest heavy tile
    misc[4] = 0;                                           // tiles the lean CX kernel hands to the general one
    misc[5] = 0;                                           // slab slots the one-pass lMHL kernel has handed out
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t x0 = (int64_t)blockIdx.x * TB_ROWS + (int64_t)threadIdx.x * TB_ITEMS;
  int32_t st[TB_ITEMS + 1], rn[TB_ITEMS + 1];              // [0] = the predecessor of the thread's first row
#pragma unroll
  for (int i = 0; i <= TB_ITEMS; i++) { st[i] = 0; rn[i] = 0; }
  if (x0 < n) {
    if (x0 > 0) { st[0] = start[x0 - 1]; rn[0] = rname[x0 - 1]; }
    if (x0 + TB_ITEMS <= n) {
#pragma unroll
      for (int q = 0; q < TB_ITEMS / 4; q++) {
        const TileI4 a = *reinterpret_cast<const TileI4 *>(start + x0 + 4 * q), c = *reinterpret_cast<const TileI4 *>(rname + x0 + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; k++) { st[1 + 4 * q + k] = a.v[k]; rn[1 + 4 * q + k] = c.v[k]; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TB_ITEMS; i++) if (x0 + i < n) { st[1 + i] = start[x0 + i]; rn[1 + i] = rname[x0 + i]; }
    }
  }
  uint32_t c[TB_ITEMS], tot = 0;
#pragma unroll
  for (int i = 0; i < TB_ITEMS; i++) {
    c[i] = 0;
    if (x0 + i < n) {
      const RowTiles rt = row_tiles(st[1 + i], rn[1 + i], st[i], rn[i], x0 + i > 0, lmax, sh);
      if (rt.b >= rt.lo) c[i] = (uint32_t)(rt.b - rt.lo + 1);
    }
    tot += c[i];
  }
  const uint32_t inc = wave_scan_u32(tot);
  if (lane == 63) s_tot[wave] = inc;
  __syncthreads();
  uint32_t btot = 0, wbase = 0;
#pragma unroll
  for (int w = 0; w < TB_THREADS / 64; w++) { const uint32_t t = s_tot[w]; if (w < wave) wbase += t; btot += t; }
  if (!FILL) {
    if (threadIdx.x == 0) bsum[blockIdx.x] = btot;
    return;
  }
  if constexpr (VERIFY) {
    if (threadIdx.x == 0) {
      const uint32_t expect = (blockIdx.x + 1 < gridDim.x ? bsum[blockIdx.x + 1] : (uint32_t)cap) - bsum[blockIdx.x];
      if (btot != expect) atomicMax(misc, 0xFFFFFFFFu);
      else if (blockIdx.x == 0) atomicMax(misc, (uint32_t)cap);
    }
  }
  uint32_t before = bsum[blockIdx.x] + wbase + inc - tot;  // tiles created by rows before the thread's first row
  const int64_t T = 1LL << sh;
#pragma unroll
  for (int i = 0; i < TB_ITEMS; i++) {
    const int64_t x = x0 + i;
    if (x >= n) break;
    const RowTiles rt = row_tiles(st[1 + i], rn[1 + i], st[i], rn[i], x > 0, lmax, sh);
    for (int64_t t = rt.hi_a; t <= rt.hi_b; t++) {          // tiles this row closes
      const int64_t j = (int64_t)before - 1 - (rt.bp - t);
      if (j >= 0 && j < cap) tiles[j].row_hi = (int32_t)x;
    }
    if (c[i]) {
      const int32_t r = rn[1 + i];
      for (uint32_t k = 0; k < c[i]; k++) {                 // tiles this row creates: it IS their row_lo
        const int64_t t = rt.lo + k;                        // (the first row with start >= pos0 - lmax + 1)
        if ((int64_t)before + k >= cap) break;
        Tile *td = tiles + before + k;
        td->pos0 = t * T - kPosBias;
        td->rname = r;
        td->row_lo = (int32_t)x;
        int32_t slot = -1;
        if (nshared > 0) {
          const int64_t key = ((int64_t)r << 32) | (int64_t)(uint32_t)t;
          int32_t a = 0, z = nshared;
          while (a < z) { int32_t m = (a + z) >> 1; if (shared_keys[m] < key) a = m + 1; else z = m; }
          if (a < nshared && shared_keys[a] == key) { slot = a; slot_tile[a] = (int32_t)(before + k); }
        }
        td->slot = slot;
      }
    }
    if (x == n - 1) {                                       // the end of the table closes what the last row reaches
      const int64_t sx = tile_of(st[1 + i], sh);
      for (int64_t t = sx; t <= rt.b; t++) {
        const int64_t j = (int64_t)before + c[i] - 1 - (rt.b - t);
        if (j >= 0 && j < cap) tiles[j].row_hi = (int32_t)n;
      }
    }
    before += c[i];
  }
}

// what the index pass leaves in misc when the table itself is reused: the tile count and the report kernels' zeroed counters
__global__ void k_misc_reset(uint32_t *__restrict__ misc, const uint32_t *__restrict__ nt) {
  if (threadIdx.x == 0) {
    misc[0] = nt[0];
    misc[1] = 0; misc[2] = 0; misc[3] = 0; misc[4] = 0; misc[5] = 0; misc[8] = 0;
  }
}

static int log2_tile(int32_t T) {
  int sh = 0;
  while ((1 << sh) < T) sh++;
  return sh;
}

// Row statistics are a property of the (immutable) batch: queued once, when the batch is created.
int launch_row_stats(epi_batch *b, hipStream_t s) {
  EPI_TRY(b->stats.ensure(sizeof(RowStats)));
  EPI_TRY(b->own_len.ensure((size_t)b->n * 4 + 4));
  b->len = b->own_len.as<int32_t>();
  EPI_HIP(hipMemsetAsync(b->stats.p, 0, sizeof(RowStats), s));
  if (b->n > 0) {
    const unsigned nb = (unsigned)((b->n + 255) / 256);
    hipLaunchKernelGGL(k_row_stats, dim3(nb), dim3(256), 0, s, b->start, b->rname, b->strand, b->off, b->n, b->stats.as<RowStats>(),
                       b->own_len.as<int32_t>());
    EPI_HIP(hipGetLastError());
  }
  if (!b->stats_done) EPI_HIP(hipEventCreateWithFlags(&b->stats_done, hipEventDisableTiming));
  EPI_HIP(hipEventRecord(b->stats_done, s));
  b->stats_queued = true;
  return EPI_OK;
}

int fetch_row_stats(epi_batch *b, hipStream_t s) {
  if (b->stats_host) return EPI_OK;
  if (!b->stats_queued) EPI_TRY(launch_row_stats(b, s));
  EPI_HIP(hipEventSynchronize(b->stats_done));              // the kernel may sit on another stream than `s`
  EPI_TRY(read_scalars(b, s, b->stats.p, sizeof(RowStats), &b->h_stats));
  b->stats_host = true;
  return EPI_OK;
}

// Validated row statistics (errors for bad offsets / strands / unsorted rows) + the tile table for tiles of T
// positions.  One host synchronisation per call (the tile count); the statistics come back with the first one.
int build_tiles(epi_batch *b, hipStream_t s, int32_t T, RowStats *h, int32_t *ntiles_out, bool *hinted) {
  *ntiles_out = 0;
  if (hinted) *hinted = false;
  memset(h, 0, sizeof(*h));
  if ((T & (T - 1)) != 0) return fail(EPI_ERR_ARG, "tile size must be a power of two");
  const int sh = log2_tile(T);
  EPI_TRY(b->misc.ensure(256));
  // misc layout (u32): [0] tile count, [1] pool cursor, [2] output rows, [3] heavy tiles, [8] largest heavy tile
  uint32_t *d_misc = b->misc.as<uint32_t>();
  if (b->n == 0) { EPI_HIP(hipMemsetAsync(d_misc, 0, 36, s)); return EPI_OK; }
  EPI_TRY(fetch_row_stats(b, s));
  *h = b->h_stats;
  if (h->bad_len) return fail(EPI_ERR_ARG, "offsets are not non-decreasing, or start+length exceeds int32");
  if (h->bad_strand) return fail(EPI_ERR_ARG, "strand values must be 1 ('+') or 2 ('-')");
  if (h->unsorted)
    return fail(EPI_ERR_UNSORTED, "rows are not sorted by (rname,start); the reference requires a pre-sorted dataset "
                                  "(src/rcpp_cx_report.cpp:19)");
  const int32_t lmax = h->max_len > 0 ? h->max_len : 1;
  const int64_t nb = (b->n + TB_ROWS - 1) / TB_ROWS;
  EPI_TRY(check_grid(nb, TB_THREADS, "tile index"));
  int slot = -1;
  for (int i = 0; i < 4; i++) if (b->tile_hint_T[i] == T) slot = i;
  const int use_hint = options().tile_hint;                // EPIHIP_TILE_HINT=0: every call counts, scans and asks (A/B runs, tests)
  auto remember_table = [&](uint32_t nt) -> int {            // (after the pass that filled b->tiles)
    b->tiles_T = 0;
    if (!b->cols_owned) return EPI_OK;
    EPI_TRY(b->tiles_nt_dev.ensure(4));
    EPI_HIP(hipMemcpyAsync(b->tiles_nt_dev.p, d_misc, 4, hipMemcpyDeviceToDevice, s));
    b->tiles_T = T; b->tiles_nt = (int32_t)nt; b->tiles_lmax = lmax; b->tiles_shared = b->shared_keys;
    return EPI_OK;
  };
  if (use_hint && hinted && b->cols_owned && b->tiles_T == T && b->tiles_lmax == lmax && b->tiles_shared == b->shared_keys) {
    // the batch owns its columns and the table of the last build was made for this tile size and these shared keys: it is
    // still there (tiles, d_slot_tile); misc[0] gets the count the pass had left, which the caller compares as always
    hipLaunchKernelGGL(k_misc_reset, dim3(1), dim3(64), 0, s, d_misc, b->tiles_nt_dev.as<uint32_t>());   // (one launch instead of three copies)
    EPI_HIP(hipGetLastError());
    *hinted = true;
    *ntiles_out = b->tiles_nt;
    return EPI_OK;
  }
  if (use_hint && hinted && slot >= 0 && b->tile_hint_lmax[slot] == lmax) {
    // The tile count and the per-block offsets are functions of the batch's rows and T alone: with those of an earlier
    // call the table is allocated up front and filled by ONE pass that verifies them block by block; the caller
    // compares misc[0] with the count when it next synchronises anyway.
    const uint32_t nt = (uint32_t)b->tile_hint_nt[slot];
    EPI_TRY(b->tiles.ensure((size_t)nt * sizeof(Tile)));
    const int32_t nshared = (int32_t)b->shared_keys.size();
    if (nshared > 0) {
      EPI_TRY(b->d_slot_tile.ensure((size_t)nshared * 4));
      EPI_HIP(hipMemsetAsync(b->d_slot_tile.p, 0xFF, (size_t)nshared * 4, s));
    }
    EPI_HIP(hipMemsetAsync(d_misc, 0, 4, s));
    prof_begin("tile_index", s);
    hipLaunchKernelGGL((k_tile_pass<true, true>), dim3((unsigned)nb), dim3(TB_THREADS), 0, s, b->start, b->rname, b->n, lmax, sh,
                       b->tile_bsum[slot].as<uint32_t>(), b->tiles.as<Tile>(), b->d_shared_keys.as<int64_t>(), nshared,
                       b->d_slot_tile.as<int32_t>(), d_misc, (int64_t)nt);
    prof_end("tile_index", s);
    EPI_HIP(hipGetLastError());
    EPI_TRY(remember_table(nt));
    *hinted = true;
    *ntiles_out = (int32_t)nt;
    return EPI_OK;
  }
  b->tiles_T = 0;
  EPI_TRY(b->scan_tmp.ensure((size_t)nb * 4));
  uint32_t *bsum = b->scan_tmp.as<uint32_t>();
  hipLaunchKernelGGL((k_tile_pass<false>), dim3((unsigned)nb), dim3(TB_THREADS), 0, s, b->start, b->rname, b->n, lmax, sh, bsum,
                     (Tile *)nullptr, (const int64_t *)nullptr, 0, (int32_t *)nullptr, d_misc, (int64_t)0);
  EPI_HIP(hipGetLastError());
  EPI_TRY(scan_block_sums_inplace(bsum, nb, d_misc, s));
  uint32_t nt = 0;
  EPI_TRY(read_scalars(b, s, d_misc, 4, &nt));
  if (nt <= 0x7FFFFFF0u) {                                 // remember count and block offsets for the one-pass path above
    for (int i = 0; i < 4 && slot < 0; i++) if (b->tile_hint_T[i] == 0) slot = i;
    if (slot >= 0 && b->tile_bsum[slot].ensure((size_t)nb * 4) == EPI_OK &&
        hipMemcpyAsync(b->tile_bsum[slot].p, bsum, (size_t)nb * 4, hipMemcpyDeviceToDevice, s) == hipSuccess) {
      b->tile_hint_T[slot] = T; b->tile_hint_nt[slot] = (int32_t)nt; b->tile_hint_lmax[slot] = lmax;
    }
  }
  if (nt > 0x7FFFFFF0u) return fail(EPI_ERR_ARG, "too many tiles (%u)", nt);
  EPI_TRY(b->tiles.ensure((size_t)nt * sizeof(Tile)));
  const int32_t nshared = (int32_t)b->shared_keys.size();
  if (nshared > 0) {   // slot -> this rank's tile index (-1: this rank has no rows near that tile)
    EPI_TRY(b->d_slot_tile.ensure((size_t)nshared * 4));
    EPI_HIP(hipMemsetAsync(b->d_slot_tile.p, 0xFF, (size_t)nshared * 4, s));
  }
  hipLaunchKernelGGL((k_tile_pass<true>), dim3((unsigned)nb), dim3(TB_THREADS), 0, s, b->start, b->rname, b->n, lmax, sh, bsum,
                     b->tiles.as<Tile>(), b->d_shared_keys.as<int64_t>(), nshared, b->d_slot_tile.as<int32_t>(), d_misc, (int64_t)nt);
  EPI_HIP(hipGetLastError());
  EPI_TRY(remember_table(nt));
  *ntiles_out = (int32_t)nt;
  return EPI_OK;
}

}  // namespace epi
