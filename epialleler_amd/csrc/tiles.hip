// Tile index over the sorted rows.
//
// Positions are cut into tiles of T positions (kTile for CX, kMhlTile for lMHL) on an ABSOLUTE grid (so every GPU of a
// sharded run agrees on tile boundaries).  A tile (rname, t) exists iff some
// row of that rname could reach it: tile(start) <= t <= tile(start + Lmax - 1).
// Because rows are sorted by (rname,start), row x only has to create the tiles
// its predecessor did not already reach -- a purely local count -- and one
// exclusive scan over the rows turns the counts into tile slots.  This replaces
// the reference's "flush the map when start > max_pos" windowing
// (src/rcpp_cx_report.cpp:113) and the key sort of a sort+segmented-reduce
// scheme: the input order already is the sort.
#include "common.hpp"
#include <string.h>

namespace epi {

// T is a power of two: tile index by shift (sh = log2 T)
__device__ __forceinline__ int64_t tile_of(int64_t pos, int32_t sh) { return (pos + kPosBias) >> sh; }

__global__ __launch_bounds__(256) void k_row_stats(const int32_t *__restrict__ start, const int32_t *__restrict__ rname,
                                                    const int32_t *__restrict__ strand, const int64_t *__restrict__ off,
                                                    int64_t n, RowStats *__restrict__ st) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int len = 0, unsorted = 0, bad_strand = 0, bad_len = 0;
  if (x < n) {
    const int64_t l = off[x + 1] - off[x];
    const int32_t s0 = start[x];
    if (l < 0 || (int64_t)s0 + l > 0x7FFFFFFFLL) bad_len = 1; else len = (int)l;
    const int32_t sd = strand[x];
    if (sd != 1 && sd != 2) bad_strand = 1;
    if (x > 0) {
      const int32_t r0 = rname[x - 1], r1 = rname[x];
      if (r1 < r0 || (r1 == r0 && s0 < start[x - 1])) unsorted = 1;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    len = max(len, __shfl_xor(len, d, 64));
    unsorted |= __shfl_xor(unsorted, d, 64);
    bad_strand |= __shfl_xor(bad_strand, d, 64);
    bad_len |= __shfl_xor(bad_len, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    // uniform-length input: after the first few waves the cached maximum already covers `len`
    if (len > __atomic_load_n(&st->max_len, __ATOMIC_RELAXED)) atomicMax(&st->max_len, len);
    if (unsorted) atomicOr(&st->unsorted, 1);
    if (bad_strand) atomicOr(&st->bad_strand, 1);
    if (bad_len) atomicOr(&st->bad_len, 1);
  }
}

// tiles row x must create: (lo .. b], see header comment
__device__ __forceinline__ void row_tile_span(const int32_t *start, const int32_t *rname, int64_t x, int32_t lmax,
                                              int32_t sh, int64_t *lo, int64_t *b) {
  const int64_t s = start[x];
  const int64_t a = tile_of(s, sh);
  *b = tile_of(s + lmax - 1, sh);
  *lo = a;
  if (x > 0 && rname[x - 1] == rname[x]) {
    const int64_t bp = tile_of((int64_t)start[x - 1] + lmax - 1, sh);
    if (bp + 1 > *lo) *lo = bp + 1;
  }
}

__global__ __launch_bounds__(256) void k_tile_counts(const int32_t *__restrict__ start, const int32_t *__restrict__ rname,
                                                      int64_t n, const RowStats *__restrict__ st, int32_t sh,
                                                      uint32_t *__restrict__ cnt) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= n) return;
  const int32_t lmax = st->max_len > 0 ? st->max_len : 1;   // written by k_row_stats earlier on this stream
  int64_t lo, b;
  row_tile_span(start, rname, x, lmax, sh, &lo, &b);
  cnt[x] = b >= lo ? (uint32_t)(b - lo + 1) : 0u;
}

// First row y >= x of rname r with start[y] >= s (rows are sorted): gallop forward from x, then bisect.
// The rows of one tile are a few hundred at most, so this is ~2*log2(rows per tile) dependent loads.
__device__ __forceinline__ int64_t gallop_rows(const int32_t *rname, const int32_t *start, int64_t n, int64_t x,
                                               int32_t r, int64_t s) {
  auto less = [&](int64_t y) { const int32_t rm = rname[y]; return rm < r || (rm == r && (int64_t)start[y] < s); };
  int64_t lo = x, step = 64;
  int64_t hi = x + step;
  while (hi < n && less(hi)) { lo = hi + 1; step <<= 1; hi = lo + step; }
  if (hi > n) hi = n;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (less(mid)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_tile_fill(const int32_t *__restrict__ start, const int32_t *__restrict__ rname,
                                                    int64_t n, const RowStats *__restrict__ st, int32_t sh,
                                                    const uint32_t *__restrict__ row_off, Tile *__restrict__ tiles,
                                                    const int64_t *__restrict__ shared_keys, int32_t nshared,
                                                    int32_t *__restrict__ slot_tile) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= n) return;
  const int32_t lmax = st->max_len > 0 ? st->max_len : 1;
  const int64_t T = 1LL << sh;
  int64_t lo, b;
  row_tile_span(start, rname, x, lmax, sh, &lo, &b);
  if (b < lo) return;
  const int32_t r = rname[x];
  uint32_t slot_base = row_off[x];
  int64_t from = x;
  for (int64_t t = lo; t <= b; t++) {
    Tile td;
    td.pos0 = t * T - kPosBias;
    td.rname = r;
    // The row that creates a tile is the first one that can reach it (start >= pos0 - lmax + 1): it IS row_lo.
    td.row_lo = (int32_t)x;
    from = gallop_rows(rname, start, n, from, r, td.pos0 + T);   // first row starting beyond the tile
    td.row_hi = (int32_t)from;
    td.slot = -1;
    if (nshared > 0) {
      const int64_t key = ((int64_t)r << 32) | (int64_t)(uint32_t)t;
      int32_t a = 0, z = nshared;
      while (a < z) { int32_t m = (a + z) >> 1; if (shared_keys[m] < key) a = m + 1; else z = m; }
      if (a < nshared && shared_keys[a] == key) { td.slot = a; slot_tile[a] = (int32_t)(slot_base + (uint32_t)(t - lo)); }
    }
    tiles[slot_base + (uint32_t)(t - lo)] = td;
  }
}

static int log2_tile(int32_t T) {
  int sh = 0;
  while ((1 << sh) < T) sh++;
  return sh;
}

// Row statistics + tile table with ONE host synchronisation: k_row_stats -> k_tile_counts (reads the maximum
// length on the device) -> scan -> read back {stats, tile count} -> k_tile_fill.
int build_tiles(epi_batch *b, hipStream_t s, int32_t T, RowStats *h, int32_t *ntiles_out) {
  *ntiles_out = 0;
  memset(h, 0, sizeof(*h));
  if ((T & (T - 1)) != 0) return fail(EPI_ERR_ARG, "tile size must be a power of two");
  const int sh = log2_tile(T);
  EPI_TRY(b->misc.ensure(256));
  // misc layout (u32): [0] tile count, [1] pool cursor, [2] output rows, [4..7] RowStats
  uint32_t *d_misc = b->misc.as<uint32_t>();
  RowStats *d_st = reinterpret_cast<RowStats *>(d_misc + 4);
  EPI_HIP(hipMemsetAsync(d_misc, 0, 32, s));
  if (b->n == 0) return EPI_OK;
  const unsigned nb = (unsigned)((b->n + 255) / 256);
  EPI_TRY(b->row_cnt.ensure((size_t)b->n * 4));
  EPI_TRY(b->row_off.ensure((size_t)b->n * 4));
  hipLaunchKernelGGL(k_row_stats, dim3(nb), dim3(256), 0, s, b->start, b->rname, b->strand, b->off, b->n, d_st);
  hipLaunchKernelGGL(k_tile_counts, dim3(nb), dim3(256), 0, s, b->start, b->rname, b->n, d_st, sh, b->row_cnt.as<uint32_t>());
  EPI_TRY(scan_exclusive_u32(b->row_cnt.as<uint32_t>(), b->row_off.as<uint32_t>(), b->n, d_misc, b->scan_tmp, s));
  uint32_t host[8];
  EPI_TRY(read_scalars(b, s, d_misc, 32, host));
  memcpy(h, host + 4, sizeof(RowStats));
  if (h->bad_len) return fail(EPI_ERR_ARG, "offsets are not non-decreasing, or start+length exceeds int32");
  if (h->bad_strand) return fail(EPI_ERR_ARG, "strand values must be 1 ('+') or 2 ('-')");
  if (h->unsorted)
    return fail(EPI_ERR_UNSORTED, "rows are not sorted by (rname,start); the reference requires a pre-sorted dataset "
                                  "(src/rcpp_cx_report.cpp:19)");
  const uint32_t nt = host[0];
  if (nt > 0x7FFFFFF0u) return fail(EPI_ERR_ARG, "too many tiles (%u)", nt);
  EPI_TRY(b->tiles.ensure((size_t)nt * sizeof(Tile)));
  const int32_t nshared = (int32_t)b->shared_keys.size();
  if (nshared > 0) {   // slot -> this rank's tile index (-1: this rank has no rows near that tile)
    EPI_TRY(b->d_slot_tile.ensure((size_t)nshared * 4));
    EPI_HIP(hipMemsetAsync(b->d_slot_tile.p, 0xFF, (size_t)nshared * 4, s));
  }
  hipLaunchKernelGGL(k_tile_fill, dim3(nb), dim3(256), 0, s, b->start, b->rname, b->n, d_st, sh, b->row_off.as<uint32_t>(),
                     b->tiles.as<Tile>(), b->d_shared_keys.as<int64_t>(), nshared, b->d_slot_tile.as<int32_t>());
  EPI_HIP(hipGetLastError());
  *ntiles_out = (int32_t)nt;
  return EPI_OK;
}

}  // namespace epi
