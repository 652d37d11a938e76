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

namespace epi {

__device__ __forceinline__ int64_t tile_of(int64_t pos, int32_t T) { return (pos + kPosBias) / T; }

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
    if (len) atomicMax(&st->max_len, len);
    if (unsorted) atomicOr(&st->unsorted, 1);
    if (bad_strand) atomicOr(&st->bad_strand, 1);
    if (bad_len) atomicOr(&st->bad_len, 1);
  }
}

// tiles row x must create: (lo .. b], see header comment
__device__ __forceinline__ void row_tile_span(const int32_t *start, const int32_t *rname, int64_t x, int32_t lmax,
                                              int32_t T, int64_t *lo, int64_t *b) {
  const int64_t s = start[x];
  const int64_t a = tile_of(s, T);
  *b = tile_of(s + lmax - 1, T);
  *lo = a;
  if (x > 0 && rname[x - 1] == rname[x]) {
    const int64_t bp = tile_of((int64_t)start[x - 1] + lmax - 1, T);
    if (bp + 1 > *lo) *lo = bp + 1;
  }
}

__global__ __launch_bounds__(256) void k_tile_counts(const int32_t *__restrict__ start, const int32_t *__restrict__ rname,
                                                      int64_t n, int32_t lmax, int32_t T, uint32_t *__restrict__ cnt) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= n) return;
  int64_t lo, b;
  row_tile_span(start, rname, x, lmax, T, &lo, &b);
  cnt[x] = b >= lo ? (uint32_t)(b - lo + 1) : 0u;
}

// first row y in [0,n) with (rname[y], start[y]) >= (r, s)
__device__ __forceinline__ int64_t lower_bound_rows(const int32_t *rname, const int32_t *start, int64_t n, int32_t r, int64_t s) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    const int32_t rm = rname[mid];
    const bool less = rm < r || (rm == r && (int64_t)start[mid] < s);
    if (less) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_tile_fill(const int32_t *__restrict__ start, const int32_t *__restrict__ rname,
                                                    int64_t n, int32_t lmax, int32_t T, const uint32_t *__restrict__ row_off,
                                                    Tile *__restrict__ tiles, const int64_t *__restrict__ shared_keys,
                                                    int32_t nshared) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= n) return;
  int64_t lo, b;
  row_tile_span(start, rname, x, lmax, T, &lo, &b);
  if (b < lo) return;
  const int32_t r = rname[x];
  uint32_t slot_base = row_off[x];
  for (int64_t t = lo; t <= b; t++) {
    Tile td;
    td.pos0 = t * T - kPosBias;
    td.rname = r;
    td.row_lo = (int32_t)lower_bound_rows(rname, start, n, r, td.pos0 - lmax + 1);
    td.row_hi = (int32_t)lower_bound_rows(rname, start, n, r, td.pos0 + T);
    td.slot = -1;
    if (nshared > 0) {
      const int64_t key = ((int64_t)r << 32) | (int64_t)(uint32_t)t;
      int32_t a = 0, z = nshared;
      while (a < z) { int32_t m = (a + z) >> 1; if (shared_keys[m] < key) a = m + 1; else z = m; }
      if (a < nshared && shared_keys[a] == key) td.slot = a;
    }
    tiles[slot_base + (uint32_t)(t - lo)] = td;
  }
}

int build_row_stats(epi_batch *b, hipStream_t s, RowStats *h) {
  EPI_TRY(b->stats.ensure(sizeof(RowStats)));
  EPI_HIP(hipMemsetAsync(b->stats.p, 0, sizeof(RowStats), s));
  if (b->n > 0) {
    const unsigned nb = (unsigned)((b->n + 255) / 256);
    hipLaunchKernelGGL(k_row_stats, dim3(nb), dim3(256), 0, s, b->start, b->rname, b->strand, b->off, b->n,
                       b->stats.as<RowStats>());
    EPI_HIP(hipGetLastError());
  }
  EPI_TRY(read_scalars(b, s, b->stats.p, sizeof(RowStats), h));
  if (h->bad_len) return fail(EPI_ERR_ARG, "offsets are not non-decreasing, or start+length exceeds int32");
  if (h->bad_strand) return fail(EPI_ERR_ARG, "strand values must be 1 ('+') or 2 ('-')");
  return EPI_OK;
}

int build_tiles(epi_batch *b, hipStream_t s, int32_t max_len, int32_t T, int32_t *ntiles_out) {
  *ntiles_out = 0;
  if (b->n == 0) return EPI_OK;
  const int32_t lmax = max_len > 0 ? max_len : 1;
  const unsigned nb = (unsigned)((b->n + 255) / 256);
  EPI_TRY(b->row_cnt.ensure((size_t)b->n * 4));
  EPI_TRY(b->row_off.ensure((size_t)b->n * 4));
  EPI_TRY(b->misc.ensure(256));
  uint32_t *d_total = b->misc.as<uint32_t>();   // misc[0] = tile count
  hipLaunchKernelGGL(k_tile_counts, dim3(nb), dim3(256), 0, s, b->start, b->rname, b->n, lmax, T, b->row_cnt.as<uint32_t>());
  EPI_TRY(scan_exclusive_u32(b->row_cnt.as<uint32_t>(), b->row_off.as<uint32_t>(), b->n, d_total, b->scan_tmp, s));
  uint32_t nt = 0;
  EPI_TRY(read_scalars(b, s, d_total, 4, &nt));
  if (nt > 0x7FFFFFF0u) return fail(EPI_ERR_ARG, "too many tiles (%u)", nt);
  EPI_TRY(b->tiles.ensure((size_t)nt * sizeof(Tile)));
  const int32_t nshared = (int32_t)b->shared_keys.size();
  hipLaunchKernelGGL(k_tile_fill, dim3(nb), dim3(256), 0, s, b->start, b->rname, b->n, lmax, T, b->row_off.as<uint32_t>(),
                     b->tiles.as<Tile>(), b->d_shared_keys.as<int64_t>(), nshared);
  EPI_HIP(hipGetLastError());
  *ntiles_out = (int32_t)nt;
  return EPI_OK;
}

}  // namespace epi
