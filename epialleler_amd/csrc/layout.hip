// Position-congruent row layout: the batch's own copy of xm in which row x starts at an offset congruent to its start
// position modulo A (16; 4 on request), i.e. behind up to A - 1 bytes of filler.
//
// Why: the tile kernels (cx_report.hip, mhl_fused.hip) request POSITION-aligned 16-byte chunks -- a dword of xm then maps
// onto one LDS cell of four positions -- at the address  xm + off[x] - (start[x] - tile position 0) + 16 c.  With rows
// back to back that address has any byte alignment, and a global_load_dwordx4 that is not dword-aligned runs at about
// 0.85 of the rate of an aligned one (scratch/align_cost.py on the product kernels: config-2 stream with every row
// aligned 0.72-0.74 ms, every row misaligned 0.83-0.90, the uniform mix 0.81; profiles/r04_layout.txt).  The reference
// keeps one std::string per template (src/epialleleR.h:28-38), so where a row starts in the arena is this engine's own
// business: off[x] = start[x] (mod 16) makes every chunk of every row 16-byte aligned at no cost per report.  The byte
// format of a row is untouched; the filler between rows is never interpreted (first / last chunk of a row are masked to
// the row's own bytes by every kernel).
//
// Cost: one pass at batch creation (read B bytes, write B + <= 15 n), 7.5 bytes per row on average (2.5 % for PE150
// templates), and int32 len[] next to off[] (rows no longer end where the next one starts).  epi_batch_upload does it
// as part of the upload; for an adopted batch (caller-owned device memory) it is the caller's choice: epi_batch_realign.
#include "common.hpp"
#include <utility>

namespace epi {

constexpr int LY_ROWS = 1024;                     // rows per block of the offset scan (256 threads x 4)

// what row x adds to the running offset: its bytes and the filler behind it, so that row x + 1 starts congruent to its
// start position (by induction: off[x] = start[x] (mod A) and off[x + 1] = off[x] + len + ((start[x + 1] - start[x] - len) mod A))
__device__ __forceinline__ uint32_t ly_item(const int32_t *__restrict__ start, const int32_t *__restrict__ len, int64_t x, int64_t n,
                                            uint32_t am) {
  const uint32_t l = (uint32_t)len[x];
  const uint32_t pad = x + 1 < n ? ((uint32_t)start[x + 1] - ((uint32_t)start[x] + l)) & am : 0u;
  return l + pad;
}

__device__ __forceinline__ unsigned long long ly_wave_incl(unsigned long long v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive prefix of one value per thread over a 256-thread block; *total = the block's sum
__device__ __forceinline__ unsigned long long ly_block_excl(unsigned long long v, unsigned long long *total, unsigned long long *s_w /* [5] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long inc = ly_wave_incl(v, lane);
  __syncthreads();                                // (s_w may still be read from a previous round)
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long acc = 0;
    for (int w = 0; w < 4; w++) { const unsigned long long t = s_w[w]; s_w[w] = acc; acc += t; }
    s_w[4] = acc;
  }
  __syncthreads();
  *total = s_w[4];
  return s_w[wave] + inc - v;
}

__global__ __launch_bounds__(256) void k_ly_sums(const int32_t *__restrict__ start, const int32_t *__restrict__ len, int64_t n, uint32_t am,
                                                  unsigned long long *__restrict__ bsum) {
  __shared__ unsigned long long s_w[5];
  const int64_t x0 = (int64_t)blockIdx.x * LY_ROWS + (int64_t)threadIdx.x * 4;
  unsigned long long v = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) if (x0 + i < n) v += ly_item(start, len, x0 + i, n, am);
  unsigned long long total;
  (void)ly_block_excl(v, &total, s_w);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// one block: exclusive scan of bsum[0 .. nb) in place, starting at `first`; the end of the data to *d_end
__global__ __launch_bounds__(256) void k_ly_scan(unsigned long long *__restrict__ bsum, int64_t nb, unsigned long long first,
                                                  unsigned long long *__restrict__ d_end) {
  __shared__ unsigned long long s_w[5];
  unsigned long long carry = first;
  for (int64_t base = 0; base < nb; base += 256) {
    const int64_t i = base + threadIdx.x;
    const unsigned long long v = i < nb ? bsum[i] : 0ull;
    unsigned long long total;
    const unsigned long long ex = ly_block_excl(v, &total, s_w);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *d_end = carry;
}

__global__ __launch_bounds__(256) void k_ly_offsets(const int32_t *__restrict__ start, const int32_t *__restrict__ len, int64_t n, uint32_t am,
                                                     const unsigned long long *__restrict__ bsum, const unsigned long long *__restrict__ d_end,
                                                     int64_t *__restrict__ off_out) {
  __shared__ unsigned long long s_w[5];
  const int64_t x0 = (int64_t)blockIdx.x * LY_ROWS + (int64_t)threadIdx.x * 4;
  uint32_t it[4];
  unsigned long long v = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { it[i] = x0 + i < n ? ly_item(start, len, x0 + i, n, am) : 0u; v += it[i]; }
  unsigned long long total;
  unsigned long long o = bsum[blockIdx.x] + ly_block_excl(v, &total, s_w);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (x0 + i < n) off_out[x0 + i] = (int64_t)o;
    o += it[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) off_out[n] = (int64_t)*d_end;
}

struct __attribute__((packed, aligned(1))) LyU4u { uint32_t x, y, z, w; };

// G lanes copy the part of a row that lies in source bytes [c0, c1) -- `src` holds exactly those bytes (the whole old arena:
// c0 = 0; a staged piece of an upload: src[0] is source byte c0) -- to its place in the new arena: the bytes up to the first
// 16-byte boundary of the destination and behind the last one singly, the chunks in between as aligned 16-byte stores (of
// 16-byte loads at the source's alignment).  The piece that holds the END of row x (c0 < end <= c1, or end = 0 in the first
// piece) also writes the filler between row x and row x + 1, so every byte between off[0] and off[n] is written exactly once.
template <int G>
__global__ __launch_bounds__(256) void k_ly_copy(const uint8_t *__restrict__ src, int64_t c0, int64_t c1, const int64_t *__restrict__ src_off,
                                                  const int32_t *__restrict__ len, const int64_t *__restrict__ dst_off, int64_t row_a,
                                                  int64_t nrows, uint8_t *__restrict__ dst) {
  const int sub = threadIdx.x & (G - 1);
  const int64_t k = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  if (k >= nrows) return;
  const int64_t x = row_a + k;
  const int64_t s = src_off[x], d = dst_off[x];
  const int64_t L = len[x], e = s + L;
  int64_t lo = c0 > s ? c0 - s : 0, hi = (c1 < e ? c1 : e) - s;    // the row's bytes [lo, hi) are in this piece
  if (hi > lo) {
    const uint8_t *sp = src + (s - c0) + lo;
    uint8_t *dp = dst + d + lo;
    const int64_t nby = hi - lo;
    int64_t h = (16 - ((d + lo) & 15)) & 15;
    if (h > nby) h = nby;
    for (int64_t i = sub; i < h; i += G) dp[i] = sp[i];
    const int64_t nfull = (nby - h) >> 4;
    for (int64_t q = sub; q < nfull; q += G) {
      const LyU4u v = *reinterpret_cast<const LyU4u *>(sp + h + (q << 4));
      *reinterpret_cast<uint4 *>(dp + h + (q << 4)) = make_uint4(v.x, v.y, v.z, v.w);
    }
    for (int64_t i = h + (nfull << 4) + sub; i < nby; i += G) dp[i] = sp[i];
  }
  if ((e > c0 && e <= c1) || (e == 0 && c0 == 0)) {
    const int64_t gap = dst_off[x + 1] - d - L;                     // filler behind the row (0 .. 15)
    for (int64_t i = sub; i < gap; i += G) dst[d + L + i] = (uint8_t)0xFB;
  }
}

// lanes per row of the copy by the mean row (a long-read batch: a wavefront per row)
int layout_group(int64_t nbytes, int64_t n) {
  const int64_t mean = n > 0 ? nbytes / n : 0;
  int g = 8;
  while (g < 64 && (int64_t)g * 64 < mean) g <<= 1;
  return g;
}

int layout_copy_range(const uint8_t *src, int64_t c0, int64_t c1, const int64_t *src_off, const int32_t *len, const int64_t *dst_off,
                      int64_t row_a, int64_t nrows, int g, uint8_t *dst, hipStream_t s) {
  if (nrows <= 0) return EPI_OK;
  const unsigned blocks = (unsigned)((nrows * g + 255) / 256);
#define EPI_LY(GG) hipLaunchKernelGGL((k_ly_copy<GG>), dim3(blocks), dim3(256), 0, s, src, c0, c1, src_off, len, dst_off, row_a, nrows, dst)
  if (g == 8) EPI_LY(8); else if (g == 16) EPI_LY(16); else if (g == 32) EPI_LY(32); else EPI_LY(64);
#undef EPI_LY
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}

// The congruent offsets of the batch's rows (from start[] and len[]) into new_off [n + 1]; *h_end = the end of the data,
// *head = the filler in front of row 0.  One host synchronisation (the arena is allocated to size).
int layout_offsets(epi_batch *b, int modulus, hipStream_t s, DevBuf *new_off, unsigned long long *h_end, uint32_t *head) {
  const int64_t n = b->n;
  const int64_t nb = (n + LY_ROWS - 1) / LY_ROWS;
  const uint32_t am = (uint32_t)modulus - 1u;
  DevBuf tmp;
  int rc = EPI_OK;
  do {
    if ((rc = tmp.ensure((size_t)(nb + 2) * 8))) break;
    if ((rc = new_off->ensure((size_t)(n + 1) * 8))) break;
    unsigned long long *bsum = tmp.as<unsigned long long>(), *d_end = bsum + nb;
    int32_t first_start = 0;
    if ((rc = read_scalars(b, s, b->start, 4, &first_start))) break;
    *head = (uint32_t)first_start & am;
    hipLaunchKernelGGL(k_ly_sums, dim3((unsigned)nb), dim3(256), 0, s, b->start, b->len, n, am, bsum);
    hipLaunchKernelGGL(k_ly_scan, dim3(1), dim3(256), 0, s, bsum, nb, (unsigned long long)*head, d_end);
    hipLaunchKernelGGL(k_ly_offsets, dim3((unsigned)nb), dim3(256), 0, s, b->start, b->len, n, am, bsum, d_end, new_off->as<int64_t>());
    if (hipGetLastError() != hipSuccess) { rc = fail(EPI_ERR_HIP, "layout: launch failed"); break; }
    if ((rc = read_scalars(b, s, d_end, 8, h_end))) break;   // (synchronises: tmp may go)
  } while (0);
  tmp.release();
  return rc;
}

int realign_batch(epi_batch *b, int modulus, hipStream_t s) {
  if (modulus != 4 && modulus != 8 && modulus != 16) return fail(EPI_ERR_ARG, "realign: modulus must be 4, 8 or 16");
  if (b->congruent == modulus || b->n == 0) { b->congruent = modulus; return EPI_OK; }
  if (!b->stats_queued || !b->len) return fail(EPI_ERR_STATE, "realign: the batch has no row lengths yet");
  if (b->last_kind != 0) return fail(EPI_ERR_STATE, "epi_batch_realign: call it before the first report on the batch");
  EPI_HIP(hipStreamWaitEvent(s, b->stats_done, 0));       // len[] is written by k_row_stats, possibly on another stream
  DevBuf new_off, new_xm;
  int rc = EPI_OK;
  do {
    unsigned long long h_end = 0;
    uint32_t head = 0;
    if ((rc = layout_offsets(b, modulus, s, &new_off, &h_end, &head))) break;
    const size_t cap = ((size_t)h_end + 15) / 16 * 16 + 64;
    if ((rc = new_xm.ensure(cap))) break;
    uint8_t *dst = new_xm.as<uint8_t>();
    if (head && hipMemsetAsync(dst, 0xFB, head, s) != hipSuccess) { rc = fail(EPI_ERR_HIP, "realign: memset failed"); break; }
    if (hipMemsetAsync(dst + h_end, 0xFB, cap - (size_t)h_end, s) != hipSuccess) { rc = fail(EPI_ERR_HIP, "realign: memset failed"); break; }
    if ((rc = layout_copy_range(b->xm, 0, INT64_MAX, b->off, b->len, new_off.as<int64_t>(), 0, b->n, layout_group(b->nbytes, b->n), dst, s))) break;
    // the old arena may be freed (an uploaded batch) or handed back to its owner (an adopted one) when this returns
    if (hipStreamSynchronize(s) != hipSuccess) { rc = fail(EPI_ERR_HIP, "realign: %s", hipGetErrorName(hipGetLastError())); break; }
    if (!b->owns) {
      // an adopted batch: the three small columns become the batch's own as well -- from here on nobody else can change a
      // row, and what is derived from the rows alone (the tile table, tiles.hip) stays valid between reports
      const size_t cb = (size_t)b->n * 4;
      if ((rc = b->own_rname.ensure(cb + 4)) || (rc = b->own_strand.ensure(cb + 4)) || (rc = b->own_start.ensure(cb + 4))) break;
      if (hipMemcpyAsync(b->own_rname.p, b->rname, cb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipMemcpyAsync(b->own_strand.p, b->strand, cb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipMemcpyAsync(b->own_start.p, b->start, cb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) { rc = fail(EPI_ERR_HIP, "realign: column copy failed"); break; }
      b->rname = b->own_rname.as<int32_t>();
      b->strand = b->own_strand.as<int32_t>();
      b->start = b->own_start.as<int32_t>();
      b->owns = true;
    }
    std::swap(b->own_xm, new_xm);
    std::swap(b->own_off, new_off);
    b->xm = b->own_xm.as<uint8_t>();
    b->off = b->own_off.as<int64_t>();
    b->nbytes = (int64_t)h_end;
    b->congruent = modulus;
    b->cols_owned = true;
  } while (0);
  new_off.release(); new_xm.release();                    // (after the swap: the batch's previous own buffers, if it had any)
  return rc;
}

}  // namespace epi

using namespace epi;

extern "C" int epi_batch_realign(epi_batch *b, void *stream) {
  if (!b) return fail(EPI_ERR_ARG, "epi_batch_realign: NULL batch");
  EPI_HIP(hipSetDevice(b->eng->device));
  const int m = options().realign;
  if (m == 0) return EPI_OK;                               // EPIHIP_REALIGN=0: rows stay where the caller put them
  return realign_batch(b, m, pick_stream(b, stream));
}

extern "C" int epi_batch_layout(const epi_batch *b) { return b ? b->congruent : -1; }

// the rows as the kernels read them (device pointers owned by the batch or its creator; valid until the batch is freed or realigned)
extern "C" int epi_batch_view(const epi_batch *b, const uint8_t **d_xm, const int64_t **d_off, const int32_t **d_len, int64_t *nbytes) {
  if (!b) return fail(EPI_ERR_ARG, "epi_batch_view: NULL batch");
  if (d_xm) *d_xm = b->xm;
  if (d_off) *d_off = b->off;
  if (d_len) *d_len = b->len;
  if (nbytes) *nbytes = b->nbytes;
  return EPI_OK;
}
