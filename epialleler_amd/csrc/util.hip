// Device-wide exclusive scan over u32 (three launches: block sums, scan of the
// block sums, rescan + offset).  Used for tile offsets and output-row offsets:
// tiny metadata passes next to the byte-scan kernels.
#include "common.hpp"

namespace epi {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;                       // per thread
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;  // 4096 items per block

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int) { return wave_scan_u32(v); }

// exclusive scan of one value per thread across a 256-thread block; returns exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total, uint32_t *s_wave /* [5] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = wave_incl_scan(v, lane);
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) { uint32_t t = s_wave[w]; s_wave[w] = acc; acc += t; }
    s_wave[SCAN_THREADS / 64] = acc;
  }
  __syncthreads();
  uint32_t ex = inc - v + s_wave[wave];
  *total = s_wave[SCAN_THREADS / 64];
  __syncthreads();
  return ex;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const uint32_t *__restrict__ in, int64_t n,
                                                                   uint32_t *__restrict__ bsum) {
  __shared__ uint32_t s_wave[SCAN_THREADS / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    int64_t k = base + i;
    if (k < n) acc += in[k];
  }
  uint32_t total;
  (void)block_excl_scan(acc, &total, s_wave);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// single block: exclusive scan of bsum[0..nb) in place, total to *d_total
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_bsums(uint32_t *__restrict__ bsum, int64_t nb,
                                                              uint32_t *__restrict__ d_total) {
  __shared__ uint32_t s_wave[SCAN_THREADS / 64 + 1];
  uint32_t carry = 0;
  for (int64_t base = 0; base < nb; base += SCAN_THREADS) {
    int64_t k = base + threadIdx.x;
    uint32_t v = k < nb ? bsum[k] : 0u;
    uint32_t total;
    uint32_t ex = block_excl_scan(v, &total, s_wave);
    if (k < nb) bsum[k] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && d_total) *d_total = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                              int64_t n, const uint32_t *__restrict__ bsum) {
  __shared__ uint32_t s_wave[SCAN_THREADS / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    int64_t k = base + i;
    v[i] = k < n ? in[k] : 0u;
    acc += v[i];
  }
  uint32_t total;
  uint32_t ex = block_excl_scan(acc, &total, s_wave) + bsum[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    int64_t k = base + i;
    if (k < n) out[k] = ex;
    ex += v[i];
  }
}

// exclusive scan, in place, of nb block sums a caller produced itself (tiles.hip); total to *d_total
int scan_block_sums_inplace(uint32_t *d_bsum, int64_t nb, uint32_t *d_total, hipStream_t s) {
  hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SCAN_THREADS), 0, s, d_bsum, nb, d_total);
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}

int scan_exclusive_u32(const uint32_t *d_in, uint32_t *d_out, int64_t n, uint32_t *d_total, DevBuf &tmp,
                       hipStream_t s) {
  if (n <= 0) {
    if (d_total) EPI_HIP(hipMemsetAsync(d_total, 0, 4, s));
    return EPI_OK;
  }
  const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  EPI_TRY(tmp.ensure((size_t)nb * 4));
  uint32_t *bsum = tmp.as<uint32_t>();
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, d_in, n, bsum);
  hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SCAN_THREADS), 0, s, bsum, nb, d_total);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, d_in, d_out, n, bsum);
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}

}  // namespace epi
