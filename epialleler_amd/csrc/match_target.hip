// rcpp_match_amplicon / rcpp_match_capture (src/rcpp_match_target.cpp:16-81): every read is matched to
// the FIRST row of a BED table it fits -- by start or end within a tolerance (amplicons) or by minimum
// overlap (capture).  One thread per read; the BED rows are staged through LDS in blocks of 1024 and
// scanned in file order ("bed is not sorted intentionally", :11).  Output is the 1-based BED row or
// R's NA_integer_ (INT_MIN).
#include "common.hpp"

namespace epi {

constexpr int MT_CHUNK = 1024;

template <bool CAPTURE>
__global__ __launch_bounds__(256) void k_match_target(const int32_t *__restrict__ rname, const int32_t *__restrict__ start,
                                                       const int32_t *__restrict__ len, int64_t n,
                                                       const int32_t *__restrict__ b_chr, const int32_t *__restrict__ b_start,
                                                       const int32_t *__restrict__ b_end, int32_t nbed, int32_t param,
                                                       int32_t *__restrict__ out) {
  __shared__ int32_t s_chr[MT_CHUNK], s_start[MT_CHUNK], s_end[MT_CHUNK];
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = x < n;
  int32_t chr = 0, rs = 0, re = 0;
  if (valid) {
    chr = rname[x];
    rs = start[x];
    re = rs + len[x] - 1;                                     // :34 / :67
  }
  int32_t res = INT32_MIN;                                    // NA_INTEGER
  bool done = !valid;
  for (int32_t base = 0; base < nbed; base += MT_CHUNK) {
    const int32_t m = nbed - base < MT_CHUNK ? nbed - base : MT_CHUNK;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 256) { s_chr[i] = b_chr[base + i]; s_start[i] = b_start[base + i]; s_end[i] = b_end[base + i]; }
    __syncthreads();
    if (!done) {
      for (int32_t i = 0; i < m; i++) {
        bool hit;
        if (CAPTURE) {
          const int32_t ov = min(re, s_end[i]) - max(rs, s_start[i]) + 1;             // :69
          hit = chr == s_chr[i] && ov >= param;                                        // :70
        } else {
          hit = chr == s_chr[i] && (abs(rs - s_start[i]) <= param || abs(re - s_end[i]) <= param);   // :36-38
        }
        if (hit) { res = base + i + 1; done = true; break; }                           // first match only
      }
    }
  }
  if (valid) out[x] = res;
}

}  // namespace epi

using namespace epi;

extern "C" int epi_batch_match_target_dev(epi_batch *b, const int32_t *d_bed_chr, const int32_t *d_bed_start,
                                          const int32_t *d_bed_end, int32_t nbed, int32_t capture, int32_t param,
                                          int32_t *d_match_out, void *stream) {
  if (!b || nbed < 0 || (nbed > 0 && (!d_bed_chr || !d_bed_start || !d_bed_end)) || (b->n > 0 && !d_match_out))
    return fail(EPI_ERR_ARG, "epi_batch_match_target_dev: bad arguments");
  if (b->n == 0) return EPI_OK;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  const unsigned nb = (unsigned)((b->n + 255) / 256);
  if (capture)
    hipLaunchKernelGGL((k_match_target<true>), dim3(nb), dim3(256), 0, s, b->rname, b->start, b->len, b->n, d_bed_chr,
                       d_bed_start, d_bed_end, nbed, param, d_match_out);
  else
    hipLaunchKernelGGL((k_match_target<false>), dim3(nb), dim3(256), 0, s, b->rname, b->start, b->len, b->n, d_bed_chr,
                       d_bed_start, d_bed_end, nbed, param, d_match_out);
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}
