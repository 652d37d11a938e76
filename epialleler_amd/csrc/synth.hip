// Synthetic packed-template generator (bench / full-size tests): writes the
// sorted SoA batch straight into HBM.  Counter-based hashing only, so the
// numpy mirror in tests/synth_np.py produces the same bytes (checked by
// tests/test_gpu_parity.py::test_synth_matches_numpy_mirror).  Model (DESIGN.md
// "Synthetic workload", after SURVEY.md 8d): n_chr chromosomes, uniform
// coverage of fixed-length templates at the given depth, one context track per
// chromosome position ('.' 0.76, h 0.135, x 0.06, z 0.035, u 0.01), 10 % hyper-
// methylated reads (CpG methylated w.p. 0.9, else 0.05; other contexts 0.01),
// 1 % of cytosine bytes re-labelled with another context.
#include "common.hpp"
#include <string.h>

namespace epi {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t hash3(uint64_t seed, uint64_t stream, uint64_t idx) {
  return mix64(mix64(seed + stream * 0xD1B54A32D192ED03ull) ^ idx);
}

constexpr int64_t kSynthChunk = 1LL << 30;   // dwords per launch of the byte kernels

struct SynthGeom {
  uint64_t seed;
  int64_t n_total, row_first, n;
  int64_t rows_per_chr, chr_len, stride;
  int32_t L, gap_from, gap_len;
};

__device__ __forceinline__ void synth_row(const SynthGeom &g, int64_t x, int32_t *chr, int32_t *start, int32_t *strand, bool *hyper) {
  const int64_t c = x / g.rows_per_chr;
  const int64_t j = x - c * g.rows_per_chr;
  *chr = (int32_t)c;
  *start = (int32_t)(1 + (j * g.chr_len) / g.rows_per_chr + (int64_t)(hash3(g.seed, 1, (uint64_t)x) % (uint64_t)g.stride));
  *strand = 1 + (int32_t)(hash3(g.seed, 2, (uint64_t)x) & 1ull);
  *hyper = (hash3(g.seed, 3, (uint64_t)x) % 10ull) == 0ull;
}

__device__ __forceinline__ uint32_t synth_byte(const SynthGeom &g, int64_t x, int32_t i, int32_t chr, int32_t start, bool hyper) {
  if (g.gap_len > 0 && i >= g.gap_from && i < g.gap_from + g.gap_len) return 0xFBu;
  const uint64_t pos = (uint64_t)((int64_t)start + i);
  const uint32_t t = (uint32_t)(hash3(g.seed, 16 + (uint64_t)chr, pos) % 1000ull);
  uint32_t code;
  if (t < 760u) return 0x10u | 12u;
  else if (t < 895u) code = 10u;   // h
  else if (t < 955u) code = 14u;   // x
  else if (t < 990u) code = 15u;   // z
  else code = 13u;                 // u
  const uint64_t v = hash3(g.seed, 4, ((uint64_t)x << 16) + (uint64_t)i);
  if ((v % 100ull) == 0ull) code = code == 10u ? 14u : (code == 14u ? 15u : 10u);   // h->x, x->z, z->h, u->h
  const uint32_t thr = code == 15u ? (hyper ? 900u : 50u) : 10u;
  if ((uint32_t)((v >> 20) % 1000ull) < thr) code -= 8u;   // methylated: upper case
  return 0x10u | code;
}

__global__ __launch_bounds__(256) void k_synth_meta(SynthGeom g, int64_t *__restrict__ off, int32_t *__restrict__ rname,
                                                     int32_t *__restrict__ strand, int32_t *__restrict__ start) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k > g.n) return;
  off[k] = k * (int64_t)g.L;
  if (k == g.n) return;
  int32_t c, st, sd;
  bool hy;
  synth_row(g, g.row_first + k, &c, &st, &sd, &hy);
  rname[k] = c + 1;
  strand[k] = sd;
  start[k] = st;
}

// one thread per 4 output bytes (one dword store)
__global__ __launch_bounds__(256) void k_synth_bytes(SynthGeom g, uint32_t *__restrict__ xm32, int64_t d0, int64_t ndw) {
  const int64_t d = d0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (d >= ndw) return;
  const int64_t total = g.n * (int64_t)g.L;
  uint32_t w = 0;
  int64_t k_prev = -1;
  int32_t c = 0, st = 0, sd = 0;
  bool hy = false;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int64_t f = d * 4 + q;
    uint32_t byte = 0xFBu;
    if (f < total) {
      const int64_t k = f / g.L;
      const int32_t i = (int32_t)(f - k * g.L);
      if (k != k_prev) { synth_row(g, g.row_first + k, &c, &st, &sd, &hy); k_prev = k; }
      byte = synth_byte(g, g.row_first + k, i, c, st, hy);
    }
    w |= byte << (8 * q);
  }
  xm32[d] = w;
}

// ---- rows given by the caller (uniform-random starts sorted on device, ragged lengths) -----------------
// Per-row hashes use the global sorted row id x = row_first + k; every `gap_every`-th template (by hash) carries
// `gap_len` filler bytes (0xFB) in its middle: two mates that do not meet.
__global__ __launch_bounds__(256) void k_synth_fill_strand(uint64_t seed, int64_t row_first, int64_t n, int32_t *__restrict__ strand) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n) strand[k] = 1 + (int32_t)(hash3(seed, 2, (uint64_t)(row_first + k)) & 1ull);
}

__global__ __launch_bounds__(256) void k_synth_fill_bytes(SynthGeom g, const int64_t *__restrict__ off, const int32_t *__restrict__ rname,
                                                           const int32_t *__restrict__ start, int32_t gap_every,
                                                           uint32_t *__restrict__ xm32, int64_t d0, int64_t ndw) {
  const int64_t d = d0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (d >= ndw) return;
  const int64_t total = off[g.n];
  uint32_t w = 0;
  int64_t k = -1, k_end = -1, k_off = 0;
  int32_t c = 0, st = 0, len = 0, g0 = 0;
  bool hy = false, gap = false;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int64_t f = d * 4 + q;
    uint32_t byte = 0xFBu;
    if (f < total) {
      if (f >= k_end) {                                    // row of byte f: the last k with off[k] <= f
        int64_t a = k < 0 ? 0 : k + 1, z = g.n;            // (zero-length rows are stepped over)
        while (z - a > 1) { const int64_t m = (a + z) >> 1; if (off[m] <= f) a = m; else z = m; }
        k = a;
        k_off = off[k]; k_end = off[k + 1];
        len = (int32_t)(k_end - k_off);
        c = rname[k] - 1; st = start[k];
        const uint64_t x = (uint64_t)(g.row_first + k);
        hy = (hash3(g.seed, 3, x) % 10ull) == 0ull;
        gap = gap_every > 0 && g.gap_len > 0 && (hash3(g.seed, 5, x) % (uint64_t)gap_every) == 0ull;
        g0 = len / 2 - g.gap_len / 2;
      }
      const int32_t i = (int32_t)(f - k_off);
      if (gap && i >= g0 && i < g0 + g.gap_len) byte = 0xFBu;
      else { SynthGeom h = g; h.gap_len = 0; byte = synth_byte(h, g.row_first + k, i, c, st, hy); }
    }
    w |= byte << (8 * q);
  }
  xm32[d] = w;
}

}  // namespace epi

using namespace epi;

extern "C" int epi_synth_fill_dev(uint64_t seed, int64_t row_first, int64_t n, const int64_t *d_off, const int32_t *d_rname,
                                  const int32_t *d_start, int64_t nbytes, int32_t gap_every, int32_t gap_len,
                                  uint8_t *d_xm, int32_t *d_strand, void *stream) {
  if (n < 0 || row_first < 0 || nbytes < 0 || gap_len < 0 || !d_off) return fail(EPI_ERR_ARG, "epi_synth_fill_dev: bad parameters");
  if (n > 0 && (!d_xm || !d_rname || !d_start || !d_strand)) return fail(EPI_ERR_ARG, "epi_synth_fill_dev: NULL buffer");
  if ((reinterpret_cast<uintptr_t>(d_xm) & 3) != 0) return fail(EPI_ERR_ARG, "epi_synth_fill_dev: d_xm must be 4-byte aligned");
  if (n == 0) return EPI_OK;
  SynthGeom g;
  memset(&g, 0, sizeof(g));
  g.seed = seed; g.row_first = row_first; g.n = n; g.gap_len = gap_len;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_synth_fill_strand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, row_first, n, d_strand);
  const int64_t ndw = (nbytes + 3) / 4;                    // the caller's buffer is padded to 16 bytes
  for (int64_t d0 = 0; d0 < ndw; d0 += kSynthChunk) {
    const int64_t cnt = ndw - d0 < kSynthChunk ? ndw - d0 : kSynthChunk;
    hipLaunchKernelGGL(k_synth_fill_bytes, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, g, d_off, d_rname, d_start, gap_every,
                       reinterpret_cast<uint32_t *>(d_xm), d0, ndw);
  }
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}

extern "C" int epi_synth_generate_dev(const epi_synth_params *p, uint8_t *d_xm, int64_t *d_off, int32_t *d_rname,
                                      int32_t *d_strand, int32_t *d_start, void *stream) {
  if (!p || !d_off) return fail(EPI_ERR_ARG, "epi_synth_generate_dev: NULL argument");
  if (p->n < 0 || p->n_total <= 0 || p->row_first < 0 || p->row_first + p->n > p->n_total || p->read_len <= 0 ||
      p->read_len > 65535 || p->n_chr <= 0 || p->depth <= 0 || p->gap_len < 0 || p->gap_from < 0)
    return fail(EPI_ERR_ARG, "epi_synth_generate_dev: bad parameters");
  if (p->n > 0 && (!d_xm || !d_rname || !d_strand || !d_start)) return fail(EPI_ERR_ARG, "epi_synth_generate_dev: NULL buffer");
  if ((reinterpret_cast<uintptr_t>(d_xm) & 3) != 0) return fail(EPI_ERR_ARG, "epi_synth_generate_dev: d_xm must be 4-byte aligned");
  SynthGeom g;
  g.seed = p->seed;
  g.n_total = p->n_total; g.row_first = p->row_first; g.n = p->n;
  g.rows_per_chr = (p->n_total + p->n_chr - 1) / p->n_chr;
  g.chr_len = g.rows_per_chr * p->read_len / p->depth;
  if (g.chr_len < 1) g.chr_len = 1;
  g.stride = g.chr_len / g.rows_per_chr;
  if (g.stride < 1) g.stride = 1;
  if (g.chr_len + p->read_len + g.stride >= 0x7FFFFFFFLL) return fail(EPI_ERR_ARG, "epi_synth_generate_dev: chromosome too long for int32 positions");
  g.L = p->read_len; g.gap_from = p->gap_from; g.gap_len = p->gap_len;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned nbm = (unsigned)((p->n + 1 + 255) / 256);
  hipLaunchKernelGGL(k_synth_meta, dim3(nbm), dim3(256), 0, s, g, d_off, d_rname, d_strand, d_start);
  const int64_t ndw = (p->n * (int64_t)p->read_len + 3) / 4;   // caller's buffer is padded to 16 bytes
  // a grid holds fewer than 2^32 threads: 2^30 dwords per launch (a 100 M x 300 B batch is 7.5e9 dwords; one launch
  // of that size silently wrapped and left everything past 12.8 GB unwritten -- found by the 30 GB full-size test)
  for (int64_t d0 = 0; d0 < ndw; d0 += kSynthChunk) {
    const int64_t cnt = ndw - d0 < kSynthChunk ? ndw - d0 : kSynthChunk;
    hipLaunchKernelGGL(k_synth_bytes, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, g, reinterpret_cast<uint32_t *>(d_xm), d0, ndw);
  }
  EPI_HIP(hipGetLastError());
  return EPI_OK;
}
