// rcpp_extract_patterns (src/rcpp_extract_patterns.cpp:26-211): methylation patterns of the reads that overlap one
// target region.  The work is confined to that region (a few thousand reads), so this is three small kernels around
// two host decisions, not a bandwidth problem:
//   k_pat_flag     one thread per row: does the read overlap the target by min_overlap (:79-86)?  -> scan -> its index
//   k_pat_count    one thread per overlapping read: how often is each in-context position seen (:87-96)
//   host           valid positions = seen in >= min_ctx_freq of the overlapping reads and not highlighted (:103-108),
//                  merged with the highlight positions, ordered (:185)
//   k_pat_extract  one thread per overlapping read: its cell per valid position, methylated / total counts and the
//                  FNV-1a hash of (position, base) pairs, highlighted bases appended (:133-166)
//   host           drops empty patterns (:152) and returns the table.
// Quirks of the reference kept as they are: with clip=TRUE the byte loop ends at `overlap`,
// not at begin+overlap (:86,:132), and position bytes enter the hash sign-extended (char pointer, epialleleR.h:8-13).
#include "common.hpp"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace epi {

struct PatArgs {
  const uint8_t *xm;
  const int64_t *off;             // row x owns xm[off[x] .. off[x] + len[x])
  const int32_t *len;
  const int32_t *rname, *strand, *start;
  int64_t n;
  uint32_t target_rname, target_start, target_end, reverse_offset;
  int32_t min_overlap, clip;
  uint32_t ctx_mask;
  int64_t pos_lo;                 // window of positions that can occur: [pos_lo, pos_lo + nwin)
  int64_t nwin;
};

struct PatSpan { uint32_t start_x, begin_i, end_i, offset_x; bool ok; };

__device__ __forceinline__ PatSpan pat_span(const PatArgs &a, int64_t x) {
  PatSpan s;
  s.ok = false; s.start_x = 0; s.begin_i = 0; s.end_i = 0; s.offset_x = 0;
  if (a.rname[x] != (int32_t)a.target_rname) return s;                      // :78
  const uint32_t size_x = (uint32_t)a.len[x];
  const uint32_t start_x = (uint32_t)a.start[x];
  const uint32_t end_x = start_x + size_x - 1u;
  const uint32_t over_start = start_x > a.target_start ? start_x : a.target_start;
  const uint32_t over_end = end_x < a.target_end ? end_x : a.target_end;
  const int32_t overlap = (int32_t)(over_end - over_start + 1u);            // :84
  if (overlap < a.min_overlap) return s;
  s.ok = true;
  s.start_x = start_x;
  s.offset_x = a.strand[x] == 2 ? a.reverse_offset : 0u;
  s.begin_i = a.clip ? over_start - start_x : 0u;
  s.end_i = a.clip ? (uint32_t)overlap : size_x;
  if (s.end_i > size_x) s.end_i = size_x;                                   // (the reference would read past the string)
  return s;
}

__global__ __launch_bounds__(256) void k_pat_flag(PatArgs a, uint32_t *__restrict__ flag) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= a.n) return;
  flag[x] = pat_span(a, x).ok ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_pat_count(PatArgs a, const uint32_t *__restrict__ flag, uint32_t *__restrict__ cnt) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= a.n || !flag[x]) return;
  const PatSpan s = pat_span(a, x);
  const uint8_t *p = a.xm + a.off[x];
  for (uint32_t i = s.begin_i; i < s.end_i; i++) {
    if (!((a.ctx_mask >> (p[i] & 15u)) & 1u)) continue;
    const int64_t w = (int64_t)(int32_t)(s.start_x + i - s.offset_x) - a.pos_lo;
    if (w >= 0 && w < a.nwin) atomicAdd(cnt + w, 1u);
  }
}

struct PatOut {
  int32_t *nonempty, *strand, *start, *end, *nbase, *meth;
  unsigned long long *fnv;
  int32_t *cells;                 // [ncol][npat0]
};

__device__ __forceinline__ void fnv_char(unsigned long long &h, uint32_t v) {   // four bytes through a (signed) char pointer
#pragma unroll
  for (int k = 0; k < 4; k++) {
    h ^= (unsigned long long)(long long)(signed char)((v >> (8 * k)) & 0xFFu);
    h *= 1099511628211ull;
  }
}

__global__ __launch_bounds__(256) void k_pat_extract(PatArgs a, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ cidx,
                                                      const int32_t *__restrict__ colmap, const int32_t *__restrict__ hlght,
                                                      const int32_t *__restrict__ hcol, int32_t nhlght, int64_t npat0, PatOut o) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= a.n || !flag[x]) return;
  const PatSpan s = pat_span(a, x);
  const uint32_t c = cidx[x];
  const uint8_t *p = a.xm + a.off[x];
  uint32_t meth = 0, total = 0;
  unsigned long long fnv = 14695981039346656037ull;
  for (uint32_t i = s.begin_i; i < s.end_i; i++) {
    const uint32_t base = p[i] & 15u;
    if (!((a.ctx_mask >> base) & 1u)) continue;
    const uint32_t pos = s.start_x + i - s.offset_x;
    const int64_t w = (int64_t)(int32_t)pos - a.pos_lo;
    if (w < 0 || w >= a.nwin) continue;
    const int32_t col = colmap[w];
    if (col < 0) continue;                                                   // :141
    o.cells[(int64_t)col * npat0 + c] = (int32_t)base;                       // :143
    meth += !(base & 8u);
    total++;
    fnv_char(fnv, pos);                                                      // :147
    fnv ^= (unsigned long long)base; fnv *= 1099511628211ull;                // :148
  }
  const bool nonempty = fnv != 14695981039346656037ull;
  if (nonempty) {
    static const uint8_t factor_map[16] = {13, 3, 4, 13, 11, 13, 13, 13, 12, 13, 13, 13, 13, 13, 13, 13};   // :47
    for (int32_t k = 0; k < nhlght; k++) {                                   // :154-164
      const uint32_t hp = (uint32_t)hlght[k] - s.start_x;
      if (hp >= s.begin_i && hp < s.end_i) {
        const uint32_t base = factor_map[(p[hp] >> 4) & 15u];
        o.cells[(int64_t)hcol[k] * npat0 + c] = (int32_t)base;
        fnv_char(fnv, (uint32_t)hlght[k]);
        fnv ^= (unsigned long long)base; fnv *= 1099511628211ull;
      }
    }
  }
  o.nonempty[c] = nonempty ? 1 : 0;
  o.strand[c] = a.strand[x];
  o.start[c] = (int32_t)(s.start_x + s.begin_i);
  o.end[c] = (int32_t)(s.start_x + s.end_i - 1u);
  o.nbase[c] = (int32_t)total;
  o.meth[c] = (int32_t)meth;
  o.fnv[c] = fnv;
}

}  // namespace epi

using namespace epi;

extern "C" {

void epi_pattern_table_free(epi_pattern_table *t) {
  if (!t) return;
  free(t->positions); free(t->strand); free(t->start); free(t->end); free(t->nbase); free(t->beta); free(t->fnv); free(t->cells);
  memset(t, 0, sizeof(*t));
}

int epi_batch_extract_patterns(epi_batch *b, int32_t target_rname, int32_t target_start, int32_t target_end, int32_t min_overlap,
                               const char *ctx, double min_ctx_freq, int32_t clip, int32_t reverse_offset, const int32_t *hlght,
                               int32_t nhlght, void *stream, epi_pattern_table *out) {
  if (!b || !ctx || !out || nhlght < 0 || (nhlght > 0 && !hlght)) return fail(EPI_ERR_ARG, "epi_batch_extract_patterns: bad arguments");
  memset(out, 0, sizeof(*out));
  if (b->n == 0) return EPI_OK;
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  EPI_TRY(fetch_row_stats(b, s));                           // the longest read bounds the window of positions
  if (b->h_stats.bad_len) return fail(EPI_ERR_ARG, "offsets are not non-decreasing, or start+length exceeds int32");
  const int64_t lmax = b->h_stats.max_len;

  PatArgs a;
  a.xm = b->xm; a.off = b->off; a.len = b->len; a.rname = b->rname; a.strand = b->strand; a.start = b->start; a.n = b->n;
  a.target_rname = (uint32_t)target_rname; a.target_start = (uint32_t)target_start; a.target_end = (uint32_t)target_end;
  a.reverse_offset = (uint32_t)reverse_offset; a.min_overlap = min_overlap; a.clip = clip ? 1 : 0;
  a.ctx_mask = 0;
  for (const unsigned char *c = reinterpret_cast<const unsigned char *>(ctx); *c; c++) a.ctx_mask |= 1u << ctx_to_idx(*c);
  a.pos_lo = (int64_t)target_start - lmax - (int64_t)reverse_offset - 2;
  a.nwin = ((int64_t)target_end - (int64_t)target_start) + 2 * lmax + (int64_t)reverse_offset + 8;
  if (a.nwin < 1) a.nwin = 1;
  if (a.nwin > (1LL << 31)) return fail(EPI_ERR_ARG, "epi_batch_extract_patterns: target too wide");

  const unsigned nb = (unsigned)((b->n + 255) / 256);
  DevBuf flag, cidx, cnt, colmap, d_hl, d_hc, outbuf, cells;
  auto cleanup = [&]() { flag.release(); cidx.release(); cnt.release(); colmap.release(); d_hl.release(); d_hc.release(); outbuf.release(); cells.release(); };
#define PAT_TRY(x) do { int rc_ = (x); if (rc_ != EPI_OK) { cleanup(); return rc_; } } while (0)
#define PAT_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(EPI_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
  PAT_TRY(flag.ensure((size_t)b->n * 4));
  PAT_TRY(cidx.ensure((size_t)b->n * 4));
  PAT_TRY(cnt.ensure((size_t)a.nwin * 4));
  PAT_TRY(colmap.ensure((size_t)a.nwin * 4));
  PAT_TRY(b->misc.ensure(256));
  uint32_t *d_total = b->misc.as<uint32_t>() + 15;          // misc[15]
  hipLaunchKernelGGL(k_pat_flag, dim3(nb), dim3(256), 0, s, a, flag.as<uint32_t>());
  PAT_TRY(scan_exclusive_u32(flag.as<uint32_t>(), cidx.as<uint32_t>(), b->n, d_total, b->scan_tmp, s));
  PAT_HIP(hipMemsetAsync(cnt.p, 0, (size_t)a.nwin * 4, s));
  hipLaunchKernelGGL(k_pat_count, dim3(nb), dim3(256), 0, s, a, flag.as<uint32_t>(), cnt.as<uint32_t>());
  PAT_HIP(hipGetLastError());
  uint32_t npat0 = 0;
  PAT_TRY(read_scalars(b, s, d_total, 4, &npat0));
  if (npat0 == 0) { cleanup(); return EPI_OK; }             // no read overlaps the target: empty table (:183)
  std::vector<uint32_t> h_cnt((size_t)a.nwin);
  PAT_HIP(hipMemcpy(h_cnt.data(), cnt.p, (size_t)a.nwin * 4, hipMemcpyDeviceToHost));

  // valid positions (:103-108), highlight positions (:110-112), merged in position order (:185)
  std::vector<int32_t> cols;
  for (int64_t w = 0; w < a.nwin; w++) {
    if (!h_cnt[(size_t)w]) continue;
    const int32_t pos = (int32_t)(a.pos_lo + w);
    if ((double)h_cnt[(size_t)w] / npat0 >= min_ctx_freq && std::find(hlght, hlght + nhlght, pos) == hlght + nhlght) cols.push_back(pos);
  }
  const size_t npatcols = cols.size();
  for (int32_t k = 0; k < nhlght; k++) cols.push_back(hlght[k]);
  std::vector<int32_t> patcols(cols.begin(), cols.begin() + (long)npatcols);
  std::sort(cols.begin(), cols.end());
  cols.erase(std::unique(cols.begin(), cols.end()), cols.end());       // std::map keys are unique
  const int32_t ncol = (int32_t)cols.size();
  std::vector<int32_t> h_colmap((size_t)a.nwin, -1), h_hcol((size_t)(nhlght > 0 ? nhlght : 1), 0);
  for (int32_t pos : patcols) h_colmap[(size_t)((int64_t)pos - a.pos_lo)] = (int32_t)(std::lower_bound(cols.begin(), cols.end(), pos) - cols.begin());
  for (int32_t k = 0; k < nhlght; k++) h_hcol[(size_t)k] = (int32_t)(std::lower_bound(cols.begin(), cols.end(), hlght[k]) - cols.begin());
  PAT_HIP(hipMemcpy(colmap.p, h_colmap.data(), (size_t)a.nwin * 4, hipMemcpyHostToDevice));
  PAT_TRY(d_hl.ensure((size_t)(nhlght > 0 ? nhlght : 1) * 4));
  PAT_TRY(d_hc.ensure((size_t)(nhlght > 0 ? nhlght : 1) * 4));
  if (nhlght > 0) {
    PAT_HIP(hipMemcpy(d_hl.p, hlght, (size_t)nhlght * 4, hipMemcpyHostToDevice));
    PAT_HIP(hipMemcpy(d_hc.p, h_hcol.data(), (size_t)nhlght * 4, hipMemcpyHostToDevice));
  }
  const size_t P0 = npat0;
  PAT_TRY(outbuf.ensure(P0 * (6 * 4 + 8) + 64));
  PAT_TRY(cells.ensure((size_t)(ncol > 0 ? ncol : 1) * P0 * 4));
  PatOut o;
  int32_t *ip = outbuf.as<int32_t>();
  o.fnv = reinterpret_cast<unsigned long long *>(ip);                  // 8-byte aligned first
  int32_t *q = ip + 2 * P0;
  o.nonempty = q; o.strand = q + P0; o.start = q + 2 * P0; o.end = q + 3 * P0; o.nbase = q + 4 * P0; o.meth = q + 5 * P0;
  o.cells = cells.as<int32_t>();
  PAT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cells.p), INT32_MIN, (size_t)(ncol > 0 ? ncol : 1) * P0, s));   // NA_INTEGER
  hipLaunchKernelGGL(k_pat_extract, dim3(nb), dim3(256), 0, s, a, flag.as<uint32_t>(), cidx.as<uint32_t>(), colmap.as<int32_t>(),
                     d_hl.as<int32_t>(), d_hc.as<int32_t>(), nhlght, (int64_t)P0, o);
  PAT_HIP(hipGetLastError());
  PAT_HIP(hipStreamSynchronize(s));
  std::vector<int32_t> h_i(6 * P0), h_cells((size_t)(ncol > 0 ? ncol : 1) * P0);
  std::vector<unsigned long long> h_f(P0);
  PAT_HIP(hipMemcpy(h_f.data(), o.fnv, P0 * 8, hipMemcpyDeviceToHost));
  PAT_HIP(hipMemcpy(h_i.data(), o.nonempty, 6 * P0 * 4, hipMemcpyDeviceToHost));
  PAT_HIP(hipMemcpy(h_cells.data(), cells.p, h_cells.size() * 4, hipMemcpyDeviceToHost));
  cleanup();
#undef PAT_TRY
#undef PAT_HIP

  // keep the non-empty patterns, in row order (:152, :166-176)
  size_t np = 0;
  for (size_t c = 0; c < P0; c++) np += h_i[c] != 0;
  if (np == 0) return EPI_OK;
  out->npat = (int64_t)np;
  out->ncol = ncol;
  out->positions = (int32_t *)malloc(((size_t)ncol + 1) * 4);
  out->strand = (int32_t *)malloc(np * 4); out->start = (int32_t *)malloc(np * 4); out->end = (int32_t *)malloc(np * 4);
  out->nbase = (int32_t *)malloc(np * 4); out->beta = (double *)malloc(np * 8); out->fnv = (uint64_t *)malloc(np * 8);
  out->cells = (int32_t *)malloc(((size_t)ncol * np + 1) * 4);
  if (!out->positions || !out->strand || !out->start || !out->end || !out->nbase || !out->beta || !out->fnv || !out->cells) {
    epi_pattern_table_free(out);
    return fail(EPI_ERR_NOMEM, "epi_batch_extract_patterns: out of host memory");
  }
  memcpy(out->positions, cols.data(), (size_t)ncol * 4);
  size_t w = 0;
  for (size_t c = 0; c < P0; c++) {
    if (!h_i[c]) continue;
    out->strand[w] = h_i[P0 + c]; out->start[w] = h_i[2 * P0 + c]; out->end[w] = h_i[3 * P0 + c];
    out->nbase[w] = h_i[4 * P0 + c];
    out->beta[w] = (double)(uint32_t)h_i[5 * P0 + c] / (uint32_t)h_i[4 * P0 + c];                 // :173
    out->fnv[w] = h_f[c];
    for (int32_t k = 0; k < ncol; k++) out->cells[(size_t)k * np + w] = h_cells[(size_t)k * P0 + c];
    w++;
  }
  return EPI_OK;
}

}  // extern "C"
