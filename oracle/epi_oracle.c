/*
 * oracle/epi_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the four hot-path functions of BBCG/epialleleR
 * (v1.13.4).  It is the checker for the HIP engine in epialleler_amd/csrc and
 * the "port" CPU baseline timed by bench.py.  Nothing in the product path
 * (epialleler_amd/, include/) may link, import or call this file: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Each function follows the reference's control flow step by step (same
 * per-base ordered-map emplace, same flush rule, same integer/floating-point
 * expressions) so that it also reproduces the reference on inputs the fast
 * engine rejects (e.g. unsorted rows).  Citations are file:line under the
 * reference's src/ directory.
 *
 * Parity pinning: the reference itself is NOT buildable in this image (it
 * needs Rcpp.h, Boost's flat_map.hpp and HTSlib, none of which are present,
 * and stand-in headers are not allowed), so there is no oracle/_ref.  This
 * restatement is pinned instead by every known-answer value the reference's
 * own RUnit tests hold for the path (tests/golden/expected.json, checked by
 * tests/test_oracle_golden.py on the reference's BAM fixtures).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* epialleleR.h:28 -- ctx_to_idx */
#define CTX_TO_IDX(c) (((((unsigned int)(unsigned char)(c)) + 2u) >> 2) & 15u)
/* epialleleR.h:38 -- unpack_ctx_idx */
#define UNPACK_CTX_IDX(b) ((unsigned int)(b) & 15u)

/* ------------------------------------------------------------------------ */
/* rcpp_threshold_reads.cpp:15-74                                            */
/* `templid` may be NULL (identity).  Output is R-logical style int32 0/1.   */
int orc_threshold_reads(const uint8_t *xm, const int64_t *off,
                        const int32_t *templid, int64_t n,
                        const char *ctx_meth, const char *ctx_unmeth,
                        const char *ooctx_meth, const char *ooctx_unmeth,
                        unsigned int min_n_ctx, double min_ctx_meth_frac,
                        double max_ooctx_meth_frac, int32_t *res)
{
  for (int64_t x = 0; x < n; x++) {                       /* :28 */
    res[x] = 0;                                           /* :27 default false */
    unsigned int ctx_map[16] = {0};                       /* :32 */
    const int64_t t = templid ? templid[x] : x;           /* :33 */
    const uint8_t *s = xm + off[t];
    const int64_t size_x = off[t + 1] - off[t];           /* :34 */
    for (int64_t i = 0; i < size_x; i++)                  /* :35-37 */
      ctx_map[UNPACK_CTX_IDX(s[i])]++;

    unsigned int n_ctx_meth = 0;                          /* :39-42 */
    for (const char *c = ctx_meth; *c; c++) n_ctx_meth += ctx_map[CTX_TO_IDX(*c)];
    if (n_ctx_meth == 0) continue;                        /* :43 */

    unsigned int n_ctx_unmeth = 0;                        /* :45-48 */
    for (const char *c = ctx_unmeth; *c; c++) n_ctx_unmeth += ctx_map[CTX_TO_IDX(*c)];
    unsigned int n_ctx_all = n_ctx_meth + n_ctx_unmeth;   /* :49 */
    if (n_ctx_all < min_n_ctx) continue;                  /* :50 */

    double ctx_meth_frac = (double)n_ctx_meth / n_ctx_all;        /* :52 */
    if (ctx_meth_frac < min_ctx_meth_frac) continue;              /* :53 */

    unsigned int n_ooctx_meth = 0;                        /* :55-58 */
    for (const char *c = ooctx_meth; *c; c++) n_ooctx_meth += ctx_map[CTX_TO_IDX(*c)];
    if (n_ooctx_meth > 0) {                               /* :59 */
      unsigned int n_ooctx_unmeth = 0;                    /* :60-63 */
      for (const char *c = ooctx_unmeth; *c; c++) n_ooctx_unmeth += ctx_map[CTX_TO_IDX(*c)];
      unsigned int n_ooctx_all = n_ooctx_meth + n_ooctx_unmeth;   /* :65 */
      double ooctx_meth_frac = (double)n_ooctx_meth / n_ooctx_all;/* :66 */
      if (ooctx_meth_frac > max_ooctx_meth_frac) continue;        /* :67 */
    }
    res[x] = 1;                                           /* :70 */
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* rcpp_get_xm_beta.cpp:10-43                                                */
int orc_get_xm_beta(const uint8_t *xm, const int64_t *off,
                    const int32_t *templid, int64_t n,
                    const char *ctx_meth, const char *ctx_unmeth, double *res)
{
  for (int64_t x = 0; x < n; x++) {                       /* :18 */
    unsigned int ctx_map[16] = {0};                       /* :22 */
    const int64_t t = templid ? templid[x] : x;
    const uint8_t *s = xm + off[t];
    const int64_t size_x = off[t + 1] - off[t];
    for (int64_t i = 0; i < size_x; i++)                  /* :25-27 */
      ctx_map[UNPACK_CTX_IDX(s[i])]++;
    unsigned int n_ctx_meth = 0, n_ctx_unmeth = 0;        /* :29-36 */
    for (const char *c = ctx_meth; *c; c++) n_ctx_meth += ctx_map[CTX_TO_IDX(*c)];
    for (const char *c = ctx_unmeth; *c; c++) n_ctx_unmeth += ctx_map[CTX_TO_IDX(*c)];
    unsigned int n_ctx_all = n_ctx_meth + n_ctx_unmeth;   /* :37 */
    if (n_ctx_all == 0) n_ctx_all = 1;                    /* :38 */
    res[x] = (double)n_ctx_meth / n_ctx_all;              /* :39 */
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Ordered unique map uint64 -> array<T,32>, the role boost::container::     */
/* flat_map plays in rcpp_cx_report.cpp:55 / rcpp_mhl_report.cpp:69: a       */
/* sorted vector with hinted unique insertion (search [hint,end) when        */
/* key >= *hint, else check the predecessor, else search [begin,hint)).      */
#define DEFINE_FMAP(NAME, VT)                                                  \
  typedef struct { uint64_t key; VT val[32]; } NAME##_ent;                     \
  typedef struct { NAME##_ent *e; size_t n, cap; } NAME;                       \
  static void NAME##_reserve(NAME *m, size_t c) {                              \
    if (c > m->cap) { m->e = (NAME##_ent *)realloc(m->e, c * sizeof(NAME##_ent)); m->cap = c; } \
  }                                                                            \
  static size_t NAME##_lower(const NAME *m, size_t lo, size_t hi, uint64_t k) {\
    while (lo < hi) { size_t mid = lo + ((hi - lo) >> 1);                      \
      if (m->e[mid].key < k) lo = mid + 1; else hi = mid; }                    \
    return lo;                                                                 \
  }                                                                            \
  static size_t NAME##_insert_at(NAME *m, size_t pos, uint64_t k, const VT *init) { \
    if (m->n == m->cap) NAME##_reserve(m, m->cap ? m->cap * 2 : 1024);         \
    memmove(m->e + pos + 1, m->e + pos, (m->n - pos) * sizeof(NAME##_ent));    \
    m->e[pos].key = k; memcpy(m->e[pos].val, init, 32 * sizeof(VT)); m->n++;   \
    return pos;                                                                \
  }                                                                            \
  /* try_emplace(hint, key, value): returns index of (new or existing) entry */\
  static size_t NAME##_try_emplace(NAME *m, size_t hint, uint64_t k, const VT *init) { \
    size_t lo, hi;                                                             \
    if (hint == m->n || k < m->e[hint].key) {                                  \
      if (hint == 0) return NAME##_insert_at(m, 0, k, init);                   \
      if (m->e[hint - 1].key < k) return NAME##_insert_at(m, hint, k, init);   \
      if (m->e[hint - 1].key == k) return hint - 1;                            \
      lo = 0; hi = hint - 1;                                                   \
    } else { lo = hint; hi = m->n; }                                           \
    size_t p = NAME##_lower(m, lo, hi, k);                                     \
    if (p < hi && m->e[p].key == k) return p;                                  \
    return NAME##_insert_at(m, p, k, init);                                    \
  }

DEFINE_FMAP(cxmap, int32_t)
DEFINE_FMAP(mhlmap, uint64_t)

typedef struct { int32_t *p; size_t n, cap; } ivec;
typedef struct { double *p; size_t n, cap; } dvec;
static void ivec_push(ivec *v, int32_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 4096; v->p = (int32_t *)realloc(v->p, v->cap * sizeof(int32_t)); }
  v->p[v->n++] = x;
}
static void dvec_push(dvec *v, double x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 4096; v->p = (double *)realloc(v->p, v->cap * sizeof(double)); }
  v->p[v->n++] = x;
}
static void ivec_resize_fill(ivec *v, size_t n, int32_t x) {   /* std::vector::resize(n, x) */
  if (n > v->cap) { v->cap = n; v->p = (int32_t *)realloc(v->p, v->cap * sizeof(int32_t)); }
  for (size_t i = v->n; i < n; i++) v->p[i] = x;
  v->n = n;
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* rcpp_cx_report.cpp:34-159.  pass may be NULL (all TRUE); NA (INT_MIN) is  */
/* non-zero and therefore counts as TRUE, as `!pass[x]` does (:118).         */
/* Outputs six malloc'd int32 columns (free with orc_free).                  */
int orc_cx_report(const uint8_t *xm, const int64_t *off, const int32_t *templid,
                  const int32_t *rname, const int32_t *strand, const int32_t *start,
                  const int32_t *pass, int64_t n, const char *ctx,
                  int64_t *nrow_out, int32_t **cols_out /* [6] */)
{
  unsigned int ctx_map[16] = {0};                         /* :88-91 */
  for (const char *c = ctx; *c; c++) ctx_map[CTX_TO_IDX(*c)] = 1;

  ivec res_rname = {0}, res_strand = {0}, res_pos = {0}, res_ctx = {0}, res_meth = {0}, res_unmeth = {0};
  cxmap cx_map = {0};
  size_t hint = 0;                                        /* :102 (== end of empty map) */
  int32_t map_val[32] = {0};                              /* :103 */
  int max_pos = 0;                                        /* :104 */
  unsigned int max_freq_idx, str_shft;
  cxmap_reserve(&cx_map, 100000);                         /* :107 */

#define CX_SPIT_RESULTS do {                                             /* :58-85 */ \
    for (size_t it = 0; it < cx_map.n; it++) {                                        \
      int32_t *second = cx_map.e[it].val;                                             \
      for (int s = 0; s < 2; s++) {                                                   \
        str_shft = (unsigned)s << 4;                                                  \
        if (second[9 + str_shft] == 0) continue;                             /* :62 */ \
        second[9 + str_shft] /= 2;                                           /* :63 */ \
        if (second[12 + str_shft] > second[9 + str_shft]) continue;          /* :64 */ \
        else if ((second[2 + str_shft] + second[10 + str_shft]) > second[9 + str_shft]) max_freq_idx = 2; \
        else if ((second[6 + str_shft] + second[14 + str_shft]) > second[9 + str_shft]) max_freq_idx = 6; \
        else if ((second[7 + str_shft] + second[15 + str_shft]) > second[9 + str_shft]) max_freq_idx = 7; \
        else continue;                                                       /* :71 */ \
        if (ctx_map[max_freq_idx]) {                                         /* :72 */ \
          ivec_push(&res_strand, s + 1);                                              \
          ivec_push(&res_pos, (int32_t)cx_map.e[it].key);                             \
          ivec_push(&res_ctx, (int32_t)max_freq_idx);                                 \
          ivec_push(&res_meth, second[max_freq_idx + str_shft]);                      \
          ivec_push(&res_unmeth, second[(max_freq_idx + str_shft) | 8]);              \
        }                                                                             \
      }                                                                               \
    }                                                                                 \
    ivec_resize_fill(&res_rname, res_strand.n, map_val[0]);                  /* :81 */ \
    max_pos = 0; cx_map.n = 0; hint = 0;                                  /* :82-84 */ \
  } while (0)

  for (int64_t x = 0; x < n; x++) {                       /* :108 */
    const int start_x = start[x];                         /* :112 */
    if ((start_x > max_pos) || (rname[x] != map_val[0])) {/* :113 */
      CX_SPIT_RESULTS;
      map_val[0] = rname[x];                              /* :115 */
    }
    str_shft = (unsigned)(strand[x] - 1) << 4;            /* :117 */
    const unsigned int pass_x = (pass ? (unsigned)(!pass[x]) : 0u) << 3;   /* :118 */
    const int64_t t = templid ? templid[x] : x;           /* :119 */
    const uint8_t *seqxm_x = xm + off[t];
    const unsigned int size_x = (unsigned int)(off[t + 1] - off[t]);       /* :120 */
    for (unsigned int i = 0; i < size_x; i++) {           /* :121 */
      const unsigned int idx_to_increase = UNPACK_CTX_IDX(seqxm_x[i]) | pass_x;   /* :122 */
      if (idx_to_increase == 11) continue;                /* :123 */
      map_val[1] = (int32_t)((unsigned int)start_x + i);  /* :124 */
      hint = cxmap_try_emplace(&cx_map, hint, (uint64_t)(int64_t)map_val[1], map_val);  /* :125 */
      cx_map.e[hint].val[idx_to_increase + str_shft]++;   /* :126 */
      cx_map.e[hint].val[9 + str_shft]++;                 /* :127 */
    }
    if (max_pos < map_val[1]) max_pos = map_val[1];       /* :129 */
  }
  CX_SPIT_RESULTS;                                        /* :131 */
#undef CX_SPIT_RESULTS

  free(cx_map.e);
  *nrow_out = (int64_t)res_strand.n;
  cols_out[0] = res_rname.p; cols_out[1] = res_strand.p; cols_out[2] = res_pos.p;
  cols_out[3] = res_ctx.p;   cols_out[4] = res_meth.p;   cols_out[5] = res_unmeth.p;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* rcpp_mhl_report.cpp:39-43 */
static uint64_t nrS(uint64_t n) { if (n < 2) return n; return (n * (n + 1) * (n + 2)) / 6; }

/* rcpp_mhl_report.cpp:46-228.  Outputs five int32 columns                   */
/* (rname,strand,pos,context,coverage) and two double columns (length,lmhl). */
/* Deviation (documented): the reference indexes a 65536-entry table with    */
/* mh_size / h_size unchecked (:169,:194); here indices >= 65536 clamp to    */
/* the last entry instead of reading out of bounds.                          */
int orc_mhl_report(const uint8_t *xm, const int64_t *off, const int32_t *templid,
                   const int32_t *rname, const int32_t *strand, const int32_t *start,
                   int64_t n, const char *ctx, int hmax, int hmin,
                   double max_ooctx_meth_frac,
                   int64_t *nrow_out, int32_t **icols_out /* [5] */, double **dcols_out /* [2] */)
{
  unsigned int ctx_map[16] = {0};                         /* :104-107 */
  for (const char *c = ctx; *c; c++) ctx_map[CTX_TO_IDX(*c)] = 1;

  const size_t mhl_lookup_len = 65536;                    /* :110-116 */
  uint64_t *mhl_lookup = (uint64_t *)calloc(mhl_lookup_len, sizeof(uint64_t));
  size_t hm = (hmax > 0) ? ((size_t)hmax < mhl_lookup_len ? (size_t)hmax : mhl_lookup_len) : mhl_lookup_len;
  for (size_t k = 0; k < hm; k++) mhl_lookup[k] = nrS(k);
  for (size_t k = hm; k < mhl_lookup_len; k++) mhl_lookup[k] = nrS(hm);
#define LOOKUP(i) (mhl_lookup[(i) < mhl_lookup_len ? (i) : mhl_lookup_len - 1])

  size_t num_buf_len = 8192;                              /* :119-120 */
  uint64_t *num_buf = (uint64_t *)malloc(num_buf_len * sizeof(uint64_t));

  ivec res_rname = {0}, res_strand = {0}, res_pos = {0}, res_ctx = {0}, res_cov = {0};
  dvec res_hlen = {0}, res_mhl = {0};
  mhlmap mhl_map = {0};
  size_t hint = 0;
  uint64_t map_val[32] = {0};                             /* :133 */
  int max_pos = 0;                                        /* :134 */
  unsigned int max_freq_idx, str_shft;
  mhlmap_reserve(&mhl_map, 100000);

#define MHL_SPIT_RESULTS do {                                            /* :72-101 */ \
    for (size_t it = 0; it < mhl_map.n; it++) {                                       \
      uint64_t *second = mhl_map.e[it].val;                                           \
      for (int s = 0; s < 2; s++) {                                                   \
        str_shft = (unsigned)s << 4;                                                  \
        if (second[9 + str_shft] == 0) continue;                                      \
        second[9 + str_shft] /= 2;                                                    \
        if (second[12 + str_shft] > second[9 + str_shft]) continue;                   \
        else if ((second[2 + str_shft] + second[10 + str_shft]) > second[9 + str_shft]) max_freq_idx = 2; \
        else if ((second[6 + str_shft] + second[14 + str_shft]) > second[9 + str_shft]) max_freq_idx = 6; \
        else if ((second[7 + str_shft] + second[15 + str_shft]) > second[9 + str_shft]) max_freq_idx = 7; \
        else continue;                                                                \
        if (ctx_map[max_freq_idx]) {                                                  \
          ivec_push(&res_strand, s + 1);                                              \
          ivec_push(&res_pos, (int32_t)mhl_map.e[it].key);                            \
          ivec_push(&res_ctx, (int32_t)max_freq_idx);                                 \
          const int cov = (int)(second[max_freq_idx + str_shft] + second[(max_freq_idx + str_shft) | 8]); /* :90 */ \
          ivec_push(&res_cov, cov);                                                   \
          dvec_push(&res_hlen, (double)second[8 + str_shft] / cov);          /* :92 */ \
          dvec_push(&res_mhl, (double)second[3 + str_shft] / second[4 + str_shft]); /* :93 */ \
        }                                                                             \
      }                                                                               \
    }                                                                                 \
    ivec_resize_fill(&res_rname, res_strand.n, (int32_t)map_val[0]);                  \
    max_pos = 0; mhl_map.n = 0; hint = 0;                                             \
  } while (0)

  for (int64_t x = 0; x < n; x++) {                       /* :138 */
    const int start_x = start[x];                         /* :142 */
    if ((start_x > max_pos) || ((uint64_t)(int64_t)rname[x] != map_val[0])) {   /* :143 */
      MHL_SPIT_RESULTS;
      map_val[0] = (uint64_t)(int64_t)rname[x];           /* :145 */
    }
    str_shft = (unsigned)(strand[x] - 1) << 4;            /* :147 */
    const int64_t t = templid ? templid[x] : x;
    const uint8_t *seqxm_x = xm + off[t];
    const unsigned int size_x = (unsigned int)(off[t + 1] - off[t]);

    if (num_buf_len < size_x) {                           /* :152-156 */
      num_buf_len = size_x;
      num_buf = (uint64_t *)realloc(num_buf, num_buf_len * sizeof(uint64_t));
    }
    memset(num_buf, 0, size_x * sizeof(uint64_t));        /* :157 */
    size_t mh_start = 0, mh_end = 0, mh_size = 0, h_size = 0;     /* :158 */
    size_t ooctx_map[16] = {0};                           /* :159 */
    for (unsigned int i = 0; i < size_x; i++) {           /* :160 */
      const unsigned int base_idx = UNPACK_CTX_IDX(seqxm_x[i]);   /* :161 */
      if (ctx_map[base_idx]) {                            /* :162 */
        h_size++;                                         /* :163 */
        if (base_idx < 8) {                               /* :164 */
          if (!mh_size) mh_start = i;                     /* :165 */
          mh_end = i;                                     /* :166 */
          mh_size++;                                      /* :167 */
        } else if (mh_size) {                             /* :168 */
          for (size_t k = mh_start; k <= mh_end; k++) num_buf[k] = LOOKUP(mh_size);   /* :169 */
          mh_size = 0;                                    /* :170 */
        }
      } else {
        ooctx_map[base_idx]++;                            /* :173 */
      }
    }
    size_t ooctx_meth = ooctx_map[2] + ooctx_map[5] + ooctx_map[6] + ooctx_map[7];          /* :176 */
    size_t ooctx_unmeth = ooctx_map[10] + ooctx_map[13] + ooctx_map[14] + ooctx_map[15];    /* :177 */
    double ooctx_meth_frac = (double)ooctx_meth / (double)(ooctx_meth + ooctx_unmeth);      /* :178 */
    if ((int)h_size < hmin || ooctx_meth_frac > max_ooctx_meth_frac) continue;              /* :179 */
    if (mh_size)                                          /* :180-182 */
      for (size_t k = mh_start; k <= mh_end; k++) num_buf[k] = LOOKUP(mh_size);

    for (unsigned int i = 0; i < size_x; i++) {           /* :185 */
      const unsigned int idx_to_increase = UNPACK_CTX_IDX(seqxm_x[i]);   /* :186 */
      if (idx_to_increase == 11) continue;                /* :187 */
      map_val[1] = (uint64_t)((unsigned int)start_x + i); /* :188 (unsigned int arithmetic, zero-extended) */
      hint = mhlmap_try_emplace(&mhl_map, hint, map_val[1], map_val);   /* :189 */
      uint64_t *second = mhl_map.e[hint].val;
      second[idx_to_increase + str_shft]++;               /* :190 */
      second[9 + str_shft]++;                             /* :191 */
      second[8 + str_shft] += h_size;                     /* :192 */
      second[3 + str_shft] += num_buf[i];                 /* :193 */
      second[4 + str_shft] += LOOKUP(h_size);             /* :194 */
    }
    if ((uint64_t)(int64_t)max_pos < map_val[1]) max_pos = (int)map_val[1];   /* :196 */
  }
  MHL_SPIT_RESULTS;                                       /* :198 */
#undef MHL_SPIT_RESULTS
#undef LOOKUP

  free(num_buf); free(mhl_lookup); free(mhl_map.e);
  *nrow_out = (int64_t)res_strand.n;
  icols_out[0] = res_rname.p; icols_out[1] = res_strand.p; icols_out[2] = res_pos.p;
  icols_out[3] = res_ctx.p;   icols_out[4] = res_cov.p;
  dcols_out[0] = res_hlen.p;  dcols_out[1] = res_mhl.p;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* rcpp_extract_patterns.cpp:26-211.                       */
/* The reference keeps positions in ordered maps; here the same order comes  */
/* from sorting.  Quirks kept as they are: with clip=TRUE the byte loop runs */
/* to `overlap`, not to begin+overlap (:86,:132); position bytes enter the   */
/* FNV-1a hash through a (signed) char pointer (:148,:163; epialleleR.h:8-13)*/
/* Outputs (malloc'ed, release with orc_free): per pattern strand, start,    */
/* end, nbase, beta, hash; the sorted column positions; cells[col*npat + p]  */
/* = context / base factor code or INT32_MIN (NA).                           */
static int cmp_i32(const void *a, const void *b) {
  const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
  return x < y ? -1 : x > y;
}

static void fnv_add_char(uint64_t *h, const void *p, unsigned size) {       /* pointer of type (signed) char */
  const signed char *c = (const signed char *)p;
  for (unsigned i = 0; i < size; i++) { *h ^= (uint64_t)(int64_t)c[i]; *h *= 1099511628211ull; }
}

int orc_extract_patterns(const uint8_t *xm, const int64_t *off, const int32_t *templid,
                         const int32_t *rname, const int32_t *strand, const int32_t *start, int64_t n,
                         unsigned int target_rname, unsigned int target_start, unsigned int target_end,
                         int min_overlap, const char *ctx, double min_ctx_freq, int clip,
                         unsigned int reverse_offset, const int32_t *hlght, int32_t nhlght,
                         int64_t *npat_out, int32_t *ncol_out, int32_t **o_strand, int32_t **o_start, int32_t **o_end,
                         int32_t **o_nbase, double **o_beta, uint64_t **o_fnv, int32_t **o_pos, int32_t **o_cells)
{
  static const unsigned int factor_map[16] = { 13, 3, 4, 13, 11, 13, 13, 13, 12, 13, 13, 13, 13, 13, 13, 13 };   /* :47 */
  unsigned int ctx_map[16] = {0};                                           /* :61-64 */
  for (const char *c = ctx; *c; c++) ctx_map[CTX_TO_IDX(*c)] = 1;
  /* first pass (:74-100): how often is every in-context position seen; npat = reads that overlap the target */
  ivec allpos; memset(&allpos, 0, sizeof(allpos));
  unsigned int npat = 0;
  for (int64_t x = 0; x < n; x++) {
    if (rname[x] != (int)target_rname) continue;
    const int64_t t = templid ? templid[x] : x;
    const unsigned int size_x = (unsigned int)(off[t + 1] - off[t]);
    const unsigned int start_x = (unsigned int)start[x];
    const unsigned int end_x = start_x + size_x - 1;
    const unsigned int over_start_x = start_x > target_start ? start_x : target_start;
    const unsigned int over_end_x = end_x < target_end ? end_x : target_end;
    const int overlap = (int)(over_end_x - over_start_x + 1);
    if (overlap >= min_overlap) {
      const uint8_t *s = xm + off[t];
      const unsigned int offset_x = strand[x] == 2 ? reverse_offset : 0;
      const unsigned int begin_i = clip ? (over_start_x - start_x) : 0;
      const unsigned int end_i = clip ? (unsigned int)overlap : size_x;
      for (unsigned int i = begin_i; i < end_i; i++)
        if (ctx_map[UNPACK_CTX_IDX(s[i])]) ivec_push(&allpos, (int32_t)(start_x + i - offset_x));
      npat++;
    }
  }
  qsort(allpos.p, allpos.n, sizeof(int32_t), cmp_i32);
  /* valid positions (:103-108), highlight positions (:110-112), merged and ordered (:185) */
  ivec cols; memset(&cols, 0, sizeof(cols));
  for (size_t i = 0; i < allpos.n;) {
    size_t j = i;
    while (j < allpos.n && allpos.p[j] == allpos.p[i]) j++;
    int is_h = 0;
    for (int32_t k = 0; k < nhlght; k++) if (hlght[k] == allpos.p[i]) is_h = 1;
    if ((double)(j - i) / npat >= min_ctx_freq && !is_h) ivec_push(&cols, allpos.p[i]);
    i = j;
  }
  const size_t npatcols = cols.n;
  for (int32_t k = 0; k < nhlght; k++) ivec_push(&cols, hlght[k]);
  /* column kind before sorting: remember which positions are pattern columns */
  int32_t *patcols = (int32_t *)malloc((npatcols + 1) * sizeof(int32_t));
  memcpy(patcols, cols.p, npatcols * sizeof(int32_t));
  qsort(cols.p, cols.n, sizeof(int32_t), cmp_i32);
  const size_t ncol = cols.n;
  const size_t cap = npat ? npat : 1;
  int32_t *cells = (int32_t *)malloc((ncol * cap + 1) * sizeof(int32_t));
  for (size_t i = 0; i < ncol * cap; i++) cells[i] = INT32_MIN;
  ivec st, sa, en, nb; dvec be;
  memset(&st, 0, sizeof(st)); memset(&sa, 0, sizeof(sa)); memset(&en, 0, sizeof(en)); memset(&nb, 0, sizeof(nb)); memset(&be, 0, sizeof(be));
  uint64_t *fnvs = (uint64_t *)malloc((cap + 1) * sizeof(uint64_t));
  unsigned int np2 = 0;
  for (int64_t x = 0; x < n; x++) {                                         /* second pass (:115-181) */
    if (rname[x] != (int)target_rname) continue;
    const int64_t t = templid ? templid[x] : x;
    const unsigned int size_x = (unsigned int)(off[t + 1] - off[t]);
    const unsigned int start_x = (unsigned int)start[x];
    const unsigned int end_x = start_x + size_x - 1;
    const unsigned int over_start_x = start_x > target_start ? start_x : target_start;
    const unsigned int over_end_x = end_x < target_end ? end_x : target_end;
    const int overlap = (int)(over_end_x - over_start_x + 1);
    if (overlap < min_overlap) continue;
    const uint8_t *s = xm + off[t];
    const unsigned int offset_x = strand[x] == 2 ? reverse_offset : 0;
    const unsigned int begin_i = clip ? (over_start_x - start_x) : 0;
    const unsigned int end_i = clip ? (unsigned int)overlap : size_x;
    unsigned int meth = 0, total = 0;
    uint64_t fnv = 14695981039346656037ull;
    for (unsigned int i = begin_i; i < end_i; i++) {
      const unsigned int base = UNPACK_CTX_IDX(s[i]);
      if (!ctx_map[base]) continue;
      const unsigned int pos = start_x + i - offset_x;
      const int32_t key = (int32_t)pos;
      const int32_t *hit = (const int32_t *)bsearch(&key, patcols, npatcols, sizeof(int32_t), cmp_i32);
      if (!hit) continue;                                                   /* :141 */
      const int32_t *col = (const int32_t *)bsearch(&key, cols.p, ncol, sizeof(int32_t), cmp_i32);
      cells[(size_t)(col - cols.p) * cap + np2] = (int32_t)base;            /* :143 */
      meth += !(base & 8);
      total++;
      fnv_add_char(&fnv, &pos, sizeof(pos));                                /* :147 */
      fnv ^= (uint64_t)base; fnv *= 1099511628211ull;                       /* :148: *(const unsigned int*)&base, one step */
    }
    if (fnv == 14695981039346656037ull) continue;                           /* :152 */
    for (int32_t k = 0; k < nhlght; k++) {                                  /* :154-164 */
      const unsigned int hp = (unsigned int)hlght[k] - start_x;
      if (hp >= begin_i && hp < end_i) {
        const unsigned int base = factor_map[(s[hp] >> 4) & 15u];
        const int32_t key = hlght[k];
        const int32_t *col = (const int32_t *)bsearch(&key, cols.p, ncol, sizeof(int32_t), cmp_i32);
        cells[(size_t)(col - cols.p) * cap + np2] = (int32_t)base;
        fnv_add_char(&fnv, &hlght[k], sizeof(int32_t));
        fnv ^= (uint64_t)base; fnv *= 1099511628211ull;
      }
    }
    ivec_push(&st, strand[x]);
    ivec_push(&sa, (int32_t)(start_x + begin_i));
    ivec_push(&en, (int32_t)(start_x + end_i - 1));
    ivec_push(&nb, (int32_t)total);
    dvec_push(&be, (double)meth / total);
    fnvs[np2] = fnv;
    np2++;
  }
  /* compact the cell matrix to np2 patterns per column (:187-189) */
  int32_t *out = (int32_t *)malloc((ncol * (size_t)(np2 ? np2 : 1) + 1) * sizeof(int32_t));
  for (size_t c = 0; c < ncol; c++)
    for (unsigned int p = 0; p < np2; p++) out[c * np2 + p] = cells[c * cap + p];
  free(cells); free(patcols); free(allpos.p);
  *npat_out = np2;
  *ncol_out = np2 ? (int32_t)ncol : 0;
  *o_strand = st.p; *o_start = sa.p; *o_end = en.p; *o_nbase = nb.p; *o_beta = be.p; *o_fnv = fnvs; *o_pos = cols.p; *o_cells = out;
  return 0;
}
