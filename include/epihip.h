/*
 * epihip.h -- C ABI of the MI355X-native per-read methylation-call aggregation
 * engine (libepihip.so).  Plain pointers and sizes only; no C++/torch types.
 *
 * The four "drop-in" entry points replace, one for one, the native functions
 * behind the reference's .Call stubs (BBCG/epialleleR v1.13.4):
 *
 *   epi_threshold_reads  <-  rcpp_threshold_reads  src/rcpp_threshold_reads.cpp:15-74
 *                            (.Call "_epialleleR_rcpp_threshold_reads", R/RcppExports.R:64-66)
 *   epi_get_xm_beta      <-  rcpp_get_xm_beta      src/rcpp_get_xm_beta.cpp:10-43
 *                            (.Call "_epialleleR_rcpp_get_xm_beta",     R/RcppExports.R:28-30)
 *   epi_cx_report        <-  rcpp_cx_report        src/rcpp_cx_report.cpp:34-159
 *                            (.Call "_epialleleR_rcpp_cx_report",       R/RcppExports.R:12-14)
 *   epi_mhl_report       <-  rcpp_mhl_report       src/rcpp_mhl_report.cpp:46-228
 *                            (.Call "_epialleleR_rcpp_mhl_report",      R/RcppExports.R:40-42)
 *
 * Input layout (what an R/Rcpp shim gathers from the data.frame + seqxm_xptr,
 * see INTEGRATION.md): templates in ROW order (i.e. already sorted by
 * (rname,start) as .readBam leaves them, R/internal.R:193-195):
 *   xm[off[n]]   packed SEQXM bytes, (nt16<<4)|ctx_idx, src/epialleleR.h:28-38
 *   off[n+1]     int64 byte offsets, row x owns xm[off[x] .. off[x+1])
 *   rname[n], strand[n] (1='+',2='-'), start[n] (1-based)   int32, R factor codes
 *   pass[n]      R logical (int32; 0 = FALSE, anything else incl. NA = TRUE)
 *
 * The resident API (epi_batch_*) is the same computation on a batch that stays
 * in HBM between calls -- the analogue of reusing a preprocessBam() object for
 * several reports (R/preprocessBam.R:6-13).
 *
 * Every function returns 0 on success or an EPI_ERR_* code; epi_last_error()
 * gives the message (thread-local).  HIP failures surface this way, never by
 * abort().  There is NO CPU fallback: without the HIP runtime and a gfx950
 * device every compute entry point fails with EPI_ERR_NODEVICE.
 */
#ifndef EPIHIP_H
#define EPIHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPI_OK            0
#define EPI_ERR_ARG       1   /* invalid argument                                   */
#define EPI_ERR_HIP       2   /* HIP runtime error (message has the hipError name)  */
#define EPI_ERR_UNSORTED  3   /* rows not sorted by (rname,start): "PRE-SORTED DATASET IS A REQUIREMENT", rcpp_cx_report.cpp:19 */
#define EPI_ERR_NOMEM     4
#define EPI_ERR_NODEVICE  5   /* no usable GPU                                      */
#define EPI_ERR_STATE     6   /* call sequence error (e.g. fetch before report)     */

const char *epi_last_error(void);
int epi_version(void);

/* Result tables: library-owned host arrays, release with the matching *_free.
 * Factor levels are the reference's: strand {"+","-"}; context
 * c("NA1","CHH","NA3","NA4","NA5","CHG","CG") i.e. 2=CHH 6=CHG 7=CG
 * (rcpp_cx_report.cpp:146-155); rname levels are the caller's. */
typedef struct {
  int64_t nrow;
  int32_t *rname, *strand, *pos, *context, *meth, *unmeth;
} epi_cx_table;

typedef struct {
  int64_t nrow;
  int32_t *rname, *strand, *pos, *context, *coverage;
  double *length, *lmhl;
} epi_mhl_table;

/* (The columns of a table are one allocation: release a table only through these, never column by column.) */
void epi_cx_table_free(epi_cx_table *t);
void epi_mhl_table_free(epi_mhl_table *t);

/* ---- drop-in entry points (host pointers in, host results out) ---------- */

int epi_threshold_reads(const uint8_t *xm, const int64_t *off, int64_t n,
                        const char *ctx_meth, const char *ctx_unmeth,
                        const char *ooctx_meth, const char *ooctx_unmeth,
                        uint32_t min_n_ctx, double min_ctx_meth_frac,
                        double max_ooctx_meth_frac, int32_t *pass_out /* [n] */);

int epi_get_xm_beta(const uint8_t *xm, const int64_t *off, int64_t n,
                    const char *ctx_meth, const char *ctx_unmeth,
                    double *beta_out /* [n] */);

int epi_cx_report(const uint8_t *xm, const int64_t *off, const int32_t *rname,
                  const int32_t *strand, const int32_t *start,
                  const int32_t *pass /* may be NULL = all TRUE */, int64_t n,
                  const char *ctx, epi_cx_table *out);

int epi_mhl_report(const uint8_t *xm, const int64_t *off, const int32_t *rname,
                   const int32_t *strand, const int32_t *start, int64_t n,
                   const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac,
                   epi_mhl_table *out);

/* ---- host-side producer (preprocessBam) ----------------------------------
 * BAM file -> packed templates sorted by (rname,start), as SoA host buffers (xm in
 * pinned memory when a HIP device is usable).  Replaces rcpp_check_bam
 * (src/rcpp_check_bam.cpp:19-60 + .checkBam, R/internal.R:75-128),
 * rcpp_read_bam_paired / rcpp_read_bam_single (src/rcpp_read_bam.cpp:19-343) and the
 * templid/sort step of .readBam (R/internal.R:154-199) for short-read XG/XM BAMs;
 * BGZF/BAM are decoded with zlib only (no HTSlib).  Same defaults and error
 * conditions as preprocessBam() (R/preprocessBam.R:197-237). */
typedef struct {
  int32_t min_mapq, min_baseq;
  int32_t skip_duplicates, skip_secondary, skip_qcfail, skip_supplementary;   /* R defaults: 0,1,1,1 */
  int32_t trim5, trim3;
  int32_t paired;      /* -1 = detect as .checkBam does; 0/1 = expected endness (error if different) */
  int32_t nthreads;    /* BGZF inflate threads (>=1) */
  /* long-read (MM/ML) alignments only, rcpp_read_bam_mm_single (src/rcpp_read_bam.cpp:364-372): */
  int32_t min_prob;    /* minimum ML probability of a 5mC call (R default -1)                            */
  int32_t highest_prob;/* the 5mC probability must be the highest of all modifications at the base (TRUE) */
  int32_t window_kib;  /* inflated KiB processed per pass (0 = 262144): bounds the host memory next to the output */
} epi_bam_options;

typedef struct {       /* library-owned; release with epi_templates_free */
  int64_t n, nbytes, xm_capacity, nrecs;
  uint8_t *xm;         /* [xm_capacity] packed SEQXM bytes of all templates in row order, 0xFB padded */
  int64_t *off;        /* [n+1] */
  int32_t *rname, *strand, *start;   /* [n] R factor codes / 1-based start */
  int32_t n_targets;
  char **target_names; /* rname levels: all BAM header targets (src/rcpp_read_bam.cpp:173-179) */
  int32_t paired, pinned;
} epi_templates;

int epi_preprocess_bam(const char *path, const epi_bam_options *opt /* NULL = R defaults */, epi_templates *out);
void epi_templates_free(epi_templates *t);

/* ---- report writer (.writeReport, R/internal.R:274-287) -------------------
 * The table as a tab-separated file with a header line, what data.table::fwrite(report, quote=FALSE, sep="\t",
 * col.names=TRUE, compress=if (gzip) "gzip" else "none") writes: integers in decimal, factor columns as their
 * labels, doubles with up to 15 significant digits; NA_integer_ (INT32_MIN), factor codes outside 1..nlevels and
 * NaN are empty fields (na = "").  Rows are formatted by `nthreads` host threads; with gzip != 0 the file is a
 * multi-member gzip file. */
enum { EPI_COL_I32 = 0, EPI_COL_F64 = 1, EPI_COL_FACTOR = 2 };
typedef struct {
  const char *name;            /* header field */
  int32_t kind;                /* EPI_COL_* */
  const void *data;            /* int32_t[nrow] (I32, FACTOR: 1-based codes) or double[nrow] (F64) */
  const char *const *levels;   /* FACTOR: labels */
  int32_t nlevels;
} epi_report_column;
int epi_write_report(const char *path, const epi_report_column *cols, int32_t ncol, int64_t nrow, int32_t gzip,
                     int32_t nthreads);

/* ---- resident API -------------------------------------------------------- */

typedef struct epi_engine epi_engine;   /* one per GPU: device id, streams, pinned staging */
typedef struct epi_batch epi_batch;     /* templates resident in HBM + reusable workspace  */

int epi_engine_create(int device, epi_engine **out);
void epi_engine_destroy(epi_engine *e);
int epi_engine_device(const epi_engine *e);

/* Host SoA -> HBM through two pinned staging buffers and hipMemcpyAsync
 * (double-buffered).  The batch owns its device memory. */
int epi_batch_upload(epi_engine *e, const uint8_t *xm, const int64_t *off,
                     const int32_t *rname, const int32_t *strand, const int32_t *start,
                     int64_t n, epi_batch **out);

/* The same four computations on a batch that is already resident, host results out: what a binding without device
 * memory of its own (the Rcpp shim) calls after ONE epi_batch_upload per preprocessBam()
 * object.  They run on the engine's own stream and return when the results are in the caller's buffers.
 * epi_batch_cytosine_report is generateCytosineReport(threshold.reads=TRUE) in one pass over the bytes
 * (epi_batch_cytosine_report_dev below); pass_out is optional. */
int epi_batch_threshold_reads(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                              const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                              double max_ooctx_meth_frac, int32_t *pass_out /* [n] */);
int epi_batch_get_xm_beta(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, double *beta_out /* [n] */);
int epi_batch_cx_report(epi_batch *b, const int32_t *pass /* host, may be NULL = all TRUE */, const char *ctx, epi_cx_table *out);
int epi_batch_cytosine_report(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                              const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                              double max_ooctx_meth_frac, const char *ctx, int32_t *pass_out /* host, may be NULL */,
                              epi_cx_table *out);
int epi_batch_mhl_report(epi_batch *b, const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac, epi_mhl_table *out);
/* The same reports in two steps, for a binding that owns the result vectors (R's IntegerVector / NumericVector,
 * src/rcpp_cx_report.cpp:133-140, src/rcpp_mhl_report.cpp:200-208): *_begin runs the report on the engine's stream and
 * returns the row count; the caller allocates its columns and epi_batch_cx_fetch_host / epi_batch_mhl_fetch_host (stream
 * NULL) copy the table straight into them -- no library-owned table, no second host copy. */
int epi_batch_cx_report_begin(epi_batch *b, const int32_t *pass /* host, may be NULL */, const char *ctx, int64_t *nrow_out);
int epi_batch_cytosine_report_begin(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                                    const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                                    double max_ooctx_meth_frac, const char *ctx, int32_t *pass_out /* host, may be NULL */,
                                    int64_t *nrow_out);
int epi_batch_mhl_report_begin(epi_batch *b, const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac, int64_t *nrow_out);
/* the lazily created engine the host-pointer entry points use (device EPIHIP_DEVICE, default 0) */
int epi_default_engine(epi_engine **out);

/* Zero-copy: the caller (e.g. a torch tensor) owns the device buffers, keeps
 * them alive and does not change them while the batch exists.  d_xm must be
 * 16-byte aligned and xm_capacity (bytes allocated) >= off[n] rounded up to
 * 16.  nbytes = off[n].
 * Both constructors queue, on the HIP null stream, one pass over the columns
 * (longest read, (rname,start) order, strand and offset validity): the data
 * must be complete as seen from that stream.  Its verdict is raised by the
 * first report call (EPI_ERR_UNSORTED / EPI_ERR_ARG); per-read functions
 * accept unsorted rows. */
int epi_batch_adopt(epi_engine *e, const uint8_t *d_xm, int64_t xm_capacity, int64_t nbytes,
                    const int64_t *d_off, const int32_t *d_rname, const int32_t *d_strand,
                    const int32_t *d_start, int64_t n, epi_batch **out);
/* Position-congruent rows.  The tile kernels read position-aligned 16-byte chunks; with rows back to back those have
 * any byte alignment and load at ~0.85 of the aligned rate.  epi_batch_realign gives the batch its OWN copy of xm in
 * which row x starts at an offset = start[x] (mod 16) (<= 15 bytes of filler between rows, one pass over the bytes, one
 * host synchronisation) -- the reference keeps one std::string per template (src/epialleleR.h:28-38), so where rows
 * start is the engine's business.  epi_batch_upload does this itself; for an adopted batch it is the caller's call:
 * afterwards the batch no longer reads d_xm / d_off (the caller may free them), and holds B + <= 15 n bytes of its own.
 * Call it before the first report on the batch.  Results never depend on it.  EPIHIP_REALIGN=0 makes it a no-op
 * (A/B runs); 4 / 8 = congruent modulo 4 / 8 only.  epi_batch_layout: 0 = as given, 4 / 16 = congruent modulo that. */
int epi_batch_realign(epi_batch *b, void *stream);
int epi_batch_layout(const epi_batch *b);
/* The rows as the kernels read them: row x owns d_xm[d_off[x] .. d_off[x] + d_len[x]) (d_off: n + 1 non-decreasing
 * entries, d_off[n] = *nbytes; d_len: n; written by a kernel queued on the null stream when the batch was created).
 * Device pointers, valid until the batch is freed or realigned.  Any out-pointer may be NULL. */
int epi_batch_view(const epi_batch *b, const uint8_t **d_xm, const int64_t **d_off, const int32_t **d_len, int64_t *nbytes);
void epi_batch_free(epi_batch *b);
int64_t epi_batch_nrows(const epi_batch *b);

/* `stream` is a hipStream_t; NULL is the HIP null stream (e.g. torch's default
 * stream), so the call is ordered after whatever the caller queued there.  The
 * *_dev functions are asynchronous on that stream unless noted. */
int epi_batch_threshold_reads_dev(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth,
                                  const char *ooctx_meth, const char *ooctx_unmeth,
                                  uint32_t min_n_ctx, double min_ctx_meth_frac,
                                  double max_ooctx_meth_frac, int32_t *d_pass_out, void *stream);
int epi_batch_get_xm_beta_dev(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth,
                              double *d_beta_out, void *stream);

/* rcpp_match_amplicon / rcpp_match_capture (src/rcpp_match_target.cpp:16-81; callers .getBedReport /
 * .getBedEcdf, R/internal.R:529-604): first BED row a read matches, 1-based, or INT32_MIN (NA_integer_).
 * bed_chr are rname factor codes; capture = 0: start or end within `param` (tolerance);
 * capture = 1: overlap >= `param`. */
int epi_batch_match_target_dev(epi_batch *b, const int32_t *d_bed_chr, const int32_t *d_bed_start,
                               const int32_t *d_bed_end, int32_t nbed, int32_t capture, int32_t param,
                               int32_t *d_match_out, void *stream);

/* CX report in two steps so the caller can allocate the output columns:
 *  1) compute: tile index, LDS-histogram tile kernel, majority rule, ordered
 *     row offsets.  Synchronises `stream` (row count comes back to the host).
 *  2) fetch: gathers the rows, in reference order, into six int32 columns of
 *     length nrow in device (fetch_dev, async) or host (fetch_host) memory. */
int epi_batch_cx_report_dev(epi_batch *b, const int32_t *d_pass /* NULL = all TRUE */,
                            const char *ctx, void *stream, int64_t *nrow_out);
/* generateCytosineReport(threshold.reads=TRUE) in one call (R/generateCytosineReport.R:181-199: .thresholdReads, then
 * .getCytosineReport with its result): the same table as epi_batch_threshold_reads_dev followed by
 * epi_batch_cx_report_dev, but the bytes are read from HBM once -- the tile kernel counts the thresholding classes of
 * a read from the registers it already holds and lower-cases the read's calls itself (rcpp_threshold_reads.cpp:28-71,
 * rcpp_cx_report.cpp:118).  Fused for reads of up to ~2.5 kb and class strings without repeated letters; other
 * batches run the two kernels one after the other (same results).  d_pass_out (optional, [n]) receives the pass
 * flags.  Continue with epi_batch_cx_fetch_* (or epi_batch_cx_finish_shared) as after epi_batch_cx_report_dev. */
int epi_batch_cytosine_report_dev(epi_batch *b, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                                  const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac,
                                  double max_ooctx_meth_frac, const char *ctx, int32_t *d_pass_out /* may be NULL */,
                                  void *stream, int64_t *nrow_out);
int epi_batch_cx_fetch_dev(epi_batch *b, int32_t *const d_cols[6], void *stream);
int epi_batch_cx_fetch_host(epi_batch *b, int32_t *const h_cols[6], void *stream);

int epi_batch_mhl_report_dev(epi_batch *b, const char *ctx, int hmax, int hmin,
                             double max_ooctx_meth_frac, void *stream, int64_t *nrow_out);
int epi_batch_mhl_fetch_dev(epi_batch *b, int32_t *const d_icols[5], double *const d_dcols[2], void *stream);
int epi_batch_mhl_fetch_host(epi_batch *b, int32_t *const h_icols[5], double *const h_dcols[2], void *stream);

/* rcpp_extract_patterns (src/rcpp_extract_patterns.cpp:26-211; caller .getPatterns, R/internal.R:683-714):
 * methylation patterns of the reads overlapping one target.  Library-owned host table: per pattern strand, start,
 * end, nbase, beta, the FNV-1a hash the R side prints as 16 hex digits ("pattern"), the ordered column positions
 * and cells[col * npat + p] = context index / base factor code (levels as :192-195) or INT32_MIN (NA).
 * npat = 0 is the reference's empty data frame.  Synchronises `stream`. */
typedef struct {
  int64_t npat;
  int32_t ncol;
  int32_t *positions;                       /* [ncol] */
  int32_t *strand, *start, *end, *nbase;    /* [npat] */
  double *beta;                             /* [npat] */
  uint64_t *fnv;                            /* [npat] */
  int32_t *cells;                           /* [ncol][npat] */
} epi_pattern_table;
int epi_batch_extract_patterns(epi_batch *b, int32_t target_rname, int32_t target_start, int32_t target_end,
                               int32_t min_overlap, const char *ctx, double min_ctx_freq, int32_t clip,
                               int32_t reverse_offset, const int32_t *hlght /* sorted, unique, inside the target */,
                               int32_t nhlght, void *stream, epi_pattern_table *out);
void epi_pattern_table_free(epi_pattern_table *t);

/* ---- multi-GPU (row-range shards; see DESIGN.md "Multi-GPU") -------------
 * Tiles are cut on an absolute position grid (epi_tile_positions() wide), so
 * ranks agree on tile boundaries.  A rank (a) reports the range of tile keys
 * its rows touch, (b) is told which keys are shared with other ranks; shared
 * tiles are accumulated into a dense counter slab instead of being emitted,
 * (c) the caller sum-reduces the slabs across ranks (RCCL all-reduce), and
 * (d) the owning rank emits them.  key = ((int64)rname << 32) | biased tile.  Slab planes of one tile: (n, M) per
 * strand and reported context, then the two strands' coverage difference arrays (cx_report.hip). */
int epi_tile_positions(void);                  /* CX tile size for a single-context report (e.g. "Z") */
int epi_cx_tile_positions(const char *ctx);    /* ... for this context string: 2048 with one reported context, else 1024 */
int epi_batch_tile_key_range(epi_batch *b, void *stream, int64_t *first_key, int64_t *last_key);
int epi_batch_cx_set_shared(epi_batch *b, const int64_t *h_keys, const int32_t *h_owned,
                            int32_t nshared, int32_t *d_slab /* [nshared][16][T] int32, zeroed by caller */);
/* With shared tiles set, epi_batch_cx_report_dev stops after accumulation
 * (nrow_out = 0); all-reduce the slab, then finish: */
int epi_batch_cx_finish_shared(epi_batch *b, const char *ctx, void *stream, int64_t *nrow_out);

/* The same exchange for the lMHL table (tiles of epi_mhl_tile_positions() positions): shared tiles
 * hand over their counters [nshared][16][T] (int32) and their numerator sums / interval arrays
 * [nshared][epi_mhl_slab_sums()] (int64, wrap-around arithmetic); all-reduce both, then finish. */
int epi_mhl_tile_positions(void);
int epi_mhl_slab_sums(void);
int epi_batch_tile_key_range_for(epi_batch *b, int tile_positions, void *stream, int64_t *first_key, int64_t *last_key);
int epi_batch_mhl_set_shared(epi_batch *b, const int64_t *h_keys, const int32_t *h_owned, int32_t nshared,
                             int32_t *d_cnt_slab, int64_t *d_sum_slab);
int epi_batch_mhl_finish_shared(epi_batch *b, void *stream, int64_t *nrow_out);
/* The one-pass lMHL kernel shards as well: tiles of epi_mhl_fused_tile_positions() positions, slabs int32
 * [nshared][4][T] (calls of the context '+', '-'; coverage differences '+', '-') and int64 [nshared][6][T] (difference
 * arrays of the three sums, two strands each).  Every rank must take the same path: epi_batch_mhl_fused_ok says whether
 * THIS rank's rows allow the one-pass kernel for `ctx` (one haplotype context, reads of at most 4 KiB); the caller
 * combines the answers (distributed.py: all ranks, once per shard) and attaches the slabs with the matching call.
 * epi_batch_mhl_report_dev and epi_batch_mhl_finish_shared are then used as for the two-kernel layout. */
int epi_mhl_fused_tile_positions(void);
int epi_batch_mhl_fused_ok(epi_batch *b, const char *ctx, void *stream, int32_t *ok_out);
int epi_batch_mhl_set_shared_fused(epi_batch *b, const int64_t *h_keys, const int32_t *h_owned, int32_t nshared,
                                   int32_t *d_cnt_slab, int64_t *d_sum_slab);

/* ---- the same exchange with RCCL called by the library (csrc/comm.hip) -------------------------------------------
 * For hosts without a collective library of their own (the R shim): one epi_comm per process
 * and GPU, one call per report.  The 128-byte id comes from ONE rank (epi_comm_unique_id = ncclGetUniqueId) and travels
 * to the others by whatever the host has; epi_comm_create is collective (ncclCommInitRank).  Every rank then calls the
 * sharded entry point on its own batch -- a contiguous range of the globally (rname, start)-sorted rows, ranges in rank
 * order: tile key ranges are all-gathered (once per batch and tile grid), shared tiles accumulate into a slab owned by
 * the batch, ncclAllReduce runs on the report's stream, owners emit.  nrow_out = THIS rank's rows; the reference's table
 * is the ranks' tables in rank order (epi_batch_cx_fetch_* / epi_batch_mhl_fetch_* as after a single-GPU report).
 * Replaces, for the sharded path, R/generateCytosineReport.R:181-203 and R/generateMhlReport.R:185-196 run on the whole
 * data set.  ctx_meth == NULL: no thresholding (d_pass: per-row flags in device memory or NULL = all TRUE). */
/* Step 2 on its own (pure host logic, no device needed): ranges = (first, last) tile key per rank, first > last for a rank
 * without rows; keys_out / owner_out (capacity cap; cap = 0: only count) receive the keys reachable from at least two ranks,
 * ascending, and the lowest rank that reaches each.  What epialleler_amd/distributed.py's shared_tile_keys computes. */
int epi_shared_tile_keys(const int64_t *ranges, int32_t world, int64_t *keys_out, int32_t *owner_out, int32_t cap, int32_t *n_out);
#define EPI_COMM_ID_BYTES 128
typedef struct epi_comm epi_comm;
int epi_comm_unique_id(void *id_out /* EPI_COMM_ID_BYTES */);
int epi_comm_create(epi_engine *eng, const void *id /* EPI_COMM_ID_BYTES; may be NULL when world == 1 */, int rank, int world,
                    epi_comm **out);
void epi_comm_free(epi_comm *c);
int epi_comm_rank(const epi_comm *c);
int epi_comm_world(const epi_comm *c);
int64_t epi_comm_last_exchange_bytes(const epi_comm *c);   /* bytes this rank handed to the last report's all-reduce(s) */
/* Test hook (world size 1): treat `ntiles` consecutive tiles in the middle of the batch as shared, so that slab, collective
 * and the owners' emit run on a one-GPU box (bench.py sharded_1rank, tests). */
void epi_comm_set_test_shared(epi_comm *c, int ntiles);
int epi_batch_cytosine_report_sharded(epi_batch *b, epi_comm *c, const char *ctx_meth, const char *ctx_unmeth,
                                      const char *ooctx_meth, const char *ooctx_unmeth, uint32_t min_n_ctx,
                                      double min_ctx_meth_frac, double max_ooctx_meth_frac, const int32_t *d_pass,
                                      const char *ctx, int32_t *d_pass_out /* may be NULL */, void *stream, int64_t *nrow_out);
int epi_batch_mhl_report_sharded(epi_batch *b, epi_comm *c, const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac,
                                 void *stream, int64_t *nrow_out);

/* ---- synthetic input (bench/tests; DESIGN.md "Synthetic workload") ------- */
typedef struct {
  uint64_t seed;
  int64_t n_total;        /* rows of the whole (all-ranks) stream              */
  int64_t row_first;      /* first global row generated by this call           */
  int64_t n;              /* rows generated by this call                       */
  int32_t read_len;       /* bytes per template                                */
  int32_t n_chr;
  int32_t depth;          /* genome length per chromosome = rows*read_len/depth */
  int32_t gap_from, gap_len; /* bytes [gap_from, gap_from+gap_len) are 0xFB filler (0 = none) */
} epi_synth_params;
int epi_synth_generate_dev(const epi_synth_params *p, uint8_t *d_xm, int64_t *d_off,
                           int32_t *d_rname, int32_t *d_strand, int32_t *d_start, void *stream);

/* Bytes (and strands) for rows the caller laid out itself -- off/rname/start in device memory, e.g. uniform-random
 * starts sorted on the device and ragged lengths: the same context track and methylation model, per-row
 * hashes keyed by the global row id row_first + k; every gap_every-th template (by hash, 0 = none) carries gap_len
 * filler bytes (0xFB) in its middle.  nbytes = off[n]. */
int epi_synth_fill_dev(uint64_t seed, int64_t row_first, int64_t n, const int64_t *d_off, const int32_t *d_rname,
                       const int32_t *d_start, int64_t nbytes, int32_t gap_every, int32_t gap_len,
                       uint8_t *d_xm, int32_t *d_strand, void *stream);

/* ---- profiling hooks (HIP events around the dominant kernels) ------------ */
void epi_prof_enable(int on);
/* name: "cx_tiles", "threshold", "mhl_tiles", ...; returns accumulated ms and launch count since reset */
int epi_prof_get(const char *name, double *ms_total, int64_t *launches);
void epi_prof_reset(void);

/* Test hook.  The library's environment switches (EPIHIP_*: result-neutral hooks that steer a call onto a rarely
 * taken path) are read once per process; this re-reads them. */
void epi_options_reload(void);

#ifdef __cplusplus
}
#endif
#endif /* EPIHIP_H */
