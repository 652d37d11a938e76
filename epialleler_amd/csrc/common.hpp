// Internal declarations shared by the engine's translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <map>
#include <memory>
#include <vector>
#include "../../include/epihip.h"

namespace epi {

// ---- errors ---------------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define EPI_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return ::epi::fail(_e == hipErrorOutOfMemory ? EPI_ERR_NOMEM : EPI_ERR_HIP,  \
                         "%s failed: %s (%s:%d)", #expr, hipGetErrorName(_e), __FILE__, __LINE__); \
  } while (0)

#define EPI_TRY(expr) do { int _rc = (expr); if (_rc != EPI_OK) return _rc; } while (0)
constexpr int EPI_RETRY_POOL = -77;   // internal (never leaves the library): a deferred sharded report found its row pool too small

// ---- geometry ---------------------------------------------------------------
#ifndef EPI_CX_TILE
#define EPI_CX_TILE 1024
#endif
constexpr int kTile = EPI_CX_TILE;     // positions per CX tile (absolute grid)
constexpr int kMhlTile = 512;          // positions per lMHL tile (56 B of LDS counters per position and strand)
constexpr int kCxPlanes = 16;          // [strand 2][counter 8] u32 planes of kTile entries
constexpr int64_t kPosBias = 1LL << 31;// makes (start + bias) non-negative for any int32 start; multiple of every tile size

// counter slots inside one strand's 8 planes
enum { SLOT_DOT = 0, SLOT_OTHER = 1, SLOT_H = 2, SLOT_h = 3, SLOT_X = 4, SLOT_x = 5, SLOT_Z = 6, SLOT_z = 7, SLOT_SKIP = 8 };

struct Tile {            // one work item of the tile kernels
  int64_t pos0;          // first position of the tile
  int32_t rname;
  int32_t row_lo, row_hi;// candidate rows: same rname, start in (pos0 - Lmax, pos0 + kTile)
  int32_t slot;          // >=0: accumulate into shared slab slot instead of emitting
};

// ---- device buffers that grow on demand --------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes);   // keeps contents only if no growth is needed
  void release();
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct ProfEntry { double ms = 0; int64_t n = 0; };

// ---- per-read class counting shared by the per-read kernels and the fused CX tile kernel (per_read.hip) ----------
struct ClassLut { uint32_t lo0, lo1, hi0, hi1; };   // one byte per code: codes 0-3, 4-7, 8-11, 12-15

// The same table in the form two v_perm_b32 look up (lut16x in tile_common.hpp).  With a selector byte s, v_perm_b32 returns
// byte s of its eight source bytes for s < 8, the sign of source byte 1 / 3 / 5 / 7 spread over the byte for s = 8 .. 11,
// 0x00 for s = 12 and 0xFF for s >= 13.  So perm(low half, code) is the entry for codes 0 .. 7 and a constant per code
// for 8 .. 15, perm(high half, code ^ 8) the other way round; the halves are stored XOR-ed with the constant the other
// lookup returns, and the XOR of the two lookups is the entry for every code.
inline ClassLut lut16_xor_form(const ClassLut &d) {
  uint8_t D[16], L[8], H[8];
  const uint32_t w[4] = {d.lo0, d.lo1, d.hi0, d.hi1};
  for (int c = 0; c < 16; c++) D[c] = (uint8_t)(w[c >> 2] >> (8 * (c & 3)));
  auto S = [](uint8_t x) { return (uint8_t)((x & 0x80u) ? 0xFFu : 0x00u); };
  L[4] = D[4]; H[4] = D[12];
  for (int j = 5; j < 8; j++) { L[j] = (uint8_t)~D[j]; H[j] = (uint8_t)~D[8 + j]; }
  L[2] = D[2] ^ S(H[5]); L[3] = D[3] ^ S(H[7]); H[2] = D[10] ^ S(L[5]); H[3] = D[11] ^ S(L[7]);
  L[1] = D[1] ^ S(H[3]); H[1] = D[9] ^ S(L[3]);
  L[0] = D[0] ^ S(H[1]); H[0] = D[8] ^ S(L[1]);
  auto pack = [](const uint8_t *b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); };
  ClassLut x;
  x.lo0 = pack(L); x.lo1 = pack(L + 4); x.hi0 = pack(H); x.hi1 = pack(H + 4);
  return x;
}
struct ThrParams {                                   // rcpp_threshold_reads.cpp:15-23
  uint32_t min_n_ctx;
  double min_ctx_meth_frac, max_ooctx_meth_frac;
};
// 2-bit membership fields of up to four class strings in one LUT; false when a class string repeats a letter
bool make_field_lut(const char *const cls[4], ClassLut *out);

}  // namespace epi

struct epi_engine {
  int device = 0;
  hipStream_t stream = nullptr;       // default compute stream
  hipStream_t copy_stream = nullptr;  // H2D staging stream
  void *pinned[2] = {nullptr, nullptr};
  size_t pinned_bytes = 0;
  hipEvent_t pinned_done[2] = {nullptr, nullptr};
  // epi_batch_upload lays the rows out while they arrive (engine.hip, layout.hip): two device-side pieces of the source
  // arena, filled by the copy stream and emptied into the batch's own arena by kernels on aux_stream
  hipStream_t aux_stream = nullptr;
  void *dev_stage[2] = {nullptr, nullptr};
  size_t dev_stage_bytes = 0;
  hipEvent_t up_done[2] = {nullptr, nullptr}, stage_free[2] = {nullptr, nullptr};
  int32_t *h_scalars = nullptr;       // pinned scratch for small D2H reads (64 x int64)
};

namespace epi {
struct RowStats {          // filled by k_row_stats
  int32_t max_len;
  int32_t unsorted;        // !=0: some row violates (rname,start) order
  int32_t bad_strand;      // !=0: strand not in {1,2}
  int32_t bad_len;         // !=0: off not non-decreasing
  int32_t deep;            // !=0: some position may be covered by more than 255 rows (row x + 255 starts inside row x)
  // rows by the number of position-aligned 16-byte chunks a row can span, (len + 30) / 16: bin k counts the rows with at most
  // kLenBins[k] chunks that did not fit bin k - 1 (the last bin: everything longer).  The tile kernels pick their lane shape
  // from this, not from the longest row (one 1 kb template among 100 000 PE150 ones must not widen the shape for all).
  uint32_t len_hist[14];
};
constexpr int kLenBinCount = 14;
constexpr int kLenBins[kLenBinCount] = {12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 0x7FFFFFFF};
}  // namespace epi

struct epi_shard_plan;      // comm.hip: shared tile keys of a (batch, tile grid, communicator) triple

struct epi_batch {
  epi_engine *eng = nullptr;
  int64_t n = 0, nbytes = 0;
  // Rows as every kernel sees them: row x owns xm[off[x] .. off[x] + len[x]); off has n + 1 non-decreasing entries
  // (off[n] = end of the data = nbytes), rows do not overlap, gaps between them are allowed.  The constructors take rows
  // back to back (include/epihip.h) and fill `len` from the offsets; epi_batch_realign (layout.hip) replaces xm / off by
  // the batch's own POSITION-CONGRUENT copy: off[x] = start[x] (mod 16), so that the position-aligned 16-byte chunks the
  // tile kernels request are address-aligned (byte-misaligned dwordx4 loads run at ~0.85 of the rate).
  const uint8_t *xm = nullptr;
  const int64_t *off = nullptr;
  const int32_t *len = nullptr;
  const int32_t *rname = nullptr, *strand = nullptr, *start = nullptr;
  bool owns = false;
  int congruent = 0;         // 0: rows as the caller laid them out; 4 / 16: off[x] = start[x] modulo this
  bool cols_owned = false;   // every column is the batch's own copy (uploaded, or adopted and realigned): nobody can change a row
  epi::DevBuf own_xm, own_off, own_len, own_rname, own_strand, own_start;

  // reusable workspace
  epi::DevBuf stats;        // RowStats of the batch (k_row_stats, queued once at creation)
  bool stats_queued = false, stats_host = false;
  hipEvent_t stats_done = nullptr;          // recorded behind k_row_stats on the stream it was queued on
  epi::RowStats h_stats = {};   // host copy, fetched by the first report call (which raises the errors)
  epi::DevBuf scan_tmp;
  epi::DevBuf tiles, tile_nrow, tile_base, tile_out;
  epi::DevBuf pool_key, pool_a, pool_b, pool_c, pool_d, pool_e, pool_f;
  epi::DevBuf misc;         // cursor etc.
  epi::DevBuf mhl_m, mhl_h, mhl_blk, mhl_cont, mhl_cur;   // lMHL pass 1: stretch records, per-read info, record table, block carries
  size_t mhl_rec_cap = 0;   // records that fit mhl_m
  epi::DevBuf heavy_list, heavy_slab, heavy_sums;   // ultra-deep tiles: ids and dense counters (+ lMHL sums)
  epi::DevBuf deep_list;                            // CX report: tiles the lean kernel hands to the general one
  epi::DevBuf diag;                     // check / timing builds only
  size_t pool_cap = 0;      // rows that fit pool_key/pool_a/pool_b
  uint32_t cx_slot_cg = 0, cx_slot_wide = 0;   // pool rows per tile slot: CpG-only reports / reports with CHG, CHH (adapted per call)
  uint32_t mhl_slot = 0, mhl_last_slot = 0, mhl_last_ovf = 0;   // the same for the lMHL report
  uint32_t mhlf_slot = 0;                      // ... and its fused kernel (1024-position tiles)
  bool mhlf_prefer_wide = false;               // most tiles of the last fused lMHL report needed the u64 sums: start with that variant
  uint32_t mhlf_prefer_wide_H = 0;             // ... learned for this haplotype clamp H (the sums shrink with H: a smaller one probes the fast variant again)
  bool mhlf_prefer_fold = false;               // ... many held more than 255 rows: use the kernel with the folded call counters
  epi::DevBuf mhlf_fold_slab;                  // call counters of tiles over 255 rows (kernels built without the LDS fold array)
  bool mhl_shared_fused = false;               // the lMHL slabs attached are in the fused kernel's layout (mhl_common.hpp)
  uint32_t cx_last_slot = 0, cx_last_ovf = 0;  // layout of the last CX report (the sharded second half emits into it)
  // One host synchronisation for a sharded CX report (comm.hip): with cx_defer set the first half queues its kernels and
  // returns without reading anything back; the second half reads and checks everything at once.  Allowed only when an
  // earlier report on this (immutable) batch and tile size found no ultra-deep tile that the host would have to finish
  // before the slab may travel (cx_noheavy_T / _rows: the tile size and threshold that was observed for).
  bool cx_defer = false, cx_deferred = false;
  bool cx_def_hinted = false;
  uint32_t cx_def_heavy_done = 0;
  size_t cx_def_headroom = 0;
  int32_t cx_noheavy_T = 0, cx_noheavy_rows = 0;
  int cx_last_np = 0;                          // ... its number of reported contexts and their codes
  uint32_t cx_last_ctx_of_plane = 0;
  epi::DevBuf pass_tmp;                        // pass flags when thresholding could not be fused and the caller wants none
  epi::DevBuf host_io;                         // device side of the host-pointer calls (pass flags / per-read beta)
  epi::DevBuf mhl_keep_tab;                    // fused lMHL: passing out-of-context counts per total (k_mhl_keep_table)
  int32_t mhl_keep_len = -1;
  double mhl_keep_oo = 0.0;
  epi::DevBuf thr_tab;                         // fused thresholding: decision table for thr_tab_prm over totals 0..thr_tab_len
  int32_t thr_tab_len = -1;
  epi::ThrParams thr_tab_prm = {0, 0.0, 0.0};
  size_t pool_cap2 = 0;     // rows that fit pool_d/pool_e (lMHL doubles)

  // state of the last report (for fetch)
  int last_kind = 0;        // 0 none, 1 cx, 2 mhl; 3 / 4 / 5: first half of a sharded cx / two-kernel lMHL / fused lMHL report
  int64_t last_nrow = 0;
  int32_t last_ntiles = 0;
  int32_t tile_hint_T[4] = {0, 0, 0, 0};    // tile counts of this (immutable) batch by tile size, as found by earlier calls
  int32_t tile_hint_nt[4] = {0, 0, 0, 0};
  int32_t tile_hint_lmax[4] = {0, 0, 0, 0};
  epi::DevBuf tile_bsum[4];                  // ... and the scanned per-block tile counts of the index build
  int32_t last_tile = 0;    // tile size of the last CX report
  // The tile table itself is a function of (rname, start, longest row, tile size, shared keys): while the batch owns its
  // columns the table of the last build is still valid for the same tile size and keys, and build_tiles skips the pass.
  int32_t tiles_T = 0, tiles_nt = 0, tiles_lmax = 0;
  std::vector<int64_t> tiles_shared;
  epi::DevBuf tiles_nt_dev;                  // the tile count as the index pass left it in misc[0]

  // multi-GPU shared tiles
  std::vector<int64_t> shared_keys;
  std::vector<int32_t> shared_owned;
  std::vector<int64_t> dev_shared_keys;      // what d_shared_keys / d_shared_owned currently hold
  std::vector<int32_t> dev_shared_owned;
  epi::DevBuf d_shared_keys, d_shared_owned, d_slot_tile;
  int32_t *d_slab = nullptr;
  epi::DevBuf own_slab, own_slab2;       // the slabs of the library's own sharded entry points (comm.hip)
  std::vector<std::shared_ptr<epi_shard_plan>> shard_plans;
  int32_t *d_mhl_cnt_slab = nullptr;     // lMHL shared tiles: counters and 64-bit sums
  int64_t *d_mhl_sum_slab = nullptr;
  uint32_t mhl_ctx_mask = 0;
};

namespace epi {

hipStream_t pick_stream(epi_batch *b, void *stream);
// device -> host memory of the caller, complete on return.  Page-locked destinations are written by the DMA engine
// directly; pageable ones (malloc, R vectors) go through the engine's two pinned staging buffers in chunks, the copy
// out of buffer k overlapping the DMA into buffer k + 1.  `parts` may list several (dst, src, bytes) pieces: one pipeline.
struct CopyPart { void *dst; const void *src; size_t bytes; };
int copy_parts_to_host(epi_engine *eng, const CopyPart *parts, int nparts, hipStream_t s);
int copy_to_host(epi_engine *eng, void *h_dst, const void *d_src, size_t bytes, hipStream_t s);
int read_scalars(epi_batch *b, hipStream_t s, const void *d_src, size_t bytes, void *h_dst);  // sync D2H of a few bytes

// Host blocks of library-owned report tables (capi.hip).  A large block given back by epi_*_table_free is kept (two blocks,
// at most 1 GiB) and handed to the next table of a similar size: its pages are already mapped, where a fresh 168 MB block
// costs ~9 ms of page faults on the copy threads -- twice the copy itself.
void *table_block(size_t bytes);
void table_block_release(void *p);

// util kernels (util.hip)
int scan_exclusive_u32(const uint32_t *d_in, uint32_t *d_out, int64_t n, uint32_t *d_total,
                       DevBuf &tmp, hipStream_t s);
int scan_block_sums_inplace(uint32_t *d_bsum, int64_t nb, uint32_t *d_total, hipStream_t s);

// tile index (tiles.hip)
// queue k_row_stats for a new batch (no sync, no error: per-read functions accept unsorted rows)
int launch_row_stats(epi_batch *b, hipStream_t s);
// host copy of the statistics (waits for the kernel whatever stream it was queued on); no validation
int fetch_row_stats(epi_batch *b, hipStream_t s);
// layout.hip: the batch's own position-congruent copy of xm (see epi_batch); `modulus` 4 or 16
int realign_batch(epi_batch *b, int modulus, hipStream_t s);
int layout_offsets(epi_batch *b, int modulus, hipStream_t s, DevBuf *new_off, unsigned long long *h_end, uint32_t *head);
int layout_group(int64_t nbytes, int64_t n);
// rows [row_a, row_a + nrows): their bytes inside source bytes [c0, c1) (src[0] = source byte c0) to their place in dst
int layout_copy_range(const uint8_t *src, int64_t c0, int64_t c1, const int64_t *src_off, const int32_t *len, const int64_t *dst_off,
                      int64_t row_a, int64_t nrows, int g, uint8_t *dst, hipStream_t s);
// row statistics (validated: errors for bad offsets/strands/unsorted rows) + tile table; one host sync
// `hinted` (may be null): the caller accepts a tile count remembered from an earlier call on this batch and tile size
// (no host round trip in the middle of the index build) and verifies it against misc[0] at its own synchronisation.
int build_tiles(epi_batch *b, hipStream_t s, int32_t tile_positions, RowStats *h_stats, int32_t *ntiles_out, bool *hinted = nullptr);

// Result-neutral switches (test hooks that steer a call onto a rarely taken path, A/B shapes): read from the environment
// ONCE per process into this struct; epi_options_reload() re-reads it (tests that change a hook inside one process).
struct Options {
  int device = 0;            // EPIHIP_DEVICE        device of the default engine (host-pointer entry points)
  int cx_slot = -1;          // EPIHIP_CX_SLOT       pool rows per tile slot of the CX report (-1: adaptive)
  int cx_lean = 1;           // EPIHIP_CX_LEAN=0     the general (u16-folding) CX kernel for every tile
  int cx_walk = 0;           // EPIHIP_CX_WALK=K     timing builds with -DEPI_CX_WALK_BUILD only: K consecutive tiles per workgroup of the lean CX kernel
  int heavy_rows = 0;        // EPIHIP_HEAVY_ROWS    candidate rows above which a tile is split / set aside (0: default)
  int tile_hint = 1;         // EPIHIP_TILE_HINT=0   tile index counted and scanned by every call
  int realign = 16;          // EPIHIP_REALIGN=0/4/8/16 epi_batch_upload / epi_batch_realign: keep the rows back to back / start them at
                             //                      offsets congruent to their start position modulo 4 / 16 (layout.hip)
  int mhl_fused = 1;         // EPIHIP_MHL_FUSED=0   two-kernel lMHL path for every batch
  int mhl_slot = -1;         // EPIHIP_MHL_SLOT
  int mhl_wg = 0;            // EPIHIP_MHL_WG        256 / 512 (two-kernel path)
  int mhl_tile_group = 0;    // EPIHIP_MHL_TILE_GROUP
  int mhl_multi = 0;         // EPIHIP_MHL_MULTI     wavefront-per-read pass 1
  int mhl_group_g = 0, mhl_group_c = 0;   // EPIHIP_MHL_GROUP="G,C"
  int mhl_sums = 0;          // EPIHIP_MHL_SUMS      32 / 64
  int mhlf_shape = 0;        // EPIHIP_MHLF_SHAPE="G,CA[,CB]"  lane shape of the one-pass lMHL kernel, as G * 100 + CA * 10 + CB
  int mhlf_fold = -1;        // EPIHIP_MHLF_FOLD=0/1 one-pass lMHL kernel without / with the LDS array of folded call counters (-1: by the batch)
  int mhlf_fold_slots = -1;  // EPIHIP_MHLF_FOLD_SLOTS  slab slots of the kernel without it (-1: default)
  int pr_group = 0;          // EPIHIP_GROUP         lanes per read of the general per-read kernel
  int pr_rpg = 0;            // EPIHIP_PR_RPG
  int pr_wide = 1;           // EPIHIP_PR_WIDE=0
  int bam_timing = 0;        // EPIHIP_BAM_TIMING    phase times of the BAM reader on stderr
  int no_hugepage = 0;       // EPIHIP_NO_HUGEPAGE   plain malloc for the BAM reader's large buffers (A/B runs)
  int no_libdeflate = 0;     // EPIHIP_NO_LIBDEFLATE zlib's inflate for the BGZF blocks although libdeflate.so.0 can be loaded
};
const Options &options();

// profiling
void prof_begin(const char *name, hipStream_t s);
void prof_end(const char *name, hipStream_t s);

// Check build (`make check`: -DEPI_CHECK -DEPI_MHL_CHECK -> libepihip_check.so): the tile kernels verify the global
// addresses and pool indices they are about to use and record the first violation {code, v0, v1, block, thread}
// instead of performing the access; the host turns it into EPI_ERR_STATE.  Compiled out of the product library.
#ifdef __HIPCC__
#ifdef EPI_CHECK
__device__ __forceinline__ bool epi_dev_check(uint32_t *dbg, bool ok, uint32_t code, int64_t v0, int64_t v1) {
  if (!ok && dbg && atomicCAS(dbg, 0u, code) == 0u) {
    dbg[1] = (uint32_t)v0; dbg[2] = (uint32_t)v1; dbg[3] = blockIdx.x; dbg[4] = threadIdx.x; dbg[5] = (uint32_t)(v0 >> 32);
  }
  return ok;
}
#define EPI_DEV_CHECK(dbg, ok, code, v0, v1) epi_dev_check(dbg, ok, code, v0, v1)
#else
#define EPI_DEV_CHECK(dbg, ok, code, v0, v1) true
#endif
#endif

// context-string helpers (host)
// Wave-level scans and reductions as DPP moves (row_shr 1, 2, 4, 8 inside a row of 16 lanes, then row_bcast:15 into
// rows 1 and 3 and row_bcast:31 into rows 2 and 3): VALU instructions, where __shfl_* is a ds_bpermute through the
// LDS pipe with ~100 cycles of latency.  Device code only.
#ifdef __HIPCC__
// (bound_ctrl = true on the row shifts: a lane without a source reads 0, the same value as `old` = 0 would give it, and the
//  compiler can then fold move and add into one v_add_u32_dpp instead of v_mov 0 + v_mov_dpp + v_add)
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {      // inclusive prefix sum over the 64 lanes
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {       // sum over the 64 lanes (uniform result)
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_u32(v), 63);
}
#endif

inline unsigned ctx_to_idx(unsigned char c) { return ((unsigned(c) + 2u) >> 2) & 15u; }   // src/epialleleR.h:28

// A grid holds fewer than 2^32 threads (a larger one wraps silently): refuse instead.
inline int check_grid(int64_t blocks, int threads, const char *what) {
  if (blocks < 0 || blocks > 0x7FFFFFFFLL || blocks * threads >= (1LL << 32))
    return fail(EPI_ERR_ARG, "%s: batch too large for one launch (%lld workgroups of %d threads)", what, (long long)blocks, threads);
  return EPI_OK;
}


}  // namespace epi
