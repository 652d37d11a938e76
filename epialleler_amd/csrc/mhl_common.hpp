// Shared by the two lMHL translation units: mhl_report.hip (the two-kernel path: k_mhl_rows + k_mhl_tiles, heavy tiles,
// gather, the extern "C" entry points) and mhl_fused.hip (the one-pass tile kernel).  Device helpers only; gfx950.
#pragma once
#include "common.hpp"
#include "tile_common.hpp"

namespace epi {

constexpr int MHL_WG = 512;                       // threads per workgroup of the heavy-tile and slab kernels
constexpr int MHL_WG_SHORT = 256;                 // k_mhl_tiles for reads of one k_mhl_rows block: five workgroups per CU,
                                                  // 10.8 against 11.6 ms on config 4 (long reads, whose record walk is
                                                  // latency-bound, keep 512: 9.4 against 12.4 ms on 10 kb reads)
constexpr int MHL_T = kMhlTile;
// One difference array: entry of tile position p (0..T, T = "after the tile") sits at p + p/8 -- the padding makes the
// prefix-sum phase, where a lane walks 8 consecutive entries, free of LDS bank conflicts (stride 9 x 8 bytes per lane).
constexpr int MHL_SLEN = (kMhlTile + 1) + ((kMhlTile + 1) >> 3) + 1;
__host__ __device__ constexpr int mhl_pad(int p) { return p + (p >> 3); }
constexpr int MHL_NSUM = 6 * MHL_SLEN;            // u64 per tile: difference arrays of sum S(M), sum h, sum S(h), two strands each
constexpr int MHL_BLK_SHIFT = 11;                 // a block of the multi-block row kernel: 64 lanes x 32 bytes
constexpr int MHL_REGIONS = 64, MHL_CUR_STRIDE = 32;   // record allocation cursors (u64 each, 256 B apart)
#ifndef EPI_MHL_NU
#define EPI_MHL_NU CX_NU
#endif
constexpr int MHL_NU = EPI_MHL_NU;                // dword loads a lane keeps in flight per row in the tile kernel
#ifndef EPI_MHL_WPS
#define EPI_MHL_WPS 6
#endif

__host__ __device__ __forceinline__ uint64_t nrS(uint64_t n) { return n < 2 ? n : (n * (n + 1) * (n + 2)) / 6; }   // :39-43
// mhl_lookup[n] (:110-116) without the table; indices clamp at 65535 (the reference's table ends there)
__device__ __forceinline__ uint64_t mhl_lut(uint32_t n, uint32_t H) {
  if (n > 65535u) n = 65535u;
  const uint32_t k = n < H ? n : H;
  // below 1024 the product fits 32 bits (three 64-bit multiplies and a 64-bit division by 6 otherwise: ~40 VALU);
  // decided per wavefront, so short-read batches never execute the wide form
  if (__builtin_expect(__ballot(k >= 1024u) != 0ull, 0)) return nrS(k);
  return k < 2u ? (uint64_t)k : (uint64_t)((k * (k + 1u) * (k + 2u)) / 6u);
}

struct Seg { uint32_t has; uint32_t cnt; };       // scan element: saw a cut? members since the last cut
__device__ __forceinline__ Seg seg_combine(Seg left, Seg right) {   // state after `left` then `right`
  Seg r;
  r.has = left.has | right.has;
  r.cnt = right.has ? right.cnt : left.cnt + right.cnt;
  return r;
}

// nibble -> flags: 1 member (in context, methylated), 2 cut (in context, unmethylated), 4 skipped ('+'/'-'/filler, :187),
// 8 / 16 out-of-context methylated / unmethylated (:176-177).  Built on the host from the context string.
struct MhlLut { uint32_t lo0, lo1, hi0, hi1; };

// Per-byte bit masks of the 16*C bytes a lane owns: u32 for C = 2, u64 for C = 3, 4.
template <int C> struct MaskOf { using T = uint64_t; };
template <> struct MaskOf<2> { using T = uint32_t; };
template <class M> struct Chunk { M U, L, K, V; uint32_t oom, oou; };

__device__ __forceinline__ int bm_popc(uint32_t x) { return __popc(x); }
__device__ __forceinline__ int bm_popc(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int bm_ctz(uint32_t x) { return __ffs(x) - 1; }                    // x != 0
__device__ __forceinline__ int bm_ctz(uint64_t x) { return __ffsll((unsigned long long)x) - 1; }
__device__ __forceinline__ int bm_msb(uint32_t x) { return 31 - __clz(x); }                   // x != 0
__device__ __forceinline__ int bm_msb(uint64_t x) { return 63 - __clzll((long long)x); }
template <class M> __device__ __forceinline__ M bm_below(int n) {                              // bits [0, n), 0 <= n <= width
  return n >= (int)(8 * sizeof(M)) ? ~(M)0 : (((M)1 << n) - (M)1);
}

// bit `bit` of the four bytes of f as a nibble
__device__ __forceinline__ uint32_t plane_nibble(uint32_t f, int bit) {
  uint32_t t = (f >> bit) & 0x01010101u;
  t |= t >> 7;
  t |= t >> 14;
  return t & 0xFu;
}

// The 16*C bytes at g0 (16-byte aligned) of the row [rs,re), loaded (so that a caller can have the next chunk's loads
// in flight while it works on this one) ...
template <int C> struct ChunkRaw { uint32_t ww[4 * C]; int lo, hi; };       // hi <= lo: nothing of the row in this chunk

template <class M> __device__ __forceinline__ uint32_t lead_members(const Chunk<M> &c) {    // members before the first cut (all if none)
  return (uint32_t)bm_popc(c.U & (c.L ? ((c.L & ((M)0 - c.L)) - (M)1) : ~(M)0));
}
template <class M> __device__ __forceinline__ uint32_t trail_members(const Chunk<M> &c) {   // members after the last cut (all if none)
  return (uint32_t)bm_popc(c.U & (c.L ? ~bm_below<M>(bm_msb(c.L) + 1) : ~(M)0));
}

// Lane exchanges inside a group of G lanes as DPP moves where the distance allows (inside a row of 16 lanes): a
// __shfl_* is a ds_bpermute through the LDS pipe with ~100 cycles of latency, and pass 1 chains about 35 of them per
// wavefront.  A lane whose partner lies outside its group gets another group's value (or 0): callers mask those lanes.
template <int CTRL>
__device__ __forceinline__ uint32_t lane_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }   // (no source: 0)
// (row shifts do not cross the rows of 16 lanes: groups of 32 or 64 lanes keep the shuffles)
template <int G, int D>
__device__ __forceinline__ uint32_t grp_up(uint32_t v) {             // value of lane - D
  if constexpr (G > 16) return __shfl_up(v, D, 64);
  else if constexpr (D == 1) return lane_dpp<0x111>(v);
  else if constexpr (D == 2) return lane_dpp<0x112>(v);
  else if constexpr (D == 4) return lane_dpp<0x114>(v);
  else return lane_dpp<0x118>(v);
}
template <int G, int D>
__device__ __forceinline__ uint32_t grp_down(uint32_t v) {           // value of lane + D
  if constexpr (G > 16) return __shfl_down(v, D, 64);
  else if constexpr (D == 1) return lane_dpp<0x101>(v);
  else if constexpr (D == 2) return lane_dpp<0x102>(v);
  else if constexpr (D == 4) return lane_dpp<0x104>(v);
  else return lane_dpp<0x108>(v);
}
// partner for a butterfly reduction over a group, applied for D = G/2 ... 1: quad_perm for 1 and 2, row_half_mirror
// (i <-> 7-i) for 4 and row_mirror (i <-> 15-i) for 8 pair up the same partial sums as lane ^ D would
template <int D>
__device__ __forceinline__ uint32_t grp_bfly(uint32_t v) {
  if constexpr (D == 1) return lane_dpp<0xB1>(v);
  else if constexpr (D == 2) return lane_dpp<0x4E>(v);
  else if constexpr (D == 4) return lane_dpp<0x141>(v);
  else if constexpr (D == 8) return lane_dpp<0x140>(v);
  else return __shfl_xor(v, D, 64);
}

// segmented scans over the G lanes of a read: members of the open segment to the left (pf) / right (sf) of a lane
template <int G, int D>
__device__ __forceinline__ void seg_scan_steps(Seg &pf, Seg &sf, int sub) {
  if constexpr (D < G) {
    Seg l, r;
    l.has = grp_up<G, D>(pf.has); l.cnt = grp_up<G, D>(pf.cnt);
    if (sub >= D) pf = seg_combine(l, pf);
    r.has = grp_down<G, D>(sf.has); r.cnt = grp_down<G, D>(sf.cnt);
    if (sub + D < G) sf = seg_combine(r, sf);            // walking leftwards: `r` was seen first
    seg_scan_steps<G, D * 2>(pf, sf, sub);
  }
}
template <int D>
__device__ __forceinline__ uint32_t grp_sum(uint32_t v) {            // over the 2*D lanes of a group, every lane gets it
  if constexpr (D >= 1) { v += grp_bfly<D>(v); return grp_sum<D / 2>(v); }
  else return v;
}
template <int D>
__device__ __forceinline__ uint32_t grp_or(uint32_t v) {
  if constexpr (D >= 1) { v |= grp_bfly<D>(v); return grp_or<D / 2>(v); }
  else return v;
}

template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t mhl_dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, ROWS == 0xF);   // lanes without a source get 0
}
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned long long mhl_dpp(unsigned long long v) {
  return ((unsigned long long)mhl_dpp<CTRL, ROWS>((uint32_t)(v >> 32)) << 32) | mhl_dpp<CTRL, ROWS>((uint32_t)v);
}
template <class ST>
__device__ __forceinline__ ST mhl_wave_scan(ST v) {
  v += mhl_dpp<0x111, 0xF>(v);
  v += mhl_dpp<0x112, 0xF>(v);
  v += mhl_dpp<0x114, 0xF>(v);
  v += mhl_dpp<0x118, 0xF>(v);
  v += mhl_dpp<0x142, 0xA>(v);
  v += mhl_dpp<0x143, 0xC>(v);
  return v;
}


// ---- host helpers defined in mhl_report.hip ---------------------------------------------------------------------------
size_t mhl_pool_rows(const epi_batch *b);
int ensure_mhl_pool(epi_batch *b, size_t rows);
int pick_mhl_group(int32_t max_len);             // lanes per read x chunks per lane as G * 8 + C (0: longer than 64 lanes cover)
MhlLut make_mhl_lut(uint32_t ctx_mask);

// ---- the fused path (mhl_fused.hip) --------------------------------------------------------------------------------------
#ifndef EPI_MHLF_T                                // (timing builds vary it; the product uses this value)
#define EPI_MHLF_T 1024
#endif
constexpr int MHLF_T = EPI_MHLF_T;                // positions per tile of the fused kernel
constexpr int MHLF_CNT_PLANES = 4, MHLF_SUM_PLANES = 6;   // shared-tile slabs: int32 [4][T] (n+, n-, coverage differences
                                                  // +, -), int64 [6][T] (difference arrays of S(M), h, S(h), two strands each)
// *done = false: the batch is not eligible -- the caller runs the two-kernel path instead.
int mhl_fused_report(epi_batch *b, uint32_t ctx_mask, uint32_t H, int hmin, double max_oo, hipStream_t s,
                     int64_t *nrow_out, bool *done);
bool mhl_fused_eligible(epi_batch *b, uint32_t ctx_mask, const RowStats &st);
int mhl_fused_finish_shared(epi_batch *b, hipStream_t s, int64_t *nrow_out);

}  // namespace epi
