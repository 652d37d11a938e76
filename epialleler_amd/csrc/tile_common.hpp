// Device code shared by the tile kernels (CX report and lMHL): the nibble LUT, the per-lane view of
// a row's in-tile slice and the branch-free "four bases -> four LDS atomics" step.
#pragma once
#include "common.hpp"

namespace epi {

#ifndef EPI_CX_NU
#define EPI_CX_NU 10
#endif
constexpr int CX_NU = EPI_CX_NU;              // dword loads a lane keeps in flight per row

// nibble -> (counter slot, increment) as a 16-entry byte LUT for v_perm_b32: bits 0-2 = slot (see
// enum in common.hpp), bits 4-5 = increment.
//   code:     0    1    2    3    4    5    6    7 |   8    9   10   11 |  12   13   14   15
//   byte:  0x11 0x11 0x12 0x11 0x11 0x11 0x14 0x16 | 0x11 0x21 0x13 0x00 | 0x10 0x11 0x15 0x17
// increment 0 = skipped ('+'/'-' and filler, rcpp_cx_report.cpp:123: the atomic still issues but adds
// nothing); 2 = nibble 9, which IS the reference's coverage slot and so counts twice (:126-127).
constexpr uint32_t kLutLo0 = 0x11121111u, kLutLo1 = 0x16141111u, kLutHi0 = 0x00132111u, kLutHi1 = 0x17151110u;
// Packed variant (two u16 counters per LDS dword: pair 0 = ('.', other), 1 = (H, h), 2 = (X, x), 3 = (Z, z); the
// even slot in the low half): byte = [increment of the high half: bits 4-5][pair: bits 2-3][increment of the low half: bits 0-1]
//   code:     0    1    2    3    4    5    6    7 |   8    9   10   11 |  12   13   14   15
//   byte:  0x10 0x10 0x05 0x50 0x90 0x10 0x09 0x0D | 0xD0 0x20 0x14 0x00 | 0x01 0x10 0x18 0x1C
// bits 6-7 (ignored by the counters) flag the stray nibbles the lMHL kernel needs: 1 = nibble 3, 2 = nibble 4, 3 = nibble 8
constexpr uint32_t kPkLo0 = 0x50051010u, kPkLo1 = 0x0D091090u, kPkHi0 = 0x001420D0u, kPkHi1 = 0x1C181001u;
constexpr int kCxGuard = 4;               // dwords of LDS padding around the counters (see cx_add_dword)

struct RowCols {                          // the batch columns a tile kernel reads
  const uint8_t *xm;
  const int64_t *off;                     // row r owns xm[off[r] .. off[r] + len[r])
  const int32_t *len;
  const int32_t *start, *strand, *pass;   // pass may be null (all TRUE)
};

struct RowSlice {                         // the part of one row that falls inside the tile, seen from one lane
  const uint32_t *src;                    // this lane's first dword of the (dword-aligned) slice in xm
  uint32_t *dst[4];                       // LDS cells (counter plane 0 of the row's strand) of the lane's first dword,
                                          // in the order the lane walks its four bytes (see rot8)
  int rot8;                               // 8 * byte rotation: sub-step j handles byte (j + rot) & 3
  int nd;                                 // dwords in the slice (0 = nothing to do)
  int tl;                                 // nd - 1 - sub: the lane's dword u*G is in the slice iff u*G <= tl, the last one iff ==
  uint32_t lc4;                           // 0x08080808 when the read failed thresholding (lower-case, :118)
  uint32_t pick0;                         // 0x03020100 | lc4 >> 1: the v_perm selector that merges the two LUT halves
  uint32_t mask_first, mask_last;         // valid bytes of the slice's first / last dword
};

struct RowVals {                          // the columns of one candidate row, loaded
  int32_t st, len, sd, ps;
  int64_t o;
  bool ok;                                // r < row_hi
};

__device__ __forceinline__ RowVals cx_load_row(const RowCols &a, const Tile &td, int r) {
  RowVals v;
  v.st = 0; v.len = 0; v.sd = 1; v.ps = 1; v.o = 0;
  v.ok = r < td.row_hi;
  if (v.ok) {
    v.st = a.start[r];
    v.o = a.off[r];
    v.len = a.len[r];                     // >= 0, start + len < 2^31 (k_row_stats)
    v.sd = a.strand[r];
    v.ps = a.pass ? a.pass[r] : 1;
  }
  return v;
}

template <int T, int G, bool PK = false>
__device__ __forceinline__ RowSlice cx_slice_of(const RowCols &a, const RowVals &v, const Tile &td, int sub, uint32_t *cnt) {
  RowSlice m;
  m.src = nullptr; m.dst[0] = m.dst[1] = m.dst[2] = m.dst[3] = cnt; m.rot8 = 0; m.nd = 0; m.tl = -1; m.lc4 = 0; m.pick0 = 0x03020100u;
  m.mask_first = ~0u; m.mask_last = ~0u;
  if (v.ok) {
    // row index of the tile's first position; |rel| < Lmax + T for a candidate row
    const int32_t rel = (int32_t)((uint32_t)td.pos0 - (uint32_t)v.st);
    const int32_t lo = rel > 0 ? rel : 0;
    const int32_t hi = v.len < rel + T ? v.len : rel + T;
    if (hi > lo) {
      const int64_t b0 = v.o + lo;
      const int32_t e_lo = (int32_t)b0 & 3;              // slice bytes are e in [e_lo, e_hi) from the aligned start
      const int32_t e_hi = e_lo + (hi - lo);
      m.nd = (e_hi + 3) >> 2;
      m.tl = m.nd - 1 - sub;
      m.src = reinterpret_cast<const uint32_t *>(a.xm + (b0 - e_lo)) + sub;
      // Bank-conflict-free LDS atomics: at sub-step j a lane adds at position d + 4*sub' + ((j+rot)&3), i.e. in
      // bank residue (d + j + rot) mod 4.  The 8 lanes of one "eighth" of a 32-lane half are 4 cells apart
      // (8 banks of one residue); rot = eighth - d gives the four eighths the residues j, j+1, j+2, j+3 whatever
      // rows (and row alignments d) they work on, so the 32 lanes always hit 32 different banks.
      const int d = lo - rel - e_lo;
      const int rot = ((int)((threadIdx.x & 31) >> 3) - d) & 3;
      uint32_t *dst0 = cnt + (v.sd - 1) * (PK ? 4 : 8) * T + d + 4 * sub;
      m.rot8 = rot * 8;
#pragma unroll
      for (int j = 0; j < 4; j++) m.dst[j] = dst0 + ((j + rot) & 3);
      m.lc4 = v.ps == 0 ? 0x08080808u : 0u;
      m.pick0 = v.ps == 0 ? 0x07060504u : 0x03020100u;
      m.mask_first = sub == 0 ? 0xFFFFFFFFu << (8 * e_lo) : ~0u;
      m.mask_last = 0xFFFFFFFFu >> (8 * (4 * m.nd - e_hi));
    }
  }
  return m;
}

template <int T, int G, bool PK = false>
__device__ __forceinline__ RowSlice cx_row_slice(const RowCols &a, const Tile &td, int r, int sub, uint32_t *cnt) {
  return cx_slice_of<T, G, PK>(a, cx_load_row(a, td, r), td, sub, cnt);
}

// One dword (four bases) of a row into the LDS counters.  Every lane issues all four atomics: bytes
// outside the slice and skipped codes are turned into "+0 on plane 0" by the masks, never branched
// around.  A masked byte can sit up to 3 cells outside [0,T): the counters carry kCxGuard cells of
// padding for that.  The byte order is rotated per lane (RowSlice::rot8) so that the 32 lanes of a half
// wavefront always hit 32 different LDS banks.
template <int T, int OFF, bool FIRST, bool PK = false>
__device__ __forceinline__ uint32_t cx_add_dword(uint32_t w, bool last, const RowSlice &m) {   // returns the stray-nibble flags (packed LUT)
  const uint32_t lo3 = w & 0x07070707u;                  // low three bits of the four codes (unpack_ctx_idx)
  // 16-entry byte LUT = two v_perm lookups (codes 0-7 / 8-15) + a third v_perm that picks, per byte,
  // the second result when bit 3 of the code is set or the read is lower-cased (selector j + 4*bit3):
  // no multiply, no masks
  const uint32_t pick = ((w >> 1) & 0x04040404u) | m.pick0;
  uint32_t s4 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(PK ? kPkHi1 : kLutHi1, PK ? kPkHi0 : kLutHi0, lo3),
                                      __builtin_amdgcn_perm(PK ? kPkLo1 : kLutLo1, PK ? kPkLo0 : kLutLo0, lo3), pick);
  uint32_t vm = last ? m.mask_last : ~0u;
  if (FIRST) vm &= m.mask_first;
  s4 &= vm;
  const uint32_t stray = s4 & 0xC0C0C0C0u;               // byte j = base j of the dword (before the rotation)
  s4 = __builtin_amdgcn_alignbit(s4, s4, m.rot8);        // rotate right by rot bytes: byte j <- byte (j+rot)&3
  if constexpr (PK) {
    // value added to the pair's dword = low increment | high increment << 16, put together by one v_perm per base
    const uint32_t lo4 = s4 & 0x03030303u, hi4 = (s4 >> 4) & 0x03030303u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t plane = __builtin_amdgcn_ubfe(s4, 8 * j + 2, 2);
      asm("" : "+v"(plane));
      const uint32_t val = __builtin_amdgcn_perm(hi4, lo4, 0x0C000C00u | ((4u + j) << 16) | (uint32_t)j);
      atomicAdd(m.dst[j] + OFF + plane * T, val);
    }
    return stray;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {                          // OFF = 4 * (this dword's index - the lane's first index)
    uint32_t plane = __builtin_amdgcn_ubfe(s4, 8 * j, 3);
    asm("" : "+v"(plane));                               // keep v_bfe_u32 + v_lshl_add_u32 (hipcc otherwise re-derives
                                                         // the address with shift + and + add: one more VALU per base)
    const uint32_t inc = __builtin_amdgcn_ubfe(s4, 8 * j + 4, 2);
    atomicAdd(m.dst[j] + OFF + plane * T, inc);
  }
  return 0u;
}

// LDS dwords of one tile's counters: u32 [strand][8][T], or packed u16 pairs [strand][4][T] (tile_common.hpp)
template <int T, bool PK> constexpr int cx_lds_dwords() { return (PK ? 8 : kCxPlanes) * T; }
// waves per SIMD the kernel is compiled for: as many workgroups per CU as LDS (160 KiB) and 2048 threads allow
template <int T, int WG, bool PK> constexpr int cx_waves_per_simd() {
  const int by_lds = (160 * 1024) / (cx_lds_dwords<T, PK>() * 4 + 256 + (PK ? 4 * T : 0)), by_thr = 2048 / WG;   // + emit candidate lists
  const int wgs = by_lds < by_thr ? by_lds : by_thr;
  return wgs * WG / 256;
}

// Adds a tile's LDS counters into its dense u32 slab [16][T] in HBM (shared tiles, heavy tiles).
template <int T, int WG, bool PK>
__device__ __forceinline__ void cx_dump_slab(const uint32_t *cnt, int32_t *slab) {
  uint32_t *dst = reinterpret_cast<uint32_t *>(slab);
  for (int i = threadIdx.x; i < cx_lds_dwords<T, PK>(); i += WG) {
    const uint32_t v = cnt[i];
    if (!v) continue;
    if constexpr (PK) {
      const int pl = i / T, p = i % T;                      // pl = strand*4 + pair -> planes 2*pl (low half), 2*pl+1
      if (v & 0xFFFFu) atomicAdd(dst + (2 * pl) * T + p, v & 0xFFFFu);
      if (v >> 16) atomicAdd(dst + (2 * pl + 1) * T + p, v >> 16);
    } else {
      atomicAdd(dst + i, v);
    }
  }
}

}  // namespace epi
