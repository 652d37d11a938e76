// rcpp_mhl_report (src/rcpp_mhl_report.cpp:46-228) in ONE pass over the packed bytes: the path of generateMhlReport's
// defaults (one haplotype context, ctx = "Zz" / "Xx" / "Hh") on reads of up to 4 KiB.
//
// A workgroup owns one tile of MHLF_T positions (tiles.hip); G lanes own a candidate row, 16*C contiguous bytes per lane,
// loaded position-aligned (global_load_dwordx4 at any byte alignment) so that a dword of xm lines up with four LDS cells.
// Per row the reference needs: h = its in-context bytes, the out-of-context (un)methylated counts
// that decide whether the row is kept (:176-179), per methylated stretch of M members S(M) on every byte between the
// stretch's first and last member (:160-171), and per counted byte of a kept row +1 coverage, +h, +S(h), +S(M)
// (:185-195).  All of that is constant over intervals of a row, so the tile keeps
//   * coverage, sum h, sum S(h), sum S(M) as DIFFERENCE arrays in LDS (an interval = two LDS atomics),
//   * n = calls of the context (either case) as u8 counters, four positions per dword (one ds_add_u64 per eight
//     positions that hold a call), folded into wider counters every 255 rows,
// and a row of the table exists iff n > cov/2 (then no other context can win the majority rule, :76-86).
//
// How the row analysis is built (the kernel is bound by integer VALU issue; every item below replaced
// something that cost more instructions per row):
//   * Lane shape: G lanes own a row, a lane holds 16 (CA + CB) contiguous bytes as one mask block (CB = 0: up to 64 bytes,
//     64-bit masks) or two (CA = 3, CB = 2: 80 bytes as a 48-byte and a 32-byte block, combined like two lanes before the
//     lane scans): PE150 templates are 4 lanes x 80 bytes, 16 rows per wavefront step, 256-thread workgroups.
//   * Row bytes come through a raw buffer descriptor over the tile's stretch of xm (32-bit offsets, unaligned b128 loads;
//     a chunk the row does not reach is sent out of range and reads as zeros); bytes outside the row are zeroed once after
//     the load (code 0 has no flag) instead of masking every plane.
//   * The member / in-context bit planes come from SWAR compares on the nibble-packed codes of two dwords
//     (((pk & 7..7) ^ k..k) + 7..7: bit 3 of a nibble says "not this context"; the case bit of a code is its bit 3) -- plain
//     two-operand VALU -- and ONE v_dot4 per plane and pair of dwords gathers the flags of eight bytes into a mask byte.
//   * The out-of-context counts need no positions: 2-bit fields of a 16-entry byte LUT (two v_perm_b32 and an XOR,
//     lut16_xor_form in common.hpp), summed field-wise and by v_sad_u8 (the per_read.hip scheme).
//   * Skipped ('+', '-', filler) and stray codes (nibbles 3, 4, 8, 9 alias the reference's sum / coverage slots) only raise
//     a flag in the counting LUT; a lane that sees one builds that plane from the bytes it STILL HOLDS in registers
//     (mhlf_eq_plane) -- nothing is reloaded.  The fast variant hands a tile with a stray code to the WIDE one.
//   * Stretches: span bits by carry propagation (mhlf_fill_up), S(members) per run from a table in LDS (k < 256); where the
//     lane holds no skipped byte a segment's span is one run and its member count a popcount.
//   * Emit: the seven difference arrays are prefix-summed in place, one array per wavefront at a time (lane l owns the quads
//     64 j + l: conflict-free ds_read_b128, 28 wavefront scans per tile); every thread rules on its own positions, the cells
//     that become rows are listed in key order in the (by then dead) coverage array and written densely, one row per lane.
//   * LDS 31.8 KB without the fold array (FOLD = false: five workgroups per CU; a tile of more than 255 rows folds its u8
//     call counters into a slot of a slab in HBM), 35.8 KB with it (four per CU); 96 VGPRs, no scratch.
//   * No whole-batch fallback: a tile whose u32 sums could wrap (checked after the rows, from the rows' own S(h)), with
//     more than 32767 candidate rows, or with a stray code is put on a list and redone by the WIDE variant (u64 sums, u32
//     coverage and call counters: no limit) -- only that tile.  Tiles shared with other ranks of a sharded run dump their raw
//     arrays into the slabs the ranks all-reduce (comm.hip / distributed.py) instead of emitting.
#include "common.hpp"
#include "mhl_common.hpp"
#include <string.h>
#include <type_traits>

namespace epi {

#ifndef EPI_MHLF_WG                               // (timing builds: workgroup size, waves per SIMD of the fast variant, phases left out)
#define EPI_MHLF_WG 512
#endif
#ifndef EPI_MHLF_WPS
#define EPI_MHLF_WPS 8
#endif
#ifndef EPI_MHLF_ABLATE
#define EPI_MHLF_ABLATE 0                         // 1: no emit, 2: no row analysis (loads only), 4: no stretch runs, 8: no call counters
#endif
constexpr int MHLF_WG = EPI_MHLF_WG, MHLF_NW = MHLF_WG / 64, MHLF_Q = MHLF_T / 4;
#ifndef EPI_MHLF_WG2
#define EPI_MHLF_WG2 256
#endif
constexpr int MHLF_WG2 = EPI_MHLF_WG2;                     // workgroup of the two-block lane shapes
constexpr int MHLF_FOLD = 255;                    // u8 call counters: a row adds at most 1 per position
constexpr int MHLF_STAB = 256;                    // S(k) table entries in LDS
constexpr uint32_t MHLF_FOLD_SLOTS = 2048;        // slab slots (8 KB each) of the kernels built without the LDS fold array
constexpr int MHLF_FAST_ROWS = 32767;             // packed u16 coverage halves / u16 folded counters of the fast variant

// mhl_keep's out-of-context test without the fp64 division per row: (double)m / (double)n > max_oo is monotone in m, so
// per n there is a count of passing m = 0 .. n; the table is filled on the device with the reference's own expression
// (:178-179; 0/0 = NaN compares false: kept).
__global__ __launch_bounds__(256) void k_mhl_keep_table(double max_oo, int32_t nmax, uint32_t *__restrict__ tab) {
  const int32_t n = (int32_t)(blockIdx.x * 256 + threadIdx.x);
  if (n > nmax) return;
  int32_t lo = 0, hi = n + 1;                              // first m in [0, n] with frac > max_oo (n + 1: none)
  while (lo < hi) {
    const int32_t m = (lo + hi) >> 1;
    const double frac = (double)(uint32_t)m / (double)(uint64_t)(uint32_t)n;
    if (frac > max_oo) hi = m; else lo = m + 1;
  }
  tab[n] = (uint32_t)lo;
}

struct MhlFArgs {
  const uint8_t *xm;
  const int64_t *off;                     // row r owns xm[off[r] .. off[r] + len[r]); n + 1 non-decreasing offsets
  const int32_t *len;
  const int32_t *start, *strand;
  int64_t xm_cap;                         // readable bytes behind xm
  const Tile *tiles;
  uint32_t k7;                            // the context's methylated code (2, 6 or 7) in every byte
  MhlLut lut2;                            // counting LUT: bit 0 out-of-context methylated, bit 2 skipped (code 11), bit 4
                                          // out-of-context unmethylated, bit 6 stray (nibbles 3, 4, 8, 9)
  int32_t hmin;
  const uint32_t *keep_tab;               // [n] = passing out-of-context methylated counts for n out-of-context calls
  uint32_t H, ctx;                        // haplotype window clamp (:112), reported context code
  uint32_t *pool_key, *pool_cov;
  unsigned long long *pool_hs, *pool_nu, *pool_de;
  uint32_t pool_cap, slot_rows, ovf_base;
  uint32_t *cursor, *tile_nrow, *tile_base;
  uint32_t *deep_count, *deep_list;       // tiles the fast variant sets aside for the WIDE one
  const uint32_t *tile_list;              // non-null: the launch covers tile_list[0 .. ntiles) (the deep list)
  int max_rows;                           // fast variant: tiles with more candidate rows go to the deep list
  uint32_t *fold_slab, *fold_cursor;      // kernels built without the LDS fold array: u32 [slot][2][T] call counters of the
  uint32_t fold_slots;                    // tiles with more than 255 rows, slots handed out through the cursor
  int32_t *slab_cnt;                      // shared tiles: [slot][MHLF_CNT_PLANES][T] int32, [slot][MHLF_SUM_PLANES][T] int64
  unsigned long long *slab_sum;
  uint32_t *dbg;                          // check build only (EPI_CHECK): first index violation; null in the product
  int64_t nrows;
};

struct __attribute__((packed, aligned(1))) MhlU4u { uint32_t x, y, z, w; };

// the W = 16*C bytes at byte offset g0 (any alignment, may reach outside [0, cap) for the first / last rows of a batch);
// the row's bytes are [g0 + lo0, g0 + hi0)
template <int C>
__device__ __forceinline__ ChunkRaw<C> mhlf_load(const uint8_t *__restrict__ xm, int64_t cap, int64_t g0, int32_t lo0, int32_t hi0) {
  constexpr int W = 16 * C;
  ChunkRaw<C> r;
  r.lo = 0; r.hi = 0;
#pragma unroll
  for (int j = 0; j < 4 * C; j++) r.ww[j] = 0u;
  int32_t lo = lo0 < 0 ? 0 : lo0, hi = hi0 > W ? W : hi0;
  if (hi <= lo) return r;
  r.lo = lo; r.hi = hi;
  // A chunk may start before or end behind the buffer (first and last rows of a batch only, by less than 16 bytes): it
  // is then put together from the two aligned 16-byte blocks it straddles, a block outside [0, cap) reading as zeros
  // (the buffer is 16-byte aligned and cap a multiple of 16).
  const bool edge = g0 < 0 || g0 + W > cap;
#pragma unroll
  for (int j = 0; j < C; j++) {
    if (j == 0 || 16 * j < hi0) {                          // (chunks behind the row's end stay zero)
      const int64_t g = g0 + 16 * j;
      if (__builtin_expect(edge, 0)) {
        const int64_t A = g & ~(int64_t)15;
        const int sh = (int)(g & 15);
        unsigned long long q[4] = {0ull, 0ull, 0ull, 0ull};
        if (A >= 0 && A + 16 <= cap) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(xm + A); q[0] = v.x; q[1] = v.y; }
        if (sh != 0 && A + 16 >= 0 && A + 32 <= cap) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(xm + A + 16); q[2] = v.x; q[3] = v.y; }
        const bool up = sh >= 8;                           // bytes sh .. sh + 15 of the 32
        const unsigned long long a0 = up ? q[1] : q[0], a1 = up ? q[2] : q[1], a2 = up ? q[3] : q[2];
        const int t = 8 * (sh & 7);
        const unsigned long long r0 = t ? (a0 >> t) | (a1 << (64 - t)) : a0, r1 = t ? (a1 >> t) | (a2 << (64 - t)) : a1;
        r.ww[4 * j] = (uint32_t)r0; r.ww[4 * j + 1] = (uint32_t)(r0 >> 32); r.ww[4 * j + 2] = (uint32_t)r1; r.ww[4 * j + 3] = (uint32_t)(r1 >> 32);
      } else {
        const MhlU4u w = *reinterpret_cast<const MhlU4u *>(xm + g);
        r.ww[4 * j] = w.x; r.ww[4 * j + 1] = w.y; r.ww[4 * j + 2] = w.z; r.ww[4 * j + 3] = w.w;
      }
    }
  }
  return r;
}

// The same through a raw buffer descriptor over [base, base + num_records) whose offsets are 32 bits wide: a chunk the row
// does not reach is sent to an offset past num_records, which the hardware answers with zeros -- no registers to clear and
// no branches, one compare and one select per chunk.  For lanes whose W bytes lie inside the descriptor (the caller sends a
// wavefront with any other lane through mhlf_load).
typedef uint32_t MhlfU4 __attribute__((ext_vector_type(4)));
template <int C>
__device__ __forceinline__ ChunkRaw<C> mhlf_load_buf(__amdgpu_buffer_rsrc_t rs, int32_t g32, int32_t lo0, int32_t hi0, bool valid) {
  constexpr int W = 16 * C;
  ChunkRaw<C> r;
  const int32_t lo = lo0 < 0 ? 0 : lo0, hi = hi0 > W ? W : hi0;
  const bool any = valid && hi > lo;
  r.lo = any ? lo : 0; r.hi = any ? hi : 0;
  const int32_t hiv = any ? hi0 : 0;
#pragma unroll
  for (int j = 0; j < C; j++) {
    // (16 j travels as the scalar offset, which the range check leaves out: an out-of-range lane stays out of range, and
    //  the caller has made sure an in-range lane's W bytes all lie inside the descriptor -- two VALU per chunk)
    const int32_t vo = 16 * j < hiv ? g32 : (int32_t)0x80000000u;
    const MhlfU4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 16 * j, 0);
    r.ww[4 * j] = w.x; r.ww[4 * j + 1] = w.y; r.ww[4 * j + 2] = w.z; r.ww[4 * j + 3] = w.w;
  }
  return r;
}

// Bytes of the lane outside the row -> 0 (a code without any flag).  Two places can hold them: the first chunk of the
// row's first lane (bytes before lo < 16) and the chunk the row ends in (bytes from hi on); the byte masks of a chunk
// come from one 64-bit shift and two selects, and the end mask is built only for the chunk slots in which some row of
// the wavefront ends (rows of similar length end in the same slot).
template <int C>
__device__ __forceinline__ void mhlf_zero_outside(ChunkRaw<C> &r) {
  constexpr int W = 16 * C;
  {
    const unsigned long long x = ~0ull << ((8 * r.lo) & 63);         // r.lo in [0, 15]
    const bool low = r.lo < 8;
    const unsigned long long a = low ? x : 0ull, b = low ? ~0ull : x;
    r.ww[0] &= (uint32_t)a; r.ww[1] &= (uint32_t)(a >> 32); r.ww[2] &= (uint32_t)b; r.ww[3] &= (uint32_t)(b >> 32);
  }
  const int je = (r.hi - 1) >> 4, rem = r.hi - 16 * je;               // rem in [1, 16] for a lane that holds bytes
  const bool cut = r.hi > r.lo && r.hi < W && rem < 16;
  const int sh = 8 * (16 - rem);                                      // 0 .. 120
  const unsigned long long y = ~0ull >> (sh & 63);
  const bool high = sh < 64;
  const unsigned long long ea = high ? ~0ull : y, eb = high ? y : 0ull;
#pragma unroll
  for (int j = 0; j < C; j++) {
    const bool last = cut && je == j;
    if (__ballot(last) != 0ull) {
      const uint32_t keep = last ? 0u : 0xFFFFFFFFu;
      r.ww[4 * j] &= (uint32_t)ea | keep; r.ww[4 * j + 1] &= (uint32_t)(ea >> 32) | keep;
      r.ww[4 * j + 2] &= (uint32_t)eb | keep; r.ww[4 * j + 3] &= (uint32_t)(eb >> 32) | keep;
    }
  }
}

// Member (U) and in-context (N = member or cut) bit planes of the lane's bytes, and the in-context flags of every pair
// of dwords (bit 3 of a byte: first dword, bit 7: second) for the call counters.
//   t = (w ^ k) & 7 per byte is 0 iff the byte's code is k or k | 8; t + 7 has bit 3 set iff t != 0.  The case bit of a
//   code is its bit 3: methylated = in context and bit 3 clear.
template <int C>
__device__ __forceinline__ void mhlf_planes(const uint32_t (&ww)[4 * C], uint32_t k7, uint32_t (&npair)[2 * C],
                                            typename MaskOf<C>::T &U, typename MaskOf<C>::T &N) {
  using M = typename MaskOf<C>::T;
  uint32_t ulo = 0, uhi = 0, nlo = 0, nhi = 0;
  const uint32_t k77 = k7 | (k7 << 4);
#pragma unroll
  for (int e = 0; e < 2 * C; e++) {
    // the codes of the two dwords as the nibbles of one word (first dword low): one compare serves eight bytes
    const uint32_t pk = (ww[2 * e] & 0x0F0F0F0Fu) | ((ww[2 * e + 1] << 4) & 0xF0F0F0F0u);
    const uint32_t z = ((pk & 0x77777777u) ^ k77) + 0x77777777u;                   // bit 3 of a nibble: code NOT of the context
    const uint32_t np = ~z & 0x88888888u;
    const uint32_t up = np & ~pk;
    npair[e] = np;
    asm("" : "+v"(npair[e]));                              // (one register per pair: keeps the compiler from carrying both sums instead)
    // byte = 8 * flag(first dword) + 128 * flag(second): weights 1, 2, 4, 8 give (mask byte) << 3
    const uint32_t nb = __builtin_amdgcn_udot4(np, 0x08040201u, 0u, false);
    const uint32_t ub = __builtin_amdgcn_udot4(up, 0x08040201u, 0u, false);
    const int q = e & 3;
    if (e < 4) { nlo |= q ? nb << (8 * q - 3) : nb >> 3; ulo |= q ? ub << (8 * q - 3) : ub >> 3; }
    else { nhi |= q ? nb << (8 * q - 3) : nb >> 3; uhi |= q ? ub << (8 * q - 3) : ub >> 3; }
  }
  U = (M)ulo; N = (M)nlo;
  if constexpr (sizeof(M) == 8) { U |= (M)uhi << 32; N |= (M)nhi << 32; }
}

// 16-entry byte LUT lookup of the four codes of a dword in two v_perm_b32 (F: a table in lut16_xor_form, common.hpp)
__device__ __forceinline__ uint32_t mhlf_lut4(uint32_t w, const MhlLut &F) {
  const uint32_t sel = w & 0x0F0F0F0Fu;
  return __builtin_amdgcn_perm(F.lo1, F.lo0, sel) ^ __builtin_amdgcn_perm(F.hi1, F.hi0, sel ^ 0x08080808u);
}

// Out-of-context counts of the lane and its rare-code flags: the LUT byte's bits 0, 2, 4, 6 are 2-bit fields (three
// dwords add without a carry); fields 0 and 4 (the counts) go on as 4-bit fields (<= 12) into v_sad_u8, fields 2 and 6
// (skipped, stray) are only OR-ed.  Returns oom | oou << 8 | (any skipped) << 16 | (any stray) << 24.
template <int C>
__device__ __forceinline__ uint32_t mhlf_counts(const uint32_t (&ww)[4 * C], const MhlLut &lut2) {
  uint32_t E = 0, rare = 0, oom = 0, oou = 0;
#pragma unroll
  for (int g = 0; g < (4 * C + 2) / 3; g++) {
    uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) if (3 * g + k < 4 * C) t += mhlf_lut4(ww[3 * g + k], lut2);
    E += t & 0x33333333u;
    rare |= t;
    if (g % 4 == 3 || g == (4 * C + 2) / 3 - 1) {
      oom = __builtin_amdgcn_sad_u8(E & 0x0F0F0F0Fu, 0u, oom);
      oou = __builtin_amdgcn_sad_u8((E >> 4) & 0x0F0F0F0Fu, 0u, oou);
      E = 0;
    }
  }
  return oom | (oou << 8) | ((rare & 0x0C0C0C0Cu) ? 1u << 16 : 0u) | ((rare & 0xC0C0C0C0u) ? 1u << 24 : 0u);
}

// bit plane "code == c" (c: 4 bits) of the lane's bytes: t = (w ^ c) & 15 is 0 iff equal, t + 15 has bit 4 set iff not
template <int C>
__device__ __forceinline__ typename MaskOf<C>::T mhlf_eq_plane(const uint32_t (&ww)[4 * C], uint32_t c) {
  using M = typename MaskOf<C>::T;
  const uint32_t c4 = c * 0x01010101u;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int e = 0; e < 2 * C; e++) {
    const uint32_t za = ((ww[2 * e] ^ c4) & 0x0F0F0F0Fu) + 0x0F0F0F0Fu, zb = ((ww[2 * e + 1] ^ c4) & 0x0F0F0F0Fu) + 0x0F0F0F0Fu;
    const uint32_t eq = ((~za >> 4) & 0x01010101u) | (~zb & 0x10101010u);           // flag of the first dword in bit 0, second in bit 4
    const uint32_t by = __builtin_amdgcn_udot4(eq, 0x08040201u, 0u, false);      // flag(first) + 16 * flag(second) per byte
    const int q = e & 3;
    if (e < 4) lo |= by << (8 * q); else hi |= by << (8 * q);
  }
  M r = (M)lo;
  if constexpr (sizeof(M) == 8) r |= (M)hi << 32;
  return r;
}

// span bits with the two segmented fills done by carry propagation: adding the member bits to the mask of non-cut bytes
// lets a carry run upward through a segment until the next cut absorbs it; the bits it flips (plus the members
// themselves) are the bytes at or above a member of their segment.  The downward fill is the same on the bit-reversed
// words.  (Bytes outside the row are neither members nor cuts; no member lies beyond them, so one of the two fills is
// empty there.)
__device__ __forceinline__ uint64_t mhlf_fill_up(uint64_t x, uint64_t m) {       // m: propagatable bits, x subset of m
  return ((((x + m) ^ m) & m) | x);
}
template <int W, class M>
__device__ __forceinline__ M mhlf_span_bits(M U, M L, M K, uint32_t enter, uint32_t cont) {
  const uint64_t nl = (uint64_t)(~L & bm_below<M>(W));                           // non-cut bytes of the lane
  const uint64_t x = (uint64_t)U | ((enter > 0u && !(L & (M)1)) ? 1ull : 0ull);
  const uint64_t y = (uint64_t)U | ((cont > 0u && !((L >> (W - 1)) & (M)1)) ? (1ull << (W - 1)) : 0ull);
  const uint64_t up = mhlf_fill_up(x, nl);
  const uint64_t dn = __brevll(mhlf_fill_up(__brevll(y), __brevll(nl)));
  return (M)(up & dn & nl) & ~K;
}

// calls fn(first bit, length, m) for every run of set bits; for stretches m = members of the run's segment (the lanes to
// the left / right contribute `enter` / `cont` when the segment reaches the lane's edge)
template <int W, class M, class FN>
__device__ __forceinline__ void mhlf_for_runs(M bits_, bool stretch, M U, M L, M K, uint32_t enter, uint32_t cont, FN fn) {
  using X = typename std::conditional<(W <= 32), uint32_t, uint64_t>::type;      // (a 32-byte block: 32-bit arithmetic)
  X bits = (X)bits_;
  while (bits) {
    const X low = bits & ((X)0 - bits);
    const X run = ((bits + low) ^ bits) & bits;                                  // the maximal run starting at `low`
    const int f = bm_ctz(low), e = bm_popc(run);
    uint32_t m = 0;
    if (stretch) {
      if (__builtin_expect(K != (M)0, 0)) {
        // skipped bytes may split the span of a segment into several runs: the members of the whole segment
        const uint64_t nl = (uint64_t)(~L & bm_below<M>(W)), rnl = __brevll(nl);
        const uint64_t up = mhlf_fill_up((uint64_t)low, nl);
        const uint64_t dn = __brevll(mhlf_fill_up(__brevll((uint64_t)low), rnl));
        const uint64_t seg = up | dn;
        m = ((seg & 1ull) ? enter : 0u) + (uint32_t)__popcll((uint64_t)U & seg) + (((seg >> (W - 1)) & 1ull) ? cont : 0u);
      } else {
        // the span of a segment is one run from its first to its last member; it reaches a lane edge iff the segment
        // goes on there (and `enter` / `cont` is 0 when it does not)
        m = (uint32_t)bm_popc((X)U & run) + (f == 0 ? enter : 0u) + (f + e == W ? cont : 0u);
      }
    }
    fn(f, e, m);
    bits ^= run;
  }
}

// S(min(m, H)) (mhl_lookup[m], :110-116): from the workgroup's table while every lane's index is below MHLF_STAB
__device__ __forceinline__ unsigned long long mhlf_S(uint32_t m, uint32_t H, const uint32_t *tab) {
  const uint32_t k = m < H ? m : H;
  if (__builtin_expect(__ballot(k >= (uint32_t)MHLF_STAB) != 0ull, 0)) return mhl_lut(m, H);
  return tab[k];
}

// +v on tile positions [a, b) of one difference array of MHLF_T entries
template <class X>
__device__ __forceinline__ void mhlf_interval(X *d, int a, int b, unsigned long long v) {
  if (a < 0) a = 0;
  if (a < b && a < MHLF_T) {
    atomicAdd(d + a, (X)v);
    if (b < MHLF_T) atomicAdd(d + b, (X)0 - (X)v);
  }
}

// segmented-scan element as one word: bit 31 = saw a cut, low bits = members since the last cut
__device__ __forceinline__ uint32_t mhlf_seg(uint32_t first, uint32_t second) {   // state after `first` then `second`
  return (second >> 31) ? second : first + second;
}
template <int G, int D>
__device__ __forceinline__ void mhlf_seg_scan(uint32_t &pf, uint32_t &sf, int sub) {
  if constexpr (D < G) {
    const uint32_t l = grp_up<G, D>(pf), r = grp_down<G, D>(sf);
    if (sub >= D) pf = mhlf_seg(l, pf);
    if (sub + D < G) sf = mhlf_seg(r, sf);                   // walking leftwards: `r` was seen first
    mhlf_seg_scan<G, D * 2>(pf, sf, sub);
  }
}

// u8 call counters -> the wide ones (u16 pairs / u32), every MHLF_FOLD rows
template <bool WIDE, int WG>
__device__ __forceinline__ void mhlf_fold(uint32_t *s_n8, uint32_t *s_nw) {
  for (int i = threadIdx.x; i < 2 * MHLF_Q; i += WG) {
    const uint32_t v = s_n8[i];
    if (v == 0u) continue;
    s_n8[i] = 0u;
    if constexpr (WIDE) {
      uint4 *w = reinterpret_cast<uint4 *>(s_nw + 4 * i);
      uint4 c = *w;
      c.x += v & 255u; c.y += (v >> 8) & 255u; c.z += (v >> 16) & 255u; c.w += v >> 24;
      *w = c;
    } else {
      uint2 *w = reinterpret_cast<uint2 *>(s_nw + 2 * i);
      uint2 c = *w;
      c.x += (v & 255u) | ((v & 0xFF00u) << 8);
      c.y += ((v >> 16) & 255u) | ((v >> 24) << 16);
      *w = c;
    }
  }
}

// in-place inclusive prefix sum of arr[0 .. T) by ONE wavefront: lane l owns the quads (64 j + l) * 4 .. + 3
template <class X>
__device__ __forceinline__ void mhlf_scan_array(X *arr, int lane) {
  X carry = 0;
#pragma unroll
  for (int j = 0; j < MHLF_T / 256; j++) {
    X *p = arr + (64 * j + lane) * 4;
    X v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    v1 += v0; v2 += v1; v3 += v2;
    const X inc = mhl_wave_scan<X>(v3);
    const X base = inc - v3 + carry;
    p[0] = v0 + base; p[1] = v1 + base; p[2] = v2 + base; p[3] = v3 + base;
    if constexpr (sizeof(X) == 8) {
      carry += ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(inc >> 32), 63) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)inc, 63);
    } else {
      carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
  }
}

// Prefix sums of the difference arrays, the rule (a row iff n > cov/2), ordered rows into the tile's pool slot.
//  fast layout: s_cov u32 [T] ('+' in the low half, '-' in the high), calls in s_n8 (u8) or, folded, s_nw (u16 pairs)
//  WIDE layout: s_cov u32 [2][T], calls in s_n8 or, folded, s_nw u32 [2][T]
template <bool WIDE, class ST, int WG, bool NW32 = WIDE>
__device__ __forceinline__ void mhlf_emit(const MhlFArgs &a, int tile, bool folded, const uint32_t *s_n8, const uint32_t *s_nw,
                                          uint32_t *s_cov, ST *s_sum, uint32_t *s_scan) {
  constexpr int T = MHLF_T, Q = MHLF_Q, NW = WG / 64, PPT = T / WG;
  static_assert(PPT == 2 || PPT == 4 || PPT == 8, "emit layout: 2, 4 or 8 consecutive positions per thread");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // difference arrays -> sums, in place: the arrays are dealt round-robin to the wavefronts
  constexpr int NA = WIDE ? 8 : 7;
  for (int k = wave; k < NA; k += NW) {
    if constexpr (WIDE) {
      if (k < 2) mhlf_scan_array<uint32_t>(s_cov + k * T, lane);
      else mhlf_scan_array<ST>(s_sum + (k - 2) * T, lane);
    } else {
      if (k == 0) mhlf_scan_array<uint32_t>(s_cov, lane);
      else mhlf_scan_array<ST>(s_sum + (k - 1) * T, lane);
    }
  }
  __syncthreads();
  // PPT consecutive positions per thread, both strands; key order: position, then '+' before '-'
  const int p0 = PPT * (int)threadIdx.x;
  auto calls = [&](int p, int sd) -> uint32_t {
    if (!folded) return (s_n8[sd * Q + (p >> 2)] >> (8 * (p & 3))) & 255u;
    if constexpr (NW32) return s_nw[sd * T + p];
    else return (s_nw[sd * (T / 2) + (p >> 1)] >> (16 * (p & 1))) & 0xFFFFu;
  };
  uint32_t okm = 0, nr = 0;                                // bit 2 j + s: cell (p0 + j, strand s) is a row of the table
#pragma unroll
  for (int j = 0; j < PPT; j++) {
#pragma unroll
    for (int sd = 0; sd < 2; sd++) {
      const int p = p0 + j;
      const uint32_t n = calls(p, sd);
      uint32_t c;
      if constexpr (WIDE) c = s_cov[sd * T + p];
      else { const uint32_t v = s_cov[p]; c = sd ? v >> 16 : v & 0xFFFFu; }
      const bool ok = n > (c >> 1);                                              // :76-86
      okm |= ok ? 1u << (2 * j + sd) : 0u;
      nr += ok;
    }
  }
  const uint32_t inc = wave_scan_u32(nr);
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int w = 0; w < NW; w++) { const uint32_t t = s_scan[w]; s_scan[w] = acc; acc += t; }
    uint32_t base = 0;
    bool fits = true;
    if (acc) {
      if (acc <= a.slot_rows) base = (uint32_t)tile * a.slot_rows;
      else {
        const uint32_t o = atomicAdd(a.cursor, acc);
        fits = (uint64_t)a.ovf_base + o + acc <= a.pool_cap;
        base = a.ovf_base + o;
      }
    }
    s_scan[NW + 1] = base;
    s_scan[NW] = fits ? acc : 0xFFFFFFFFu;
    a.tile_nrow[tile] = acc;
    a.tile_base[tile] = base;
  }
  __syncthreads();
  const uint32_t total = s_scan[NW], base = s_scan[NW + 1];
  if (total == 0xFFFFFFFFu || total == 0u) return;
  // The rows of a tile are few (a few per cent of its cells): the cells are listed in key order -- in the coverage array,
  // which nobody reads any more -- and written densely, one row per lane with consecutive addresses, instead of every
  // thread walking its own eight cells.
  uint16_t *list = reinterpret_cast<uint16_t *>(s_cov);
  {
    uint32_t k = s_scan[wave] + inc - nr;
#pragma unroll
    for (int i = 0; i < 2 * PPT; i++) {
      if ((okm >> i) & 1u) { list[k] = (uint16_t)(2 * (p0 + (i >> 1)) + (i & 1)); k++; }
    }
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < total; j += WG) {
    const uint32_t id = list[j];
    const int p = (int)(id >> 1), sd = (int)(id & 1u);
    const uint32_t w = base + j;
    if (EPI_DEV_CHECK(a.dbg, w < a.pool_cap, 33, w, a.pool_cap)) {
      a.pool_key[w] = ((uint32_t)p << 4) | ((uint32_t)sd << 3) | a.ctx;
      a.pool_cov[w] = calls(p, sd);                                              // coverage column, :90
      a.pool_nu[w] = (unsigned long long)s_sum[(0 + sd) * T + p];                // :93 numerator
      a.pool_hs[w] = (unsigned long long)s_sum[(2 + sd) * T + p];                // :92 numerator
      a.pool_de[w] = (unsigned long long)s_sum[(4 + sd) * T + p];                // :93 denominator
    }
  }
}

// raw arrays of a tile that other ranks contribute to -> its slot of the slabs (summed across ranks by the caller)
template <bool WIDE, class ST, int WG, bool NW32 = WIDE>
__device__ __forceinline__ void mhlf_dump_slab(const MhlFArgs &a, int slot, bool folded, const uint32_t *s_n8, const uint32_t *s_nw,
                                               const uint32_t *s_cov, const ST *s_sum) {
  constexpr int T = MHLF_T, Q = MHLF_Q;
  int32_t *cnt = a.slab_cnt + (int64_t)slot * (MHLF_CNT_PLANES * T);
  unsigned long long *sum = a.slab_sum + (int64_t)slot * (MHLF_SUM_PLANES * T);
  for (int i = threadIdx.x; i < 2 * T; i += WG) {
    const int s = i / T, p = i % T;
    uint32_t n;
    if (!folded) n = (s_n8[s * Q + (p >> 2)] >> (8 * (p & 3))) & 255u;
    else if constexpr (NW32) n = s_nw[s * T + p];
    else n = (s_nw[s * (T / 2) + (p >> 1)] >> (16 * (p & 1))) & 0xFFFFu;
    if (n) atomicAdd(cnt + s * T + p, (int32_t)n);
    int32_t d;
    if constexpr (WIDE) d = (int32_t)s_cov[s * T + p];
    else {
      const uint32_t v = s_cov[p];
      const int32_t lo = (int32_t)(int16_t)(v & 0xFFFFu);                        // both halves are signed before the prefix sum
      d = s ? ((int32_t)v - lo) >> 16 : lo;
    }
    if (d) atomicAdd(cnt + (2 + s) * T + p, d);
  }
  for (int i = threadIdx.x; i < MHLF_SUM_PLANES * T; i += WG) {
    const ST v = s_sum[i];
    if (v == (ST)0) continue;
    unsigned long long x;
    if constexpr (WIDE) x = (unsigned long long)v; else x = (unsigned long long)(long long)(int32_t)v;   // (|entry| < 2^31: checked)
    atomicAdd(sum + i, x);
  }
}

template <bool WIDE> constexpr int mhlf_lds_words() { return WIDE ? 2 * MHLF_T : MHLF_T; }

// What one block of a lane's bytes (W of them, starting at tile position P0) adds for a kept row: S(M) over its pieces of
// the methylated stretches, the call counters of its pairs of dwords and -- rows with skipped bytes -- h and S(h) per
// counted run and the coverage correction.
template <class ST> struct MhlfRow {
  const uint32_t *tab;                    // S(k) for k < MHLF_STAB
  ST *dn, *dh, *dd;                       // difference arrays of S(M), h, S(h) of the row's strand
  uint32_t *covp, *n8;                    // coverage array (of the strand, WIDE) and u8 call counters of the strand
  uint32_t unit, h, H;
  unsigned long long sh;                  // S(h)
  bool anyk;
};
template <int W, int NPAIR, class M, class ST>
__device__ __forceinline__ void mhlf_block(const MhlfRow<ST> &c, M U, M L, M K, uint32_t enter, uint32_t cont, int32_t P0,
                                           const uint32_t (&np)[NPAIR], int vlo, int vhi) {
  // stretches: S(M) on every counted byte between the first and the last member (:168-171, :193)
  const M P = (EPI_MHLF_ABLATE & 4) ? (M)0 : mhlf_span_bits<W, M>(U, L, K, enter, cont);
  mhlf_for_runs<W, M>(P, true, U, L, K, enter, cont, [&](int f, int e, uint32_t m) { mhlf_interval(c.dn, P0 + f, P0 + f + e, mhlf_S(m, c.H, c.tab)); });
  // calls of the context: u8 counters, one LDS atomic per two dwords of xm that hold any
  unsigned long long *n8 = reinterpret_cast<unsigned long long *>(c.n8 + (P0 >> 2));
#pragma unroll
  for (int e = 0; e < NPAIR; e++) {                        // (P0 is a multiple of 16: the two pairs of a chunk are in or out of the tile together)
    if (np[e] != 0u && (uint32_t)((P0 >> 2) + 4 * (e >> 1)) < (uint32_t)MHLF_Q && !(EPI_MHLF_ABLATE & 8))
      atomicAdd(n8 + e, (unsigned long long)((np[e] >> 3) & 0x01010101u) | ((unsigned long long)((np[e] >> 7) & 0x01010101u) << 32));
  }
  if (__builtin_expect(c.anyk, 0)) {
    // reads with skipped bytes: h and S(h) per counted run, coverage -1 over the skipped runs
    const M V = vhi > vlo ? bm_below<M>(vhi) & ~bm_below<M>(vlo) : (M)0;       // (lanes behind the row's end: nothing)
    mhlf_for_runs<W, M>(V & ~K, false, U, L, K, enter, cont, [&](int f, int e, uint32_t) {
      mhlf_interval(c.dh, P0 + f, P0 + f + e, (unsigned long long)c.h);
      mhlf_interval(c.dd, P0 + f, P0 + f + e, c.sh);
    });
    mhlf_for_runs<W, M>(K, false, U, L, K, enter, cont, [&](int f, int e, uint32_t) { mhlf_interval(c.covp, P0 + f, P0 + f + e, (unsigned long long)(0u - c.unit)); });
  }
}

// segmented-scan elements of one block: (saw a cut, members after the last cut) and (saw a cut, members before the first)
template <class M>
__device__ __forceinline__ void mhlf_block_seg(M U, M L, uint32_t &pf, uint32_t &sf) {
  const uint32_t has = L ? 0x80000000u : 0u;
  pf = has | (uint32_t)bm_popc(U & (L ? ~bm_below<M>(bm_msb(L) + 1) : ~(M)0));
  sf = has | (uint32_t)bm_popc(U & (L ? ((L & ((M)0 - L)) - (M)1) : ~(M)0));
}

template <bool WIDE, int WG, bool FOLD> constexpr int mhlf_wps() {      // waves per SIMD the register allocation aims at
  return WIDE ? (WG >= 512 ? 4 : 2) : (WG >= 512 ? EPI_MHLF_WPS : (FOLD ? 4 : 5));      // (what the LDS of a workgroup allows)
}

// G lanes own a row.  A lane holds 16 * (CA + CB) contiguous bytes as one or two mask blocks: CB = 0 is one block of up to
// 64 bytes (64-bit masks); CA = 3, CB = 2 is 80 bytes as a 48-byte and a 32-byte block, each with masks of its own, the
// two combined like two lanes before the lane scans -- 4 lanes x 80 bytes hold a PE150 template (16 rows per wavefront
// step where 8 lanes x 48 bytes hold 8: the work that is per lane, not per byte, halves per row).
template <int G, int CA, int CB, bool WIDE, int WG, bool FOLD = true>
__global__ __launch_bounds__(WG, (mhlf_wps<WIDE, WG, FOLD>())) void k_mhl_fused(MhlFArgs a, int ntiles) {
  static_assert(FOLD || !WIDE, "only the fast variant is built without the folded counters");
  static_assert(!(WIDE && CB), "the WIDE variant is built with one block per lane");
  using MA = typename MaskOf<CA>::T;
  using MB = uint32_t;                                     // CB <= 2
  using ST = typename std::conditional<WIDE, unsigned long long, uint32_t>::type;
  constexpr int C = CA + CB, WA = 16 * CA, WB = 16 * CB, W = 16 * C, T = MHLF_T, Q = MHLF_Q, R = 64 / G, NW = WG / 64;
  __shared__ __attribute__((aligned(16))) uint32_t s_n8[2 * Q];                      // [strand][Q]: calls of the context, u8 x 4 positions
  __shared__ __attribute__((aligned(16))) uint32_t s_nw[FOLD ? mhlf_lds_words<WIDE>() : 4];   // the same, folded every 255 rows.  !FOLD:
  // 4 KB less LDS (a fifth workgroup per CU); the rare tile with more than 255 rows folds into a slot of a slab in HBM
  __shared__ __attribute__((aligned(16))) uint32_t s_cov[mhlf_lds_words<WIDE>()];    // coverage difference array(s)
  __shared__ __attribute__((aligned(16))) ST s_sum[6 * T];                           // [S(M), h, S(h)][strand][T] difference arrays
  __shared__ uint32_t s_scan[NW + 2];
  __shared__ uint32_t s_tab[MHLF_STAB];                                              // S(k), k < MHLF_STAB (:110-116)
  __shared__ uint32_t s_hmax;                                                        // largest h among the kept rows; 0xFFFFFFFF: a stray code
#ifdef EPI_MHLF_LDS_PAD                                    // timing builds: LDS nobody uses (how many workgroups fit a CU)
  __shared__ uint32_t s_pad[EPI_MHLF_LDS_PAD / 4];
  if (a.xm_cap == -12345) s_pad[threadIdx.x % (EPI_MHLF_LDS_PAD / 4)] = 1;
#endif
  int tile;
  if (a.tile_list) {
    if ((int)blockIdx.x >= ntiles) return;
    tile = (int)a.tile_list[blockIdx.x];
  } else {
    const int chunk = (ntiles + 7) >> 3;                   // XCD-aware tile order, as the CX kernel
    tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
  }
  const Tile td = a.tiles[tile];
  const int nrows = td.row_hi - td.row_lo;
  if (!WIDE && nrows > a.max_rows) {                       // too many rows for the packed coverage halves (or the test hook)
    if (threadIdx.x == 0) { a.deep_list[atomicAdd(a.deep_count, 1u)] = (uint32_t)tile; a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
    return;
  }
  uint32_t *nw = s_nw;                                     // folded call counters (LDS, or the tile's slot of the slab)
  if constexpr (!FOLD) {
    if (nrows > MHLF_FOLD) {
      if (threadIdx.x == 0) s_scan[0] = atomicAdd(a.fold_cursor, 1u);
      __syncthreads();
      const uint32_t fs = s_scan[0];
      if (fs >= a.fold_slots) {                            // (no slot left: the WIDE variant takes the tile)
        if (threadIdx.x == 0) { a.deep_list[atomicAdd(a.deep_count, 1u)] = (uint32_t)tile; a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
        return;
      }
      nw = a.fold_slab + (size_t)fs * (2 * T);
      uint4 *z = reinterpret_cast<uint4 *>(nw);            // (thread i zeroes the words it folds into: same index map as mhlf_fold)
      for (int i = threadIdx.x; i < 2 * T / 4; i += WG) z[i] = make_uint4(0, 0, 0, 0);
    }
  }
  {
    uint4 *z = reinterpret_cast<uint4 *>(s_n8);
    for (int i = threadIdx.x; i < 2 * Q / 4; i += WG) z[i] = make_uint4(0, 0, 0, 0);
    if constexpr (FOLD) {
      if (nrows > MHLF_FOLD) {
        z = reinterpret_cast<uint4 *>(s_nw);
        for (int i = threadIdx.x; i < mhlf_lds_words<WIDE>() / 4; i += WG) z[i] = make_uint4(0, 0, 0, 0);
      }
    }
    z = reinterpret_cast<uint4 *>(s_cov);
    for (int i = threadIdx.x; i < mhlf_lds_words<WIDE>() / 4; i += WG) z[i] = make_uint4(0, 0, 0, 0);
    z = reinterpret_cast<uint4 *>(s_sum);
    for (int i = threadIdx.x; i < (int)(6 * T * sizeof(ST) / 16); i += WG) z[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) s_hmax = 0u;
    if (threadIdx.x < MHLF_STAB) {                         // S(k) = k (k + 1) (k + 2) / 6 (= k below 2): 32 bits hold it for k < 256
      const uint32_t k = threadIdx.x;
      s_tab[k] = (k * (k + 1u) * (k + 2u)) / 6u;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1), grp = lane / G;
  // the tile's rows through a buffer descriptor that starts 16 .. 31 bytes before the first of them (row offsets increase
  // with the row index); a tile whose rows span 2 GB or more -- millions of rows on one kilobase -- stays with 64-bit addresses
  const int64_t xm_lo = a.off[td.row_lo], xm_hi = a.off[td.row_hi];
  const int64_t buf_base = (xm_lo & ~(int64_t)15) >= 16 ? (xm_lo & ~(int64_t)15) - 16 : 0;
  const bool use_buf = xm_hi - buf_base < 0x7FFF0000ll;
  const int64_t buf_left = a.xm_cap - buf_base;
  const int32_t buf_n = buf_left < 0x7FFFFF00ll ? (int32_t)buf_left : 0x7FFFFF00;
  const __amdgpu_buffer_rsrc_t buf = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.xm) + buf_base, (short)0, buf_n, 0x00020000);
  uint32_t hmax = 0;                                       // largest h this lane has seen on a kept row (0xFFFFFFFF: a stray code)

  // ---- accumulate: the next step's row columns are fetched early; the u8 call counters are folded every MHLF_FOLD rows ----
  for (int blo = td.row_lo; blo < td.row_hi; blo += MHLF_FOLD) {
    const int bhi = td.row_hi - blo > MHLF_FOLD ? blo + MHLF_FOLD : td.row_hi;
    // Row columns are fetched one step ahead.  (The row's bytes a step ahead as well -- 106 VGPRs -- changed nothing: the
    // kernel does not wait for memory, profiles/r03_mhl_ablation.txt.)
    int64_t n_rs = 0, n_re = 0;                             // columns of the next row this lane loads bytes for
    int32_t n_st = 0, n_sd = 1;
    auto load_cols = [&](int rr) {
      n_rs = 0; n_re = 0; n_st = 0; n_sd = 1;
      if (rr < bhi) { n_rs = a.off[rr]; n_re = n_rs + a.len[rr]; n_st = a.start[rr]; n_sd = a.strand[rr]; }
    };
    struct Geo { int32_t rel, len, sd; bool valid; };
    // (a candidate row -- start within the longest row's reach in front of the tile -- that ends in front of the tile has nothing
    //  for it.  Fast variant: the lane shape is picked for the bulk of the rows, pick_mhlf_shape_hist; a row it cannot hold sends
    //  the tile to the WIDE variant, whose shape holds the batch's longest row.)
    auto geo_of = [&](int rr) {
      Geo g; g.rel = (int32_t)((uint32_t)n_st - (uint32_t)td.pos0); g.len = (int32_t)(n_re - n_rs); g.sd = n_sd;
      g.valid = rr < bhi && g.rel + g.len > 0;
      if constexpr (!WIDE) { if (g.valid && g.len + 15 > G * W) { g.valid = false; hmax = 0xFFFFFFFFu; } }
      return g;
    };
    auto load_bytes = [&](const Geo &g) {                   // (uses n_rs: call before the columns move on)
      const int32_t lo0 = (g.rel & 15) - sub * W;
      const int32_t g32 = (int32_t)((uint32_t)n_rs - (uint32_t)buf_base) - lo0;     // the lane's byte 0 in the descriptor
      // (a lane whose bytes reach outside the batch's buffer -- first and last rows only -- takes the wavefront along)
      const bool out = g.valid && (g32 < 0 || g32 > buf_n - W);
      if (__builtin_expect(use_buf && __ballot(out) == 0ull, 1)) return mhlf_load_buf<C>(buf, g32, lo0, lo0 + g.len, g.valid);
      return g.valid ? mhlf_load<C>(a.xm, a.xm_cap, n_rs - lo0, lo0, lo0 + g.len) : ChunkRaw<C>{{0}, 0, 0};
    };
    load_cols(blo + wave * R + grp);
    for (int rbase = blo + wave * R; rbase < bhi; rbase += NW * R) {
      const int r = rbase + grp;
      if (!EPI_DEV_CHECK(a.dbg, r >= bhi || (r >= 0 && r < a.nrows), 31, r, 0)) return;
      const Geo g = geo_of(r);
      ChunkRaw<C> raw = load_bytes(g);
      load_cols(r + NW * R);                                                  // (in flight with the bytes)
      const bool valid = g.valid;
      const int32_t rel = g.rel, len = g.len, sd = g.sd;                      // tile position of the row's byte 0, its bytes, strand
      if (!EPI_DEV_CHECK(a.dbg, !valid || (len >= 0 && len <= G * W), 32, r, len)) return;
      const int32_t P0 = rel - (rel & 15) + sub * W;                          // tile position of this lane's byte 0 (multiple of 16)
      const int32_t lo0 = (rel & 15) - sub * W, hi0 = lo0 + len;              // the row's bytes relative to the lane's byte 0
      if (EPI_MHLF_ABLATE & 2) {                                              // timing builds: loads only
        uint32_t x = 0;
#pragma unroll
        for (int d = 0; d < 4 * C; d++) x ^= raw.ww[d];
        if (x == 0x12345678u) atomicAdd(s_cov, 1u);
        continue;
      }
      mhlf_zero_outside<C>(raw);
      uint32_t npair[2 * C];
      const uint32_t (&wwA)[4 * CA] = reinterpret_cast<const uint32_t (&)[4 * CA]>(raw.ww[0]);
      uint32_t (&npA)[2 * CA] = reinterpret_cast<uint32_t (&)[2 * CA]>(npair[0]);
      MA UA, NA;
      mhlf_planes<CA>(wwA, a.k7, npA, UA, NA);
      MB UB = 0, NB = 0;
      if constexpr (CB > 0) {
        const uint32_t (&wwB)[4 * CB] = reinterpret_cast<const uint32_t (&)[4 * CB]>(raw.ww[4 * CA]);
        uint32_t (&npB)[2 * CB] = reinterpret_cast<uint32_t (&)[2 * CB]>(npair[2 * CA]);
        mhlf_planes<CB>(wwB, a.k7, npB, UB, NB);
      }
      const uint32_t cnts = mhlf_counts<C>(raw.ww, a.lut2);
      const MA LA = NA & ~UA;
      const MB LB = NB & ~UB;
      // rare codes, from the bytes while the lane still holds them: skipped ('+', '-', filler between mates: common where
      // mates do not meet) and -- WIDE variant only -- the stray nibbles 9 / 3 / 4 / 8, which ARE the reference's
      // coverage slot and the slots of its three sums (:190-194).  The fast variant hands a tile with a stray code over.
      MA KA = 0;
      MB KB = 0;
      if (__builtin_expect(__ballot((cnts >> 16) & 1u) != 0ull, 0)) {
        if ((cnts >> 16) & 1u) {
          KA = mhlf_eq_plane<CA>(wwA, 11u);
          if constexpr (CB > 0) KB = mhlf_eq_plane<CB>(reinterpret_cast<const uint32_t (&)[4 * CB]>(raw.ww[4 * CA]), 11u);
        }
      }
      MA dbl = 0, s3 = 0, s4 = 0, s8 = 0;
      if constexpr (WIDE) {
        if (__builtin_expect(__ballot((cnts >> 24) & 1u) != 0ull, 0)) {
          if ((cnts >> 24) & 1u) { dbl = mhlf_eq_plane<CA>(wwA, 9u); s3 = mhlf_eq_plane<CA>(wwA, 3u); s4 = mhlf_eq_plane<CA>(wwA, 4u); s8 = mhlf_eq_plane<CA>(wwA, 8u); }
        }
      }

      // members of the open segment to the left (enter) and to the right (cont) of each block: the lane's blocks are
      // combined like two lanes, the lanes scanned, and the blocks' own elements applied again
      uint32_t pfA, sfA, pfB = 0u, sfB = 0u;
      mhlf_block_seg<MA>(UA, LA, pfA, sfA);
      if constexpr (CB > 0) mhlf_block_seg<MB>(UB, LB, pfB, sfB);
      uint32_t pf = CB > 0 ? mhlf_seg(pfA, pfB) : pfA, sf = CB > 0 ? mhlf_seg(sfB, sfA) : sfA;
      mhlf_seg_scan<G, 1>(pf, sf, sub);
      uint32_t up = grp_up<G, 1>(pf), down = grp_down<G, 1>(sf);             // state of the lanes to the left / right
      if (sub == 0) up = 0u;
      if (sub == G - 1) down = 0u;
      const uint32_t enterA = up & 0x7FFFFFFFu, contLast = down & 0x7FFFFFFFu;
      const uint32_t enterB = mhlf_seg(up, pfA) & 0x7FFFFFFFu, contA = CB > 0 ? mhlf_seg(down, sfB) & 0x7FFFFFFFu : contLast;
      // row totals, two per word: h | oo_m << 16, oo_u | (lanes with skipped bytes) << 16 | (lanes with stray codes) << 24
      const uint32_t s1 = grp_sum<G / 2>((uint32_t)(bm_popc(NA) + bm_popc(NB)) | ((cnts & 255u) << 16));
      const uint32_t s2 = grp_sum<G / 2>(((cnts >> 8) & 255u) | (cnts & 0x01010000u));
      const uint32_t h = s1 & 0xFFFFu, oo_m = s1 >> 16, oo_u = s2 & 0xFFFFu;
      const bool anyk = (s2 & 0x00FF0000u) != 0u, anystray = (s2 >> 24) != 0u;
      const bool keep = valid && len > 0 && !((int)h < a.hmin) && oo_m < a.keep_tab[oo_m + oo_u];   // :176-179
      if (keep) {
        const int sidx = sd - 1;
        MhlfRow<ST> c;
        c.dn = s_sum + (0 + sidx) * T; c.dh = s_sum + (2 + sidx) * T; c.dd = s_sum + (4 + sidx) * T;
        c.covp = WIDE ? s_cov + sidx * T : s_cov;
        c.n8 = s_n8 + sidx * Q;
        c.unit = WIDE ? 1u : (sidx ? 65536u : 1u);
        c.h = h; c.H = a.H; c.anyk = anyk; c.tab = s_tab;
        c.sh = mhl_lut(h, a.H);                                                // S(h), :194
        if (sub == 0) {
          mhlf_interval(c.covp, rel, rel + len, c.unit);                       // coverage of the whole row; skipped bytes corrected per block
          if (!anyk) {                                                         // every byte counted: one interval per sum (:192, :194)
            mhlf_interval(c.dh, rel, rel + len, (unsigned long long)h);
            mhlf_interval(c.dd, rel, rel + len, c.sh);
          }
        }
        hmax = h > hmax ? h : hmax;
        const int vlo = lo0 < 0 ? 0 : lo0, vhi = hi0 > W ? W : hi0;            // the row's bytes of this lane: [vlo, vhi)
        mhlf_block<WA, 2 * CA, MA, ST>(c, UA, LA, KA, enterA, contA, P0, npA, vlo < WA ? vlo : WA, vhi < WA ? vhi : WA);
        if constexpr (CB > 0) {
          const uint32_t (&npB)[2 * CB] = reinterpret_cast<const uint32_t (&)[2 * CB]>(npair[2 * CA]);
          mhlf_block<WB, 2 * CB, MB, ST>(c, UB, LB, KB, enterB, contLast, P0 + WA, npB, vlo > WA ? vlo - WA : 0, vhi > WA ? vhi - WA : 0);
        }
        if (__builtin_expect(anystray, 0)) {
          if constexpr (WIDE) {
            // nibble 9 IS the reference's coverage slot (+1 more, :191); nibbles 3 / 4 / 8 are the slots of the sums of :193 / :194 / :192
            for (MA m = dbl; m; m &= m - 1) { const int p = P0 + bm_ctz(m); mhlf_interval(c.covp, p, p + 1, c.unit); }
            for (MA m = s3; m; m &= m - 1) { const int p = P0 + bm_ctz(m); mhlf_interval(c.dn, p, p + 1, 1ull); }
            for (MA m = s4; m; m &= m - 1) { const int p = P0 + bm_ctz(m); mhlf_interval(c.dd, p, p + 1, 1ull); }
            for (MA m = s8; m; m &= m - 1) { const int p = P0 + bm_ctz(m); mhlf_interval(c.dh, p, p + 1, 1ull); }
          } else {
            hmax = 0xFFFFFFFFu;                                                // the tile goes to the WIDE variant
          }
        }
      }
    }
    if constexpr (!WIDE) {                                 // (one LDS atomic per wavefront and block of rows)
      if (__ballot(hmax != 0u) != 0ull) {
        // maximum over the wavefront: four DPP steps inside a row of 16 lanes, then the four rows through SGPRs
        uint32_t m = hmax;
        { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true); m = o > m ? o : m; }    // quad_perm [1,0,3,2]
        { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true); m = o > m ? o : m; }    // quad_perm [2,3,0,1]
        { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, true); m = o > m ? o : m; }   // row_half_mirror
        { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xF, 0xF, true); m = o > m ? o : m; }   // row_mirror
        uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)m, 0);
#pragma unroll
        for (int rr = 1; rr < 4; rr++) { const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)m, 16 * rr); mw = o > mw ? o : mw; }
        if (lane == 0) atomicMax(&s_hmax, mw);
      }
    }
    if (bhi < td.row_hi || nrows > MHLF_FOLD) { __syncthreads(); mhlf_fold<WIDE || !FOLD, WG>(s_n8, nw); }   // (deep tiles only)
    __syncthreads();
  }
  const bool folded = nrows > MHLF_FOLD;
  if constexpr (!WIDE) {
    // u32 sums: exact while rows x (S(largest h) + 1) stays below 2^32 (2^31 for a shared tile: its entries travel as
    // signed differences); otherwise -- or when a kept row holds a stray code -- the tile is redone by the WIDE variant
    const unsigned long long lim = td.slot >= 0 ? (1ull << 31) : (1ull << 32);
    const uint32_t hm = s_hmax;
    if (hm == 0xFFFFFFFFu || (unsigned long long)nrows * (nrS(hm < a.H ? hm : a.H) + 1ull) >= lim) {
      if (threadIdx.x == 0) { a.deep_list[atomicAdd(a.deep_count, 1u)] = (uint32_t)tile; a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
      return;
    }
  }
  if (td.slot >= 0) {                                      // shared with another rank: hand the raw arrays over
    mhlf_dump_slab<WIDE, ST, WG, WIDE || !FOLD>(a, td.slot, folded, s_n8, nw, s_cov, s_sum);
    if (threadIdx.x == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; }
    return;
  }
  if (EPI_MHLF_ABLATE & 1) { if (threadIdx.x == 0) { a.tile_nrow[tile] = 0; a.tile_base[tile] = 0; } return; }   // timing builds: no emit
  mhlf_emit<WIDE, ST, WG, WIDE || !FOLD>(a, tile, folded, s_n8, nw, s_cov, s_sum, s_scan);
}

// Emits the shared tiles this rank owns from the (already cross-rank reduced) slabs: one workgroup per shared slot.
__global__ __launch_bounds__(MHLF_WG) void k_mhlf_emit_slab(MhlFArgs a, const int32_t *__restrict__ owned, const int32_t *__restrict__ slot_tile) {
  constexpr int T = MHLF_T;
  __shared__ __attribute__((aligned(16))) uint32_t s_nw[2 * T];
  __shared__ __attribute__((aligned(16))) uint32_t s_cov[2 * T];
  __shared__ __attribute__((aligned(16))) unsigned long long s_sum[6 * T];
  __shared__ uint32_t s_scan[MHLF_NW + 2];
  if (!owned[blockIdx.x]) return;
  const int tile = slot_tile[blockIdx.x];
  if (tile < 0) return;
  const int32_t *cnt = a.slab_cnt + (int64_t)blockIdx.x * (MHLF_CNT_PLANES * T);
  const unsigned long long *sum = a.slab_sum + (int64_t)blockIdx.x * (MHLF_SUM_PLANES * T);
  for (int i = threadIdx.x; i < 2 * T; i += MHLF_WG) { s_nw[i] = (uint32_t)cnt[i]; s_cov[i] = (uint32_t)cnt[2 * T + i]; }
  for (int i = threadIdx.x; i < 6 * T; i += MHLF_WG) s_sum[i] = sum[i];
  __syncthreads();
  mhlf_emit<true, unsigned long long, MHLF_WG>(a, tile, true, nullptr, s_nw, s_cov, s_sum, s_scan);
}

// ---- host side -------------------------------------------------------------------------------------------------------

// counting LUT of the fused kernel (MhlFArgs::lut2) for one context set (rcpp_mhl_report.cpp:104-107, :176-177, :187)
static MhlLut make_mhlf_lut2(uint32_t ctx_mask) {
  uint32_t w[4] = {0, 0, 0, 0};
  for (uint32_t code = 0; code < 16; code++) {
    const bool in = (ctx_mask >> code) & 1u;
    uint32_t f = 0;
    if (!in && ((0x00E4u >> code) & 1u)) f |= 1u;            // codes 2, 5, 6, 7 outside the context: methylated
    if (!in && ((0xE400u >> code) & 1u)) f |= 16u;           // codes 10, 13, 14, 15: unmethylated
    if (code == 11) f |= 4u;                                 // skipped (:187)
    if (code == 3 || code == 4 || code == 8 || code == 9) f |= 64u;   // their counters are sums / the coverage slot (:190-194)
    w[code >> 2] |= f << (8 * (code & 3));
  }
  ClassLut d;
  d.lo0 = w[0]; d.lo1 = w[1]; d.hi0 = w[2]; d.hi1 = w[3];
  const ClassLut x = lut16_xor_form(d);                      // (the form mhlf_lut4 looks up)
  MhlLut l;
  l.lo0 = x.lo0; l.lo1 = x.lo1; l.hi0 = x.hi0; l.hi1 = x.hi1;
  return l;
}

// Lane shape of the one-pass kernel as G * 100 + CA * 10 + CB: the smallest G * 16 * (CA + CB) that holds the longest row
// wherever it starts inside its first 16 bytes (ties: fewer lanes).  One block per lane (CB = 0: CA = 2, 3, 4, 512-thread
// workgroups) or two (CA = 3, CB = 2: 80 bytes per lane, 256-thread workgroups).
static int pick_mhlf_shape(int32_t max_len) {
  if (options().mhlf_shape > 0) {                          // A/B runs (EPIHIP_MHLF_SHAPE="G,CA[,CB]"); must cover the reads
    const int g = options().mhlf_shape / 100, ca = options().mhlf_shape / 10 % 10, cb = options().mhlf_shape % 10;
    if ((int64_t)g * 16 * (ca + cb) >= (int64_t)max_len + 15) return options().mhlf_shape;
  }
  int best = 0;
  int64_t best_cap = 0;
  for (int g = 2; g <= 64; g <<= 1)
    for (int k = 0; k < 4; k++) {
      const int ca = k < 3 ? 2 + k : 3, cb = k < 3 ? 0 : 2;
      if (cb && (g < 4 || g > 32)) continue;
      const int64_t cap = (int64_t)g * 16 * (ca + cb);
      if (cap < (int64_t)max_len + 15) continue;
      if (!best || cap < best_cap) { best = g * 100 + ca * 10 + cb; best_cap = cap; }
    }
  return best;
}

// The fast variant's lane shape from the batch's length histogram: a shape that holds rows of up to `cap` bytes costs ~ cap per
// row (+ its share of the step); the tiles that see a longer row are done again by the WIDE variant with the shape for the
// longest row.  One 1 kb template among 100 000 PE150 ones used to make every row pay for 1 kb (scratch/outlier_cost.py).
static int pick_mhlf_shape_hist(const RowStats &st, int64_t n, int32_t nt, int T) {
  const int wide = pick_mhlf_shape(st.max_len);
  if (options().mhlf_shape) return wide;
  double total = 0;
  for (int k = 0; k < kLenBinCount; k++) total += st.len_hist[k];
  if (total == 0 || nt <= 0 || !wide) return wide;
  auto cost_of = [](int shape) { const int g = shape / 100, w = 16 * (shape / 10 % 10 + shape % 10); return (double)g * w + 8.0 * g; };
  const double rows_per_tile = (double)n / nt, tiles_per_long = 1.0 + (double)st.max_len / T;
  int best = wide;
  double best_cost = cost_of(wide);
  for (int k = 0; k < kLenBinCount - 1; k++) {
    const int32_t len_k = kLenBins[k] * 16 - 15;             // the longest row of bin k: (len + 30) / 16 <= bound
    if (len_k >= st.max_len) break;
    const int shape = pick_mhlf_shape(len_k);
    if (!shape) continue;
    const int64_t cap = (int64_t)(shape / 100) * 16 * (shape / 10 % 10 + shape % 10);
    double longer = 0;                                       // rows this shape cannot hold: len + 15 > cap
    for (int j = 0; j < kLenBinCount; j++) if ((int64_t)kLenBins[j] * 16 > cap) longer += st.len_hist[j];
    double aside = longer / total * rows_per_tile * tiles_per_long;
    if (aside > 1.0) aside = 1.0;
    const double c = cost_of(shape) + aside * 1.5 * cost_of(wide);
    if (c < best_cost) { best = shape; best_cost = c; }
  }
  return best;
}

template <bool WIDE>
static void launch_mhl_fused(int shape, unsigned grid, int nt, hipStream_t s, const MhlFArgs &a, bool fold = true) {
  const int g0 = shape / 100, ca = shape / 10 % 10, cb = shape % 10;
  if constexpr (WIDE) {
    // the WIDE variant is the rarely taken one: only the four-chunk lane shapes are built (the smallest that holds the rows)
    const int64_t cap = (int64_t)g0 * 16 * (ca + cb);
    int g = 2;
    while (g < 64 && (int64_t)g * 64 < cap) g <<= 1;
    switch (g) {
      case 2: hipLaunchKernelGGL((k_mhl_fused<2, 4, 0, true, MHLF_WG>), dim3(grid), dim3(MHLF_WG), 0, s, a, nt); break;
      case 4: hipLaunchKernelGGL((k_mhl_fused<4, 4, 0, true, MHLF_WG>), dim3(grid), dim3(MHLF_WG), 0, s, a, nt); break;
      case 8: hipLaunchKernelGGL((k_mhl_fused<8, 4, 0, true, MHLF_WG>), dim3(grid), dim3(MHLF_WG), 0, s, a, nt); break;
      case 16: hipLaunchKernelGGL((k_mhl_fused<16, 4, 0, true, MHLF_WG>), dim3(grid), dim3(MHLF_WG), 0, s, a, nt); break;
      case 32: hipLaunchKernelGGL((k_mhl_fused<32, 4, 0, true, MHLF_WG>), dim3(grid), dim3(MHLF_WG), 0, s, a, nt); break;
      default: hipLaunchKernelGGL((k_mhl_fused<64, 4, 0, true, MHLF_WG>), dim3(grid), dim3(MHLF_WG), 0, s, a, nt); break;
    }
  } else {
#define EPI_LAUNCH1(GG, CCA, CCB)                                                                                                        \
  case GG * 100 + CCA * 10 + CCB:                                                                                                        \
    if (fold) hipLaunchKernelGGL((k_mhl_fused<GG, CCA, CCB, false, MHLF_WG2, true>), dim3(grid), dim3(MHLF_WG2), 0, s, a, nt);         \
    else hipLaunchKernelGGL((k_mhl_fused<GG, CCA, CCB, false, MHLF_WG2, false>), dim3(grid), dim3(MHLF_WG2), 0, s, a, nt);             \
    break;
#define EPI_LAUNCH(GG) EPI_LAUNCH1(GG, 2, 0) EPI_LAUNCH1(GG, 3, 0) EPI_LAUNCH1(GG, 4, 0)
#define EPI_LAUNCH2(GG) EPI_LAUNCH1(GG, 3, 2)
    switch (shape) {
      EPI_LAUNCH(2) EPI_LAUNCH(4) EPI_LAUNCH(8) EPI_LAUNCH(16) EPI_LAUNCH(32) EPI_LAUNCH(64)
      EPI_LAUNCH2(4) EPI_LAUNCH2(8) EPI_LAUNCH2(16) EPI_LAUNCH2(32)
      default: break;
    }
#undef EPI_LAUNCH
#undef EPI_LAUNCH1
#undef EPI_LAUNCH2
  }
}

bool mhl_fused_eligible(epi_batch *b, uint32_t ctx_mask, const RowStats &st) {
  (void)b;
  if (!options().mhl_fused || options().mhl_group_g != 0) return false;    // test hooks: the two-kernel path / its lane shapes
  bool one = false;
  for (uint32_t c : {2u, 6u, 7u}) if (ctx_mask == ((1u << c) | (1u << (c + 8)))) one = true;
  if (!one) return false;                                  // one context, both cases (generateMhlReport's "Zz", "Xx", "Hh")
  return pick_mhl_group(st.max_len) != 0;                  // reads of one block of lanes (up to 64 x 64 bytes)
}

static void fill_args_common(epi_batch *b, MhlFArgs &a) {
  a.tiles = b->tiles.as<Tile>();
  a.cursor = b->misc.as<uint32_t>() + 1;                   // misc layout as in the CX report: [1] cursor, [2] rows, [3] deep tiles
  a.tile_nrow = b->tile_nrow.as<uint32_t>();
  a.tile_base = b->tile_base.as<uint32_t>();
  a.pool_key = b->pool_key.as<uint32_t>();
  a.pool_cov = b->pool_a.as<uint32_t>();
  a.pool_hs = b->pool_d.as<unsigned long long>();
  a.pool_nu = b->pool_e.as<unsigned long long>();
  a.pool_de = b->pool_f.as<unsigned long long>();
  a.pool_cap = (uint32_t)(mhl_pool_rows(b) > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : mhl_pool_rows(b));
  a.slab_cnt = b->d_mhl_cnt_slab;
  a.slab_sum = reinterpret_cast<unsigned long long *>(b->d_mhl_sum_slab);
}

// The fused path.  *done = false: the batch is not eligible -- the caller runs the two-kernel path instead.  With shared
// tiles attached (epi_batch_mhl_set_shared with the fused layout) the report stops after the accumulation (last_kind 5):
// the caller all-reduces the slabs and continues with mhl_fused_finish_shared.
int mhl_fused_report(epi_batch *b, uint32_t ctx_mask, uint32_t H, int hmin, double max_oo, hipStream_t s,
                     int64_t *nrow_out, bool *done) {
  *done = false;
  const int32_t nshared = (int32_t)b->shared_keys.size();
  if (nshared > 0 && !b->mhl_shared_fused) return EPI_OK;  // the slabs attached are the two-kernel path's
  constexpr int T = MHLF_T;
  RowStats st;
  int32_t nt = 0;
  bool nt_hinted = false;                                  // (a remembered tile count is verified at the synchronisation below)
  EPI_TRY(build_tiles(b, s, T, &st, &nt, &nt_hinted));
  if (!mhl_fused_eligible(b, ctx_mask, st)) {
    if (nshared > 0) return fail(EPI_ERR_STATE, "shared tiles were attached for the one-pass lMHL kernel, but this batch needs the two-kernel path");
    return EPI_OK;
  }
  const int gc_wide = pick_mhlf_shape(st.max_len);         // holds every row
  const int gc = pick_mhlf_shape_hist(st, b->n, nt, T);    // the fast variant's: for the bulk of the rows
  uint32_t k = 0;
  for (uint32_t c : {2u, 6u, 7u}) if (ctx_mask == ((1u << c) | (1u << (c + 8)))) k = c;
  b->last_ntiles = nt;
  if (nt == 0) { b->last_kind = nshared > 0 ? 5 : 2; b->last_nrow = 0; *done = true; return EPI_OK; }
  EPI_TRY(b->tile_nrow.ensure((size_t)nt * 4));
  EPI_TRY(b->tile_base.ensure((size_t)nt * 4));
  EPI_TRY(b->tile_out.ensure((size_t)(nt + 1) * 4));
  EPI_TRY(b->heavy_list.ensure((size_t)nt * 4));          // the deep list
  if (!b->mhlf_slot) b->mhlf_slot = T / 8;
  uint32_t slot = b->mhlf_slot > 2u * T ? 2u * T : b->mhlf_slot;
  if (options().mhl_slot >= 0 && options().mhl_slot <= 2 * T) slot = (uint32_t)options().mhl_slot;   // test hook (EPIHIP_MHL_SLOT)
  while (slot && (unsigned long long)nt * slot > 0xC0000000ull) slot >>= 1;
  const size_t headroom = nshared > 0 ? (size_t)nshared * 2 * T : 0;   // shared tiles are emitted later into the same pool
  size_t ovf_base = (size_t)nt * slot;
  for (;;) {
    const size_t ovf = (ovf_base >> 4) > 65536 ? (ovf_base >> 4) : 65536;
    if (mhl_pool_rows(b) >= ovf_base + ovf + headroom) break;
    const int rc = ensure_mhl_pool(b, ovf_base + ovf + headroom);
    if (rc == EPI_OK) break;
    b->pool_cap = 0; b->pool_cap2 = 0;
    if (!slot) return rc;
    slot = 0;
    ovf_base = 0;
  }
  MhlFArgs a{};
  memset(&a, 0, sizeof(a));
  a.xm = b->xm; a.off = b->off; a.len = b->len; a.start = b->start; a.strand = b->strand;
  a.xm_cap = (b->nbytes + 15) / 16 * 16;
  a.k7 = k * 0x01010101u;
  a.lut2 = make_mhlf_lut2(ctx_mask);
  a.hmin = (int32_t)hmin; a.H = H; a.ctx = k;
  {                                                        // decision table over 0 .. longest row; kept while max_oo does not change
    const bool same = b->mhl_keep_len == st.max_len && memcmp(&b->mhl_keep_oo, &max_oo, sizeof(double)) == 0;
    if (!same) {
      EPI_TRY(b->mhl_keep_tab.ensure((size_t)(st.max_len + 1) * 4));
      hipLaunchKernelGGL(k_mhl_keep_table, dim3((unsigned)(st.max_len / 256 + 1)), dim3(256), 0, s, max_oo, st.max_len, b->mhl_keep_tab.as<uint32_t>());
      EPI_HIP(hipGetLastError());
      b->mhl_keep_len = st.max_len;
      b->mhl_keep_oo = max_oo;
    }
    a.keep_tab = b->mhl_keep_tab.as<uint32_t>();
  }
  a.deep_count = b->misc.as<uint32_t>() + 3;
  a.deep_list = b->heavy_list.as<uint32_t>();
  // The fast variant comes with and without the LDS array of folded call counters.  Without it a workgroup needs 4 KB less
  // LDS (five per CU instead of four) and a tile of more than 255 rows folds its u8 counters into a slot of a slab in HBM
  // (MHLF_FOLD_SLOTS of them; a tile that finds none left goes to the WIDE variant with the other deep tiles).  Batches whose
  // tiles average well below 255 rows start there; one that runs out of slots switches the batch over.
  const bool fold = options().mhlf_fold >= 0 ? options().mhlf_fold != 0 : (b->mhlf_prefer_fold || (double)b->n > 150.0 * (double)nt);
  const uint32_t fold_slots = options().mhlf_fold_slots >= 0 && (uint32_t)options().mhlf_fold_slots < MHLF_FOLD_SLOTS
                                  ? (uint32_t)options().mhlf_fold_slots : MHLF_FOLD_SLOTS;   // (test hook: fewer)
  a.max_rows = MHLF_FAST_ROWS;
  a.fold_slab = nullptr; a.fold_cursor = b->misc.as<uint32_t>() + 5; a.fold_slots = 0;
  if (!fold) {
    EPI_TRY(b->mhlf_fold_slab.ensure((size_t)MHLF_FOLD_SLOTS * 2 * T * 4));
    a.fold_slab = b->mhlf_fold_slab.as<uint32_t>();
    a.fold_slots = fold_slots;
  }
  if (options().heavy_rows > 0 && options().heavy_rows < a.max_rows) a.max_rows = options().heavy_rows;   // test hook (EPIHIP_HEAVY_ROWS)
  a.slot_rows = slot;
  a.ovf_base = (uint32_t)ovf_base;
  b->mhl_last_slot = slot;
  b->mhl_last_ovf = (uint32_t)ovf_base;
  b->mhl_ctx_mask = ctx_mask;
  EPI_TRY(check_grid(((int64_t)nt + 7) / 8 * 8, MHLF_WG, "lMHL tile kernel"));
  a.nrows = b->n;
#ifdef EPI_CHECK
  EPI_TRY(b->diag.ensure(256));
  a.dbg = b->diag.as<uint32_t>();
  EPI_HIP(hipMemsetAsync(a.dbg, 0, 32, s));
#endif
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;
  uint32_t host[3] = {0, 0, 0};
  for (int attempt = 0; attempt < 2; attempt++) {
    fill_args_common(b, a);
    if (attempt > 0) {
      EPI_HIP(hipMemsetAsync(cursor, 0, 20, s));           // misc[1..5]
      if (nshared > 0) {                                   // the rerun adds into the slabs again
        EPI_HIP(hipMemsetAsync(a.slab_cnt, 0, (size_t)nshared * MHLF_CNT_PLANES * T * 4, s));
        EPI_HIP(hipMemsetAsync(a.slab_sum, 0, (size_t)nshared * MHLF_SUM_PLANES * T * 8, s));
      }
    }
    a.tile_list = nullptr;
    const unsigned grid = (unsigned)(((nt + 7) / 8) * 8);
    prof_begin("mhl_tiles", s);
    // (the preference was learned for one H: S(min(h, H)) shrinks with H, so a report with a smaller H starts on the fast
    //  variant again, which lists the tiles that still need the wide sums)
    const bool wide_first = b->mhlf_prefer_wide && H >= b->mhlf_prefer_wide_H;
    if (wide_first) launch_mhl_fused<true>(gc_wide, grid, nt, s, a); else launch_mhl_fused<false>(gc, grid, nt, s, a, fold);
    prof_end("mhl_tiles", s);
    EPI_HIP(hipGetLastError());
    EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
    uint32_t host4[6];
    EPI_TRY(read_scalars(b, s, cursor - 1, 24, host4));    // {tile count, overflow rows handed out, total rows, deep tiles, -, fold slots asked for}
    if (nt_hinted && host4[0] != (uint32_t)nt) {
      for (int i = 0; i < 4; i++) b->tile_hint_T[i] = 0;
      return fail(EPI_ERR_STATE, "the rows of this batch changed since an earlier report (tile count %u, was %d)", host4[0], nt);
    }
    if (host4[3] > 0) {
      // tiles the fast variant set aside (too many rows, sums that could wrap u32): the WIDE variant redoes exactly those
      a.tile_list = a.deep_list;
      prof_begin("mhl_deep", s);
      launch_mhl_fused<true>(gc_wide, host4[3], (int)host4[3], s, a);
      prof_end("mhl_deep", s);
      EPI_HIP(hipGetLastError());
      EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
      uint32_t again[2];
      EPI_TRY(read_scalars(b, s, cursor, 8, again));
      host4[1] = again[0]; host4[2] = again[1];
      b->mhlf_prefer_wide = host4[3] > (uint32_t)nt / 2;   // most tiles needed the wide sums: start there next time
      b->mhlf_prefer_wide_H = H;
    }
    if (!fold && host4[5] > fold_slots / 2) b->mhlf_prefer_fold = true;   // many tiles over 255 rows: LDS fold array next time
    // (rows per tile are a property of the immutable batch, not of the report's parameters: this one may stay)
    host[0] = host4[1]; host[1] = host4[2]; host[2] = host4[3];
#ifdef EPI_CHECK
    {
      uint32_t d[8];
      EPI_HIP(hipMemcpy(d, a.dbg, 32, hipMemcpyDeviceToHost));
      if (d[0]) return fail(EPI_ERR_STATE, "fused lMHL index check %u failed: v0=%d v1=%d block=%u thread=%u (n=%lld nt=%d)", d[0],
                            (int)d[1], (int)d[2], d[3], d[4], (long long)b->n, nt);
    }
#endif
    if (ovf_base + host[0] + headroom <= a.pool_cap) break;
    if (attempt == 1) return fail(EPI_ERR_STATE, "row pool overflow after regrow");
    EPI_TRY(ensure_mhl_pool(b, ovf_base + host[0] + (host[0] >> 4) + 1024 + headroom));
  }
  if (host[0] > host[1] / 8 && b->mhlf_slot < 2u * T) b->mhlf_slot *= 2;
  *done = true;
  if (nshared > 0) { b->last_kind = 5; return EPI_OK; }    // caller continues with epi_batch_mhl_finish_shared
  b->last_kind = 2;
  b->last_nrow = host[1];
  *nrow_out = host[1];
  return EPI_OK;
}

// Second half of a sharded lMHL report on the fused path: the slabs have been sum-reduced across ranks.
int mhl_fused_finish_shared(epi_batch *b, hipStream_t s, int64_t *nrow_out) {
  const int32_t nt = b->last_ntiles;
  uint32_t *cursor = b->misc.as<uint32_t>() + 1;
  if (nt > 0 && !b->shared_keys.empty()) {
    MhlFArgs a{};
    memset(&a, 0, sizeof(a));
    fill_args_common(b, a);
    for (uint32_t c : {2u, 6u, 7u}) if (b->mhl_ctx_mask == ((1u << c) | (1u << (c + 8)))) a.ctx = c;
    a.slot_rows = b->mhl_last_slot;
    a.ovf_base = b->mhl_last_ovf;
    hipLaunchKernelGGL(k_mhlf_emit_slab, dim3((unsigned)b->shared_keys.size()), dim3(MHLF_WG), 0, s, a, b->d_shared_owned.as<int32_t>(),
                       b->d_slot_tile.as<int32_t>());
    EPI_HIP(hipGetLastError());
    EPI_TRY(scan_exclusive_u32(a.tile_nrow, b->tile_out.as<uint32_t>(), nt, cursor + 1, b->scan_tmp, s));
    uint32_t ut[2] = {0, 0};
    EPI_TRY(read_scalars(b, s, cursor, 8, ut));
    if ((size_t)a.ovf_base + ut[0] > a.pool_cap) return fail(EPI_ERR_STATE, "row pool overflow in sharded lMHL report");
    b->last_nrow = ut[1];
  } else {
    b->last_nrow = 0;
    if (nt > 0) { uint32_t ut[2] = {0, 0}; EPI_TRY(read_scalars(b, s, cursor, 8, ut)); b->last_nrow = ut[1]; }
  }
  b->last_kind = 2;
  *nrow_out = b->last_nrow;
  return EPI_OK;
}

}  // namespace epi
