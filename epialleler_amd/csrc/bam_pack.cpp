// Host-side producer: BAM file -> sorted, packed SoA templates in pinned host memory.
//
// Replaces the reference's
//   rcpp_check_bam            src/rcpp_check_bam.cpp:19-60   + .checkBam  R/internal.R:75-128
//   rcpp_read_bam_paired      src/rcpp_read_bam.cpp:19-192   (short-read XG/XM alignments)
//   rcpp_read_bam_single      src/rcpp_read_bam.cpp:199-343
//   rcpp_read_bam_mm_single   src/rcpp_read_bam.cpp:364-579  (long-read MM/ML alignments: pack_mm below)
//   .readBam (templid, sort)  R/internal.R:154-199
// without HTSlib: BGZF is a series of gzip members whose compressed size is in the
// BC extra field (SAM spec 4.1), so the blocks are located without inflating (the file is
// mmap-ed) and inflated in parallel by worker threads, window by window (a record or template
// that straddles a window seam is carried over); BAM records have a fixed layout (SAM spec 4.2)
// and are validated before use.  The packed byte per reference position is
// (nt16 << 4) | ctx_to_idx(XM) (src/epialleleR.h:28-35), filler 0xFB (N,'-').
// Output is what the GPU engine consumes: one contiguous byte stream in (rname,start)
// order + offsets + int32 columns, allocated with hipHostMalloc when a HIP device is
// usable (so it can be streamed to HBM with hipMemcpyAsync) and with malloc otherwise.
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <dlfcn.h>
#include <stdio.h>
#include <limits.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <deque>
#include <new>
#include <atomic>
#include <numeric>
#include <string>
#include <thread>
#include <chrono>
#include <vector>
#include "common.hpp"

using namespace epi;

namespace {

// Large host buffers (the inflated window, the packed bytes of a thread, the record index): 2 MiB-aligned and advised
// as huge pages -- a first touch then faults 2 MiB at a time instead of 4 KiB (hundreds of thousands of faults per file
// from 16 threads otherwise).  Released with free().
inline void *big_malloc(size_t bytes) {
  constexpr size_t HUGE = (size_t)2 << 20;
  if (bytes >= 4 * HUGE && !epi::options().no_hugepage) {
    const size_t r = (bytes + HUGE - 1) & ~(HUGE - 1);
    void *p = aligned_alloc(HUGE, r);
    if (p) { (void)madvise(p, r, MADV_HUGEPAGE); return p; }
  }
  return malloc(bytes);
}
template <class T> struct BigAlloc {
  using value_type = T;
  BigAlloc() = default;
  template <class U> BigAlloc(const BigAlloc<U> &) {}
  T *allocate(size_t n) { void *p = big_malloc(n * sizeof(T)); if (!p) throw std::bad_alloc(); return static_cast<T *>(p); }
  void deallocate(T *p, size_t) { free(p); }
  template <class U> bool operator==(const BigAlloc<U> &) const { return true; }
  template <class U> bool operator!=(const BigAlloc<U> &) const { return false; }
};

struct Block {
  size_t cpos, clen; size_t upos, ulen;                     // compressed payload / uncompressed placement
  // what the inflating thread found when it walked the block's bytes as a chain of BAM records starting at the block's
  // first byte: their number, and whether the chain ends exactly at the block's end.  (HTSlib never lets a record
  // straddle two blocks unless it is larger than one, so this is the true chain nearly always -- the index uses it
  // only where the true chain does arrive at the block's first byte.)
  uint32_t spec_n = 0;
  bool spec_ok = false;
};

inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// The compressed file, memory-mapped (read into a buffer where mmap is not possible): nothing of it is copied.
struct FileView {
  const uint8_t *p = nullptr;
  size_t n = 0;
  bool mapped = false;
  std::vector<uint8_t> own;
  ~FileView() { if (mapped && p) munmap(const_cast<uint8_t *>(p), n); }
};

int open_file(const char *path, FileView &v) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(EPI_ERR_ARG, "Unable to open BAM file for reading");   // src/rcpp_read_bam.cpp:34
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 0) { close(fd); return fail(EPI_ERR_ARG, "Unable to read BAM file"); }
  v.n = (size_t)st.st_size;
  if (v.n == 0) { close(fd); return EPI_OK; }
  void *m = mmap(nullptr, v.n, PROT_READ, MAP_PRIVATE, fd, 0);
  if (m != MAP_FAILED) { v.p = static_cast<const uint8_t *>(m); v.mapped = true; close(fd); return EPI_OK; }
  v.own.resize(v.n);
  size_t got = 0;
  while (got < v.n) {
    const ssize_t k = read(fd, v.own.data() + got, v.n - got);
    if (k <= 0) break;
    got += (size_t)k;
  }
  close(fd);
  if (got != v.n) return fail(EPI_ERR_ARG, "Unable to read BAM file");
  v.p = v.own.data();
  return EPI_OK;
}

// Locate the BGZF blocks (no inflation).
int bgzf_scan(const uint8_t *in, size_t n, std::vector<Block> &blocks) {
  size_t p = 0;
  while (p + 18 <= n) {
    const uint8_t *h = in + p;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail(EPI_ERR_ARG, "not a BGZF/BAM file");
    const unsigned xlen = rd16(h + 10);
    size_t q = p + 12;
    const size_t xend = p + 12 + xlen;
    int bsize = -1;
    while (q + 4 <= xend && xend <= n) {
      const unsigned slen = rd16(in + q + 2);
      if (in[q] == 'B' && in[q + 1] == 'C' && slen == 2 && q + 6 <= xend) bsize = rd16(in + q + 4);
      q += 4 + slen;
    }
    if (bsize < 0 || p + (size_t)bsize + 1 > n) return fail(EPI_ERR_ARG, "truncated BGZF block");
    const size_t blen = (size_t)bsize + 1;
    if (blen < (xend - p) + 8) return fail(EPI_ERR_ARG, "corrupt BGZF block");
    Block b;
    b.cpos = xend;
    b.clen = blen - (xend - p) - 8;
    b.ulen = rd32(in + p + blen - 4);
    b.upos = 0;
    if (b.ulen > 65536) return fail(EPI_ERR_ARG, "corrupt BGZF block");
    blocks.push_back(b);
    p += blen;
  }
  if (p != n) return fail(EPI_ERR_ARG, "truncated BGZF block");
  return EPI_OK;
}

// libdeflate (its shared library ships with the image; 2-3x zlib's inflate rate) is used when it can be loaded at run
// time -- no headers needed for its three-call ABI -- and zlib otherwise.  Both produce the same bytes or an error.
struct FastInflate {
  void *(*alloc)() = nullptr;
  int (*run)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
  void (*release)(void *) = nullptr;
  FastInflate() {
    if (epi::options().no_libdeflate) return;
    void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc = reinterpret_cast<void *(*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
    run = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(dlsym(h, "libdeflate_deflate_decompress"));
    release = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_decompressor"));
    if (!alloc || !run || !release) { alloc = nullptr; run = nullptr; release = nullptr; }
  }
  bool ok() const { return run != nullptr; }
};
const FastInflate &fast_inflate() { static const FastInflate f; return f; }

// Inflate blocks [b0, b1) to out + their upos, in parallel.
int bgzf_inflate_range(const uint8_t *in, std::vector<Block> &blocks, size_t b0, size_t b1, uint8_t *out, int nthreads) {
  std::atomic<size_t> next(b0);
  std::atomic<int> bad(0);
  auto walk = [&](Block &b) {                                // (the bytes are still in this core's cache)
    const uint8_t *q = out + b.upos;
    size_t p = 0;
    uint32_t n = 0;
    while (p + 4 <= b.ulen) {
      const size_t bs = rd32(q + p);
      if (p + 4 + bs > b.ulen) break;
      p += 4 + bs;
      n++;
    }
    b.spec_n = n;
    b.spec_ok = p == b.ulen;
  };
  auto work = [&]() {
    const FastInflate &fi = fast_inflate();
    if (fi.ok()) {
      void *d = fi.alloc();
      if (d) {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= b1) break;
          Block &b = blocks[i];
          b.spec_n = 0; b.spec_ok = false;
          if (b.ulen == 0) continue;
          size_t got = 0;
          if (fi.run(d, in + b.cpos, b.clen, out + b.upos, b.ulen, &got) != 0 || got != b.ulen) bad = 1;
          else walk(b);
        }
        fi.release(d);
        return;
      }
    }
    z_stream zs;                                             // one stream per thread, reset per block (an init / end pair per
    memset(&zs, 0, sizeof(zs));                              // block allocates and frees its state and window every 64 KiB)
    if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= b1) break;
      Block &b = blocks[i];
      b.spec_n = 0; b.spec_ok = false;
      if (b.ulen == 0) continue;
      if (inflateReset(&zs) != Z_OK) { bad = 1; continue; }
      zs.next_in = const_cast<Bytef *>(in + b.cpos);
      zs.avail_in = (uInt)b.clen;
      zs.next_out = out + b.upos;
      zs.avail_out = (uInt)b.ulen;
      const int rc = inflate(&zs, Z_FINISH);
      if (rc != Z_STREAM_END || zs.avail_out != 0) bad = 1;
      else walk(b);
    }
    inflateEnd(&zs);
  };
  int nt = nthreads > 0 ? nthreads : 1;
  if ((size_t)nt > b1 - b0) nt = b1 > b0 ? (int)(b1 - b0) : 1;
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
  if (bad) return fail(EPI_ERR_ARG, "corrupt BGZF block");
  return EPI_OK;
}

struct Rec {                 // one BAM alignment record, pointing into the inflated stream
  int32_t tid, pos, mtid, mpos, isize, l_seq;
  uint32_t n_cigar;
  uint16_t flag;
  uint8_t mapq;
  const char *qname;
  const uint8_t *cigar, *seq, *qual, *aux, *end;
};

// bam_aux_get for Z-typed tags: pointer to the first character, or NULL
const char *aux_z(const Rec &r, char a, char b, bool *present) {
  const uint8_t *p = r.aux;
  *present = false;
  while (p + 3 <= r.end) {
    const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2];
    const bool hit = t0 == a && t1 == b;
    p += 3;
    size_t adv = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': adv = 1; break;
      case 's': case 'S': adv = 2; break;
      case 'i': case 'I': case 'f': adv = 4; break;
      case 'Z': case 'H': {
        const uint8_t *e = (const uint8_t *)memchr(p, 0, (size_t)(r.end - p));
        if (!e) return nullptr;
        if (hit) { *present = true; return (const char *)p; }
        p = e + 1;
        continue;
      }
      case 'B': {
        if (p + 5 > r.end) return nullptr;
        const char sub = (char)p[0];
        const uint32_t cnt = rd32(p + 1);
        const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        adv = 5 + (size_t)cnt * es;
        break;
      }
      default: return nullptr;
    }
    if (hit) { *present = true; return nullptr; }   // present but not a string
    p += adv;
  }
  return nullptr;
}

bool has_tag(const Rec &r, char a, char b) { bool pr; (void)aux_z(r, a, b, &pr); return pr; }

inline uint8_t ctx_idx(char c) { return (uint8_t)ctx_to_idx((unsigned char)c); }
inline uint8_t seqi_shifted(const uint8_t *s, uint32_t i) { return (uint8_t)((s[i >> 1] << ((i & 1) << 2)) & 0xF0); }   // epialleleR.h:32


// bam_aux_get for B-typed (array) tags: element type, count and a pointer to the first element; false if absent
bool aux_b(const Rec &r, char a, char b, char *sub, uint32_t *count, const uint8_t **data) {
  const uint8_t *p = r.aux;
  while (p + 3 <= r.end) {
    const bool hit = (char)p[0] == a && (char)p[1] == b;
    const char ty = (char)p[2];
    p += 3;
    size_t adv = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': adv = 1; break;
      case 's': case 'S': adv = 2; break;
      case 'i': case 'I': case 'f': adv = 4; break;
      case 'Z': case 'H': {
        const uint8_t *e = (const uint8_t *)memchr(p, 0, (size_t)(r.end - p));
        if (!e || hit) return false;
        p = e + 1;
        continue;
      }
      case 'B': {
        if (p + 5 > r.end) return false;
        const char st = (char)p[0];
        const uint32_t cnt = rd32(p + 1);
        const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
        if (p + 5 + (size_t)cnt * es > r.end) return false;
        if (hit) { *sub = st; *count = cnt; *data = p + 5; return true; }
        adv = 5 + (size_t)cnt * es;
        break;
      }
      default: return false;
    }
    if (hit) return false;
    p += adv;
  }
  return false;
}

// ---- long-read (MM/ML) records -------------------------------------------------------------------------------
// The reference leaves the MM/ML tags to HTSlib (bam_parse_basemod / bam_next_basemod; Rhtslib, version not pinned in
// DESCRIPTION).  HTSlib is not part of the reference tree, so this restates the published rules of the SAM tags
// specification (SAMtags.pdf section 1.7 "Base modifications"):
//   MM:Z:  ([ACGTUN][-+]([a-z]+|[0-9]+)[.?]?(,[0-9]+)*;)*   one entry per (canonical base, strand, modification codes);
//          each number = how many bases of that canonical type to skip before the next modified one, counted along
//          the read AS SEQUENCED: for a reverse-strand alignment from the end of SEQ, on the complemented base;
//          base N counts every base; several one-letter codes in one entry share the positions.
//   ML:B:C one probability (0..255) per listed position and code, in MM order, codes of an entry interleaved per
//          position.  Without ML the probability is unknown (-1, as HTSlib reports it).
// A ChEBI number n is carried as code -n (HTSlib's convention, which rcpp_read_bam.cpp:474 relies on for 27551).
struct ModHit { int32_t pos; int32_t code; int32_t strand; int32_t qual; };

inline int nt16_of(char c) {
  switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8; case 'N': return 15; default: return -1; }
}
inline int nt16_complement(int c) { return c == 1 ? 8 : c == 8 ? 1 : c == 2 ? 4 : c == 4 ? 2 : c; }
inline int seqi(const uint8_t *s, uint32_t i) { return (s[i >> 1] >> ((~i & 1) << 2)) & 0xF; }

void parse_basemods(const Rec &r, const char *mm, bool has_ml, const uint8_t *ml, uint32_t n_ml, std::vector<ModHit> &hits) {
  hits.clear();
  const bool rev = (r.flag & 16) != 0;
  const int32_t L = r.l_seq;
  uint32_t ml_idx = 0;
  const char *p = mm;
  while (*p) {
    const int base = nt16_of(*p);
    if (base < 0) return;                                        // malformed: HTSlib gives up on the tag
    p++;
    if (*p != '+' && *p != '-') return;
    const int32_t strand = *p == '-' ? 1 : 0;
    p++;
    int32_t codes[16];
    int ncodes = 0;
    if (*p >= '0' && *p <= '9') {                                // one ChEBI number
      long v = 0;
      while (*p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; }
      codes[ncodes++] = (int32_t)-v;
    } else {
      while (*p >= 'a' && *p <= 'z') { if (ncodes < 16) codes[ncodes++] = (int32_t)*p; p++; }
    }
    if (ncodes == 0) return;
    if (*p == '.' || *p == '?') p++;                             // implicit / explicit mode: only listed bases matter here
    const int target = rev ? nt16_complement(base) : base;
    int32_t i = rev ? L - 1 : 0;                                 // next base of SEQ to look at, in sequencing order
    const int32_t step = rev ? -1 : 1;
    while (*p == ',') {
      p++;
      long skip = 0;
      if (!(*p >= '0' && *p <= '9')) return;
      while (*p >= '0' && *p <= '9') { skip = skip * 10 + (*p - '0'); p++; }
      // advance to the (skip+1)-th base of the canonical type
      int32_t found = -1;
      while (i >= 0 && i < L) {
        const bool match = target == 15 || seqi(r.seq, (uint32_t)i) == target;
        const int32_t at = i;
        i += step;
        if (match) { if (skip == 0) { found = at; break; } skip--; }
      }
      if (found < 0) return;                                     // the tag points beyond the sequence
      for (int k = 0; k < ncodes; k++) {
        ModHit h;
        h.pos = found; h.code = codes[k]; h.strand = strand;
        h.qual = has_ml ? (ml_idx < n_ml ? (int32_t)ml[ml_idx] : -1) : -1;
        ml_idx++;
        hits.push_back(h);
      }
    }
    if (*p != ';') return;
    p++;
  }
}

// cytosine context of query base i from the base and its two neighbours, keyed like the reference's 512-entry tables
// (src/epialleleR.h:43-116) by the low three bits of the IUPAC letters: A=1 C=3 T=4 N=6 G=7, anything else gives '.'
inline bool tri_ok(int c) { return c == 1 || c == 3 || c == 4 || c == 6 || c == 7; }
inline char ctx_forward(int b0, int b1, int b2) {                // C at i: CG -> z, CHG -> x, CHH -> h
  if (b0 != 3 || !tri_ok(b1) || !tri_ok(b2)) return '.';
  return b1 == 7 ? 'z' : b2 == 7 ? 'x' : 'h';
}
inline char ctx_reverse(int b0, int b1, int b2) {                // G at i with (i-2, i-1, i): CG -> z, CHG -> x, CHH -> h
  if (b2 != 7 || !tri_ok(b0) || !tri_ok(b1)) return '.';
  return b1 == 3 ? 'z' : b0 == 3 ? 'x' : 'h';
}

struct Packed {
  std::vector<int32_t> rname, strand, start;
  std::vector<int64_t> off;          // template t owns bytes [off[t], off[t+1])
  std::vector<uint8_t, BigAlloc<uint8_t>> bytes;
};

// (nt16 << 4) | ctx_idx(XM) of every query base of a record (src/epialleleR.h:28,32): two bases per byte of SEQ
inline void packed_bytes(const Rec &r, const char *xm, std::vector<uint8_t> &pb) {
  const size_t n = (size_t)r.l_seq;
  pb.resize(n + 2);
  uint8_t *__restrict__ o = pb.data();
  const uint8_t *__restrict__ s = r.seq;
  const unsigned char *__restrict__ x = reinterpret_cast<const unsigned char *>(xm);
  for (size_t i = 0; i + 1 < n; i += 2) {
    const uint8_t b = s[i >> 1];
    o[i] = (uint8_t)((b & 0xF0) | (((x[i] + 2u) >> 2) & 15u));
    o[i + 1] = (uint8_t)(((b << 4) & 0xF0) | (((x[i + 1] + 2u) >> 2) & 15u));
  }
  if (n & 1) o[n - 1] = (uint8_t)((s[(n - 1) >> 1] & 0xF0) | (((x[n - 1] + 2u) >> 2) & 15u));
}

// walk the CIGAR of one record into the template buffers; returns the reference position after the last op
template <class F>
int apply_cigar(const Rec &r, uint32_t dest0, F &&on_match, uint32_t *dest_end) {
  uint32_t qpos = 0, dpos = dest0;
  for (uint32_t i = 0; i < r.n_cigar; i++) {
    const uint32_t c = rd32(r.cigar + 4 * i), op = c & 0xF, len = c >> 4;
    switch (op) {
      case 0: case 7: case 8: on_match(qpos, dpos, len); qpos += len; dpos += len; break;    // M = X
      case 1: case 4: qpos += len; break;                                                     // I S
      case 2: case 3: dpos += len; break;                                                     // D N
      case 5: case 6: case 9: break;                                                          // H P B
      default: return fail(EPI_ERR_ARG, "Unknown CIGAR operation for BAM entry %s", r.qname);
    }
  }
  *dest_end = dpos;
  return EPI_OK;
}

}  // namespace

extern "C" {

void epi_templates_free(epi_templates *t) {
  if (!t) return;
#ifdef EPI_HOST_ONLY
  free(t->xm);                                             // (sanitizer build: no HIP runtime, the bytes came from malloc)
#else
  if (t->xm) { if (t->pinned) (void)hipHostFree(t->xm); else free(t->xm); }
#endif
  free(t->off); free(t->rname); free(t->strand); free(t->start);
  if (t->target_names) { for (int32_t i = 0; i < t->n_targets; i++) free(t->target_names[i]); free(t->target_names); }
  memset(t, 0, sizeof(*t));
}

static int preprocess_impl(const char *path, const epi_bam_options *opt_in, epi_templates *out);

int epi_preprocess_bam(const char *path, const epi_bam_options *opt_in, epi_templates *out) {
  if (!path || !out) return fail(EPI_ERR_ARG, "epi_preprocess_bam: NULL argument");
  memset(out, 0, sizeof(*out));
  int rc;
  try {                                                     // nothing may unwind through the C boundary
    rc = preprocess_impl(path, opt_in, out);
  } catch (const std::bad_alloc &) {
    rc = fail(EPI_ERR_NOMEM, "epi_preprocess_bam: out of host memory");
  } catch (...) {
    rc = fail(EPI_ERR_ARG, "epi_preprocess_bam: unexpected failure while reading %s", path);
  }
  if (rc != EPI_OK) epi_templates_free(out);
  return rc;
}

}  // extern "C"

// One record of the inflated stream -> Rec, with the structural checks HTSlib's bam_read1 makes (sizes consistent
// with block_size, NUL-terminated name); false: the record is not well-formed.
static bool parse_record(const uint8_t *b, uint32_t bs, Rec *r) {
  if (bs < 32) return false;
  r->tid = (int32_t)rd32(b); r->pos = (int32_t)rd32(b + 4);
  const uint32_t l_qname = b[8];
  r->mapq = b[9];
  r->n_cigar = rd16(b + 12); r->flag = rd16(b + 14);
  r->l_seq = (int32_t)rd32(b + 16); r->mtid = (int32_t)rd32(b + 20); r->mpos = (int32_t)rd32(b + 24); r->isize = (int32_t)rd32(b + 28);
  if (l_qname < 1 || r->l_seq < 0) return false;
  const uint64_t need = 32ull + l_qname + 4ull * r->n_cigar + ((uint64_t)r->l_seq + 1) / 2 + (uint64_t)r->l_seq;
  if (need > bs) return false;
  r->qname = (const char *)b + 32;
  if (b[32 + l_qname - 1] != 0) return false;
  r->cigar = b + 32 + l_qname;
  r->seq = r->cigar + 4 * (size_t)r->n_cigar;
  r->qual = r->seq + ((size_t)r->l_seq + 1) / 2;
  r->aux = r->qual + (size_t)r->l_seq;
  r->end = b + bs;
  return true;
}

// query bases the CIGAR consumes (M I S = X)
static uint64_t cigar_qlen(const Rec &r) {
  uint64_t q = 0;
  for (uint32_t i = 0; i < r.n_cigar; i++) {
    const uint32_t c = rd32(r.cigar + 4 * i), op = c & 0xF;
    if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) q += c >> 4;
  }
  return q;
}

static int preprocess_impl(const char *path, const epi_bam_options *opt_in, epi_templates *out) {
  epi_bam_options opt;
  if (opt_in) opt = *opt_in;
  else { memset(&opt, 0, sizeof(opt)); opt.skip_secondary = opt.skip_qcfail = opt.skip_supplementary = 1; opt.paired = -1; opt.nthreads = 1; opt.min_prob = -1; opt.highest_prob = 1; }
  if (opt.trim5 < 0 || opt.trim3 < 0) return fail(EPI_ERR_ARG, "trim must be non-negative");

  const bool timing = epi::options().bam_timing != 0;
  auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tm0 = tnow();
  auto lap = [&](const char *what) { if (timing) { const double t = tnow(); fprintf(stderr, "[bam] %-10s %.3f s\n", what, t - tm0); tm0 = t; } };
  FileView file;
  EPI_TRY(open_file(path, file));
  std::vector<Block> blocks;
  EPI_TRY(bgzf_scan(file.p, file.n, blocks));
  // The output bytes go to page-locked memory (the upload DMAs straight from it).  Pinning ~100 MB takes tens of
  // milliseconds, so it starts now, next to the inflate / pack work, sized by an estimate (a base is at least 2.5 bytes
  // of record: half a byte of SEQ, QUAL, XM); a template set that turns out larger is allocated at the end instead.
  struct PinAhead {
    std::thread th;
    void *p = nullptr;
    size_t cap = 0;
    bool ok = false;
    ~PinAhead() { release(); }
    void wait() { if (th.joinable()) th.join(); }
    void release() {
      wait();
#ifndef EPI_HOST_ONLY
      if (ok && p) (void)hipHostFree(p);
#endif
      p = nullptr; ok = false;
    }
  } pin;
#ifndef EPI_HOST_ONLY
  {
    size_t total = 0;
    for (const Block &b : blocks) total += b.ulen;
    if (total >= ((size_t)8 << 20)) {
      // template bytes per inflated byte, sampled from the file's first records (sum of l_seq over sum of record sizes: a
      // PE150 Bismark / DRAGEN file gives ~0.35, where the worst-case bound is 0.4): the buffer pinned ahead is sized by it
      // plus 15 %; 0.5 when the sample cannot be read
      double ratio = 0.5;
      {
        std::vector<Block> head;
        size_t up = 0;
        for (size_t i = 0; i < blocks.size() && head.size() < 4; i++) { Block b = blocks[i]; b.upos = up; up += b.ulen; head.push_back(b); }
        std::vector<uint8_t> buf(up + 8);
        if (up >= 16 && bgzf_inflate_range(file.p, head, 0, head.size(), buf.data(), 1) == EPI_OK && memcmp(buf.data(), "BAM\1", 4) == 0) {
          size_t p = 8 + (size_t)rd32(buf.data() + 4);
          uint64_t nref = p + 4 <= up ? rd32(buf.data() + p) : 0;
          p += 4;
          while (nref > 0 && p + 4 <= up) { p += 8 + (size_t)rd32(buf.data() + p); nref--; }
          uint64_t bytes = 0, bases = 0;
          while (nref == 0 && p + 36 <= up) {
            const size_t bs = rd32(buf.data() + p);
            if (bs < 32 || p + 4 + bs > up) break;
            bytes += 4 + bs; bases += rd32(buf.data() + p + 4 + 16);
            p += 4 + bs;
          }
          if (bytes >= 4096 && bases > 0 && bases < bytes) ratio = (double)bases / (double)bytes;
        }
      }
      pin.cap = ((size_t)((double)total * ratio * 1.15) + ((size_t)1 << 20) + 15) / 16 * 16;
      int cur_dev = -1;                                      // the caller's device, not device 0: every rank of a multi-GPU job pins under its own GPU
      if (hipGetDevice(&cur_dev) != hipSuccess) { (void)hipGetLastError(); cur_dev = -1; }
      pin.th = std::thread([&pin, cur_dev]() {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && (cur_dev < 0 || hipSetDevice(cur_dev) == hipSuccess) &&
            hipHostMalloc(&pin.p, pin.cap, hipHostMallocDefault) == hipSuccess) pin.ok = true;
        else (void)hipGetLastError();
      });
    }
  }
#endif

  // The file is processed in windows of inflated data (the reference streams records through HTSlib): inflate a run
  // of BGZF blocks, index the complete records, pack the complete templates, carry the rest (a record cut by the
  // window, or the records of a template whose mate is still to come) over to the next window.
  const size_t window = opt.window_kib > 0 ? (size_t)opt.window_kib * 1024 : (size_t)256 << 20;
  // the window of inflated bytes: grown without clearing it (std::vector::resize would zero-fill hundreds of megabytes on
  // one thread before every inflate)
  struct RawBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    ~RawBuf() { free(p); }
    uint8_t *data() { return p; }
    size_t size() const { return n; }
    uint8_t &operator[](size_t i) { return p[i]; }
    void resize(size_t m) {                                 // keeps the first n bytes (the carry)
      if (m > cap) {
        uint8_t *q = static_cast<uint8_t *>(big_malloc(m));
        if (!q) throw std::bad_alloc();
        if (n) memcpy(q, p, n);
        free(p);
        p = q; cap = m;
      }
      n = m;
    }
    void release() { free(p); p = nullptr; n = cap = 0; }
  } buf;
  size_t carry = 0, bi = 0, hdr_end = 0;
  bool header_done = false, checked = false, paired = false, tMM = false;
  std::vector<std::string> names;
  struct RecBuf {                                           // the window's records (not value-initialised: 80 bytes x millions)
    Rec *p = nullptr;
    size_t n = 0, cap = 0;
    ~RecBuf() { free(p); }
    size_t size() const { return n; }
    void clear() { n = 0; }
    Rec &operator[](size_t i) { return p[i]; }
    const Rec &operator[](size_t i) const { return p[i]; }
    const Rec *begin() const { return p; }
    const Rec *end() const { return p + n; }
    void resize_uninit(size_t m) {
      if (m > cap) {
        free(p);
        p = static_cast<Rec *>(big_malloc((m + m / 8 + 16) * sizeof(Rec)));
        if (!p) { cap = 0; n = 0; throw std::bad_alloc(); }
        cap = m + m / 8 + 16;
      }
      n = m;
    }
  } recs;
  std::vector<size_t> roff;                                 // offsets of the window's records in buf
  size_t nrecs_total = 0;
  uint16_t skip_flags = 4;
  if (opt.skip_secondary) skip_flags |= 256;
  if (opt.skip_qcfail) skip_flags |= 512;
  if (opt.skip_duplicates) skip_flags |= 1024;
  if (opt.skip_supplementary) skip_flags |= 2048;
  const int trim5 = opt.trim5, trim3 = opt.trim3;
  Packed P;                                                 // the small columns of all templates (the bytes stay in `segs`)

  // ---- .readBam: skip flags (R/internal.R:173-177, above) and the packers ----
  // a record that enters a template must be self-consistent: the CIGAR consumes exactly the stored bases, XM covers
  // them, the reference id exists (HTSlib rejects such records while reading; without the checks they index past
  // the record)
  auto use_record = [&](const Rec &r, const char *xm) -> int {
    if (r.tid < 0 || (size_t)r.tid >= names.size()) return fail(EPI_ERR_ARG, "corrupt BAM record %s: reference id out of range", r.qname);
    if (cigar_qlen(r) != (uint64_t)r.l_seq) return fail(EPI_ERR_ARG, "corrupt BAM record %s: CIGAR does not match the sequence length", r.qname);
    if (xm && strlen(xm) < (size_t)r.l_seq) return fail(EPI_ERR_ARG, "corrupt BAM record %s: XM tag shorter than the sequence", r.qname);
    return EPI_OK;
  };
  // Each packer turns records [r_lo, r_hi) into templates appended to P; ranges are packed by several threads and
  // concatenated in order (a paired-end range never starts inside a template).
  auto pack_mm = [&](size_t r_lo, size_t r_hi, Packed &P) -> int {
    // ---- rcpp_read_bam_mm_single (src/rcpp_read_bam.cpp:364-579) ----
    static const char nt16_str[] = "=ACMGRSVTWYHKDBN";
    std::vector<char> seq, xm[2];
    std::vector<uint8_t> rs[2];
    std::vector<ModHit> hits;
    for (size_t ri = r_lo; ri < r_hi; ri++) {
      const Rec &r = recs[ri];
      if ((r.flag & skip_flags) || (int)r.mapq < opt.min_mapq) continue;                        // :423-424
      EPI_TRY(use_record(r, nullptr));
      const int record_strand = (r.flag & 16) ? 1 : 0;                                          // :426
      const int32_t qw = r.l_seq < 0 ? -r.l_seq : r.l_seq;                                      // :436
      uint32_t width = 0;                                                                       // bam_cigar2rlen, :437
      for (uint32_t i = 0; i < r.n_cigar; i++) {
        const uint32_t c = rd32(r.cigar + 4 * i), op = c & 0xF;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) width += c >> 4;
      }
      rs[0].assign(width, 0xFB); rs[1].assign(width, 0xFB);                                     // :453-454
      seq.assign((size_t)qw + 4, 'N');                                                          // :457-461 NN + SEQ + NN
      for (int32_t i = 0; i < qw; i++) seq[(size_t)i + 2] = nt16_str[seqi(r.seq, (uint32_t)i)];
      xm[0].resize((size_t)qw); xm[1].resize((size_t)qw);
      for (int32_t i = 0; i < qw; i++) {                                                        // :464-467
        xm[0][(size_t)i] = ctx_forward(seq[(size_t)i + 2] & 7, seq[(size_t)i + 3] & 7, seq[(size_t)i + 4] & 7);
        xm[1][(size_t)i] = ctx_reverse(seq[(size_t)i] & 7, seq[(size_t)i + 1] & 7, seq[(size_t)i + 2] & 7);
      }
      bool strand_has_mods[2] = {false, false};
      bool pmm = false;
      const char *mm = aux_z(r, 'M', 'M', &pmm);
      if (!mm) mm = aux_z(r, 'M', 'm', &pmm);
      if (mm) {
        char sub = 0; uint32_t n_ml = 0; const uint8_t *ml = nullptr;
        bool has_ml = aux_b(r, 'M', 'L', &sub, &n_ml, &ml) || aux_b(r, 'M', 'l', &sub, &n_ml, &ml);
        if (has_ml && sub != 'C' && sub != 'c') has_ml = false;
        parse_basemods(r, mm, has_ml, ml, n_ml, hits);
        std::stable_sort(hits.begin(), hits.end(), [](const ModHit &x, const ModHit &y) { return x.pos < y.pos; });
        for (size_t a0 = 0; a0 < hits.size();) {                                                // one query position at a time, :469
          size_t a1 = a0;
          int ismeth[2] = {0, 0}, meth_prob[2] = {-2, -2}, max_other[2] = {-2, -2};             // :470-472
          for (; a1 < hits.size() && hits[a1].pos == hits[a0].pos; a1++) {
            const ModHit &h = hits[a1];
            if (h.code == 'm' || h.code == -27551) { ismeth[h.strand] = 1; meth_prob[h.strand] = h.qual; }   // :474-476
            else if (max_other[h.strand] < h.qual) max_other[h.strand] = h.qual;                             // :477-479
          }
          const int32_t mod_pos = hits[a0].pos;
          for (int sidx = 0; sidx < 2; sidx++) {                                                // :481-490
            const int cs = record_strand > sidx ? record_strand - sidx : sidx - record_strand;
            if (ismeth[sidx] && meth_prob[sidx] >= opt.min_prob && (!opt.highest_prob || meth_prob[sidx] > max_other[sidx]) &&
                xm[cs][(size_t)mod_pos] > 'A') {
              xm[cs][(size_t)mod_pos] &= (char)0xDF;
              strand_has_mods[cs] = true;
            }
          }
          a0 = a1;
        }
      }
      uint32_t dest_end = 0;
      EPI_TRY(apply_cigar(r, 0, [&](uint32_t qpos, uint32_t dpos, uint32_t len) {              // :494-531
        for (uint32_t j = 0; j < len; j++)
          if ((int)r.qual[qpos + j] >= opt.min_baseq) {
            const uint8_t hi = seqi_shifted(r.seq, qpos + j);
            rs[0][dpos + j] = (uint8_t)(hi | ctx_idx(xm[0][qpos + j]));
            rs[1][dpos + j] = (uint8_t)(hi | ctx_idx(xm[1][qpos + j]));
          }
      }, &dest_end));
      strand_has_mods[record_strand] = true;                                                    // :534
      for (int sidx = 0; sidx < 2; sidx++) {
        if (!strand_has_mods[sidx]) continue;
        P.rname.push_back(r.tid + 1);                                                           // :537-541
        P.strand.push_back(sidx + 1);
        P.start.push_back(r.pos + trim5 + 1);
        const int keep = (int)dest_end - (trim5 + trim3);
        if (keep > 0) P.bytes.insert(P.bytes.end(), rs[sidx].begin() + trim5, rs[sidx].begin() + trim5 + keep);
        P.off.push_back((int64_t)P.bytes.size());
      }
    }
    return EPI_OK;
  };
  auto pack_pe = [&](size_t r_lo, size_t r_hi, Packed &P) -> int {
    const uint16_t skip_flags_pe = skip_flags | 8;
    const uint8_t q0 = (uint8_t)(opt.min_baseq - (opt.min_baseq > 0 ? 1 : 0));   // src/rcpp_read_bam.cpp:30,57
    std::vector<uint8_t> tq(8192, q0), ts(8192, 0xFB), pb;
    const char *tname = nullptr;
    int t_rname = 0, t_start = 0, t_strand = 0, t_width = 0;
    auto push_template = [&]() {                                                   // :61-69
      P.rname.push_back(t_rname + 1);
      P.strand.push_back(t_strand);
      P.start.push_back(t_start + trim5 + 1);
      const int keep = t_width - (trim5 + trim3);
      if (keep > 0) P.bytes.insert(P.bytes.end(), ts.begin() + trim5, ts.begin() + trim5 + keep);
      P.off.push_back((int64_t)P.bytes.size());
      std::fill(tq.begin(), tq.begin() + t_width, q0);
      std::fill(ts.begin(), ts.begin() + t_width, (uint8_t)0xFB);
    };
    for (size_t ri = r_lo; ri < r_hi; ri++) {
      const Rec &r = recs[ri];
      if ((r.flag & skip_flags_pe) || !(r.flag & 0x2) || (int)r.mapq < opt.min_mapq) continue;   // :76-78
      bool pg, pm;
      const char *xg = aux_z(r, 'X', 'G', &pg), *xm = aux_z(r, 'X', 'M', &pm);
      if (!pg || !pm || !xg || !xm) continue;                                                   // :80-82
      EPI_TRY(use_record(r, xm));
      if (!tname || strcmp(tname, r.qname) != 0) {                                              // :85
        if (t_strand != 0) push_template();
        tname = r.qname;
        t_rname = r.tid;
        t_start = r.pos < r.mpos ? r.pos : r.mpos;                                              // :92-93
        if (r.isize == INT32_MIN) return fail(EPI_ERR_ARG, "corrupt BAM record %s: template length", r.qname);
        t_width = r.isize < 0 ? -r.isize : r.isize;                                             // :94
        t_strand = 2 - (xg[0] == 'C' ? 1 : 0);                                                  // :95
        if ((size_t)t_width > tq.size()) { tq.resize((size_t)t_width, q0); ts.resize((size_t)t_width, 0xFB); }
      }
      uint32_t dest_end = 0;
      if (r.pos < t_start) return fail(EPI_ERR_ARG, "corrupt BAM record %s: starts before its template", r.qname);
      const uint32_t dest0 = (uint32_t)(r.pos - t_start);                                       // :118
      packed_bytes(r, xm, pb);                                                                  // (nt16 << 4) | ctx_idx per query base
      EPI_TRY(apply_cigar(r, dest0, [&](uint32_t qpos, uint32_t dpos, uint32_t len) {
        if ((size_t)dpos + len > tq.size()) { tq.resize((size_t)dpos + len, q0); ts.resize((size_t)dpos + len, 0xFB); }
        const uint8_t *__restrict__ ql = r.qual + qpos, *__restrict__ pq = pb.data() + qpos;
        uint8_t *__restrict__ tqd = tq.data() + dpos, *__restrict__ tsd = ts.data() + dpos;
        for (uint32_t j = 0; j < len; j++) {                                                    // :127 strictly higher quality wins
          const bool w = ql[j] > tqd[j];                                                        // (selects, not branches: the loop vectorises)
          tqd[j] = w ? ql[j] : tqd[j];
          tsd[j] = w ? pq[j] : tsd[j];
        }
      }, &dest_end));
      if (dest_end > 0x7FFFFFFFu) return fail(EPI_ERR_ARG, "corrupt BAM record %s: template too wide", r.qname);
      if (t_width < (int)dest_end) t_width = (int)dest_end;                                     // :151
      if ((size_t)t_width > tq.size()) { tq.resize((size_t)t_width, q0); ts.resize((size_t)t_width, 0xFB); }   // (a CIGAR ending in D / N)
    }
    if (t_strand != 0) push_template();                                                         // :155 (see below for "none")
    return EPI_OK;
  };
  auto pack_se = [&](size_t r_lo, size_t r_hi, Packed &P) -> int {
    std::vector<uint8_t> buf, pb;
    for (size_t ri = r_lo; ri < r_hi; ri++) {
      const Rec &r = recs[ri];
      if ((r.flag & skip_flags) || (int)r.mapq < opt.min_mapq) continue;                        // :240-241
      bool pg, pm;
      const char *xg = aux_z(r, 'X', 'G', &pg), *xm = aux_z(r, 'X', 'M', &pm);
      if (!pg || !pm || !xg || !xm) continue;
      EPI_TRY(use_record(r, xm));
      uint32_t width = 0;                                                                       // bam_cigar2rlen, :255
      for (uint32_t i = 0; i < r.n_cigar; i++) {
        const uint32_t c = rd32(r.cigar + 4 * i), op = c & 0xF;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) width += c >> 4;
      }
      buf.assign(width, 0xFB);                                                                  // :265
      uint32_t dest_end = 0;
      packed_bytes(r, xm, pb);
      EPI_TRY(apply_cigar(r, 0, [&](uint32_t qpos, uint32_t dpos, uint32_t len) {
        const uint8_t *__restrict__ ql = r.qual + qpos, *__restrict__ pq = pb.data() + qpos;
        uint8_t *__restrict__ bd = buf.data() + dpos;
        for (uint32_t j = 0; j < len; j++) bd[j] = (int)ql[j] >= opt.min_baseq ? pq[j] : bd[j]; // :278
      }, &dest_end));
      P.rname.push_back(r.tid + 1);                                                             // :303-306
      P.strand.push_back(xg[0] == 'C' ? 1 : 2);
      P.start.push_back(r.pos + trim5 + 1);
      const int keep = (int)dest_end - (trim5 + trim3);
      if (keep > 0) P.bytes.insert(P.bytes.end(), buf.begin() + trim5, buf.begin() + trim5 + keep);
      P.off.push_back((int64_t)P.bytes.size());
    }
    return EPI_OK;
  };

  std::deque<Packed> segs;                                  // what the packing threads produced, kept until the ordered copy
  std::vector<const uint8_t *> src;                         // per template: its bytes inside a segment ...
  std::vector<int32_t> len;                                 // ... and how many
  // packs records [0, r_end) of the current window with K threads and appends the templates to P
  auto pack_window = [&](size_t r_end) -> int {
    size_t K = opt.nthreads > 1 ? (size_t)(opt.nthreads > 16 ? 16 : opt.nthreads) : 1;
    if (r_end < 1024) K = 1;
    std::vector<size_t> cut(K + 1);
    for (size_t k = 0; k <= K; k++) {
      size_t c = r_end * k / K;
      if (!tMM && paired)                                   // a template's records are neighbours with one QNAME
        while (c > 0 && c < r_end && strcmp(recs[c].qname, recs[c - 1].qname) == 0) c++;
      cut[k] = c;
    }
    std::vector<Packed> part(K);
    std::vector<int> rcs(K, EPI_OK);
    std::vector<std::string> msgs(K);
    auto run = [&](size_t k) {
      try {
        part[k].off.push_back(0);
        {                                                   // (one allocation instead of a doubling series per column)
          size_t bases = 0;
          for (size_t i = cut[k]; i < cut[k + 1]; i++) bases += (size_t)recs[i].l_seq;
          const size_t nrec = cut[k + 1] - cut[k];
          part[k].bytes.reserve(bases + bases / 8 + 4096);
          part[k].off.reserve(nrec + 2); part[k].rname.reserve(nrec + 1); part[k].strand.reserve(nrec + 1); part[k].start.reserve(nrec + 1);
        }
        rcs[k] = tMM ? pack_mm(cut[k], cut[k + 1], part[k]) : paired ? pack_pe(cut[k], cut[k + 1], part[k]) : pack_se(cut[k], cut[k + 1], part[k]);
        if (rcs[k] != EPI_OK) msgs[k] = epi_last_error();   // the message is thread-local
      } catch (const std::bad_alloc &) {
        rcs[k] = EPI_ERR_NOMEM; msgs[k] = "epi_preprocess_bam: out of host memory";
      } catch (...) {
        rcs[k] = EPI_ERR_ARG; msgs[k] = "epi_preprocess_bam: unexpected failure while packing templates";
      }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < K; k++) th.emplace_back(run, k);
    run(0);
    for (auto &t : th) t.join();
    for (size_t k = 0; k < K; k++)
      if (rcs[k] != EPI_OK) return fail(rcs[k], "%s", msgs[k].c_str());
    // the packed bytes stay where the threads wrote them (a segment per thread and window): only the small columns are
    // concatenated, and the final ordered copy reads the segments directly
    for (size_t k = 0; k < K; k++) {
      segs.push_back(std::move(part[k]));
      const Packed &q = segs.back();
      P.rname.insert(P.rname.end(), q.rname.begin(), q.rname.end());
      P.strand.insert(P.strand.end(), q.strand.begin(), q.strand.end());
      P.start.insert(P.start.end(), q.start.begin(), q.start.end());
      for (size_t i = 0; i + 1 < q.off.size(); i++) {
        src.push_back(q.bytes.data() + q.off[i]);
        len.push_back((int32_t)(q.off[i + 1] - q.off[i]));
      }
    }
    return EPI_OK;
  };

  // ---- the windows ----
  for (bool final = blocks.empty(); ;) {
    size_t b1 = bi, add = 0;
    const size_t win_b0 = bi;                                // first block inflated into this window
    while (b1 < blocks.size() && (b1 == bi || add + blocks[b1].ulen <= window)) { blocks[b1].upos = carry + add; add += blocks[b1].ulen; b1++; }
    final = b1 == blocks.size();
    buf.resize(carry + add);
    EPI_TRY(bgzf_inflate_range(file.p, blocks, bi, b1, buf.data(), opt.nthreads));
    bi = b1;
    size_t p = hdr_end;
    if (!header_done) {                                      // BAM header: magic, text, reference names
      bool complete = false;
      do {
        if (buf.size() < 12) break;
        if (memcmp(buf.data(), "BAM\1", 4) != 0) return fail(EPI_ERR_ARG, "Unable to read BAM header");
        size_t q = 8 + (size_t)rd32(buf.data() + 4);
        if (q + 4 > buf.size()) break;
        const uint32_t n_ref = rd32(buf.data() + q);
        q += 4;
        names.clear();
        bool ok = true;
        for (uint32_t i = 0; i < n_ref && ok; i++) {
          if (q + 4 > buf.size()) { ok = false; break; }
          const uint32_t l = rd32(buf.data() + q);
          if (q + 4 + (size_t)l + 4 > buf.size()) { ok = false; break; }
          if (l == 0 || buf[q + 4 + l - 1] != 0) return fail(EPI_ERR_ARG, "Unable to read BAM header");
          names.emplace_back((const char *)buf.data() + q + 4);
          q += 4 + (size_t)l + 4;
        }
        if (!ok) break;
        hdr_end = q;
        complete = true;
      } while (0);
      if (!complete) {
        if (final) return fail(EPI_ERR_ARG, "Unable to read BAM header");
        carry = buf.size();                                  // the header is longer than a window: read on
        continue;
      }
      header_done = true;
      p = hdr_end;
    }
    // Index of the complete records of the window.  The chain of block sizes is followed from the first record; wherever
    // it arrives exactly at the first byte of a BGZF block whose own walk (done by the thread that inflated it) ended
    // exactly at the block's end, the block's records are taken as a whole -- they are enumerated and parsed by all
    // threads below -- so the serial part is one step per block, not per record.
    struct Span { size_t p; uint32_t n; };                  // n records starting at byte p (a block, or a single record)
    std::vector<Span> spans;
    size_t nrec = 0;
    {
      size_t kb = win_b0;
      for (;;) {
        while (kb < bi && blocks[kb].upos < p) kb++;
        if (kb < bi && blocks[kb].upos == p && blocks[kb].spec_ok && blocks[kb].spec_n > 0) {
          spans.push_back({p, blocks[kb].spec_n});
          nrec += blocks[kb].spec_n;
          p += blocks[kb].ulen;
          continue;
        }
        if (p + 4 > buf.size()) break;
        const uint32_t bs = rd32(buf.data() + p);
        if (p + 4 + (size_t)bs > buf.size()) break;          // cut by the window (or by the end of the file)
        spans.push_back({p, 1u});
        nrec++;
        p += 4 + (size_t)bs;
      }
    }
    recs.resize_uninit(nrec);
    roff.resize(nrec);
    {
      std::vector<size_t> base(spans.size() + 1, 0);
      for (size_t i = 0; i < spans.size(); i++) base[i + 1] = base[i] + spans[i].n;
      size_t K = opt.nthreads > 1 ? (size_t)(opt.nthreads > 16 ? 16 : opt.nthreads) : 1;
      if (nrec < 4096) K = 1;
      std::atomic<size_t> next(0);
      std::atomic<int> bad(0);
      auto parse = [&]() {
        for (;;) {
          const size_t i0 = next.fetch_add(16);              // a few spans at a time
          if (i0 >= spans.size()) break;
          const size_t i1 = i0 + 16 < spans.size() ? i0 + 16 : spans.size();
          for (size_t i = i0; i < i1; i++) {
            size_t q = spans[i].p;
            for (size_t j = base[i]; j < base[i + 1]; j++) {
              const uint8_t *rp = buf.data() + q;
              const uint32_t bs = rd32(rp);
              roff[j] = q;
              if (!parse_record(rp + 4, bs, &recs[j])) { bad = 1; return; }
              q += 4 + (size_t)bs;
            }
          }
        }
      };
      std::vector<std::thread> th;
      for (size_t k = 1; k < K; k++) th.emplace_back(parse);
      parse();
      for (auto &t : th) t.join();
      if (bad) return fail(EPI_ERR_ARG, "corrupt BAM record");
    }
    if (final && p != buf.size()) return fail(EPI_ERR_ARG, "truncated BAM record");
    if (!checked) {
      if (recs.size() < 1024 && !final) { carry = buf.size(); continue; }   // .checkBam looks at the first 1024 records
      lap("index");
      // ---- .checkBam over the first 1024 records (src/rcpp_check_bam.cpp:40-50, R/internal.R:82-120) ----
      size_t nrecs = 0, npaired = 0, ntempls = 0;
      bool tXG = false, tXM = false, tYD = false, tZS = false;
      const char *prevq = nullptr;
      for (const Rec &r : recs) {
        if (nrecs >= 1024) break;
        nrecs++;
        if (r.flag & 0x2) npaired++;
        tXG |= has_tag(r, 'X', 'G'); tXM |= has_tag(r, 'X', 'M'); tYD |= has_tag(r, 'Y', 'D'); tZS |= has_tag(r, 'Z', 'S');
        tMM |= has_tag(r, 'M', 'M') || has_tag(r, 'M', 'm');
        if (prevq && strcmp(prevq, r.qname) == 0) ntempls++;
        prevq = r.qname;
      }
      paired = npaired * 2 > nrecs;
      const bool sorted = ntempls > 0 && (ntempls >= nrecs / 2 || ntempls >= npaired / 2);
      if (nrecs == 0) return fail(EPI_ERR_ARG, "Empty file provided! Exiting");
      if (!tXG && tYD) return fail(EPI_ERR_ARG, "No XG tags found (though YD tags are there)! BWA-meth alignment? If so, make methylation calls using epialleleR::callMethylation. Exiting");
      if (!tXG && tZS) return fail(EPI_ERR_ARG, "No XG tags found (though ZS tags are there)! BSMAP alignment? If so, make methylation calls using epialleleR::callMethylation. Exiting");
      if (!tXM && tXG) return fail(EPI_ERR_ARG, "No XM tags found! Was methylation called successfully? If not, make methylation calls using epialleleR::callMethylation. Exiting");
      if (!tMM && !(tXG && tXM)) return fail(EPI_ERR_ARG, "No known methylation tags found! Exiting");
      if (paired && !sorted) return fail(EPI_ERR_ARG, "BAM file seems to be paired-end but not sorted by name! Please sort using 'samtools sort -n -o out.bam in.bam'. Exiting");
      if (opt.paired >= 0 && (opt.paired != 0) != paired) return fail(EPI_ERR_ARG, "Expected endness is different from detected! Exiting");
      checked = true;
    }
    // paired-end: the records of the window's last QNAME wait for the next window (their mate may be in it)
    size_t r_end = recs.size();
    if (!final && paired && !tMM && r_end > 0) {
      const char *lastq = recs[r_end - 1].qname;
      while (r_end > 0 && strcmp(recs[r_end - 1].qname, lastq) == 0) r_end--;
      if (r_end == 0) { carry = buf.size(); continue; }      // one template fills the window: read on
    }
    EPI_TRY(pack_window(r_end));
    nrecs_total += r_end;
    const size_t keep_from = r_end < recs.size() ? roff[r_end] : p;
    carry = buf.size() - keep_from;
    if (carry) memmove(buf.data(), buf.data() + keep_from, carry);
    hdr_end = 0;                                             // (the header is gone from the buffer)
    recs.clear();
    if (final) break;
  }
  if (!tMM && paired && P.rname.empty()) {                  // the reference pushes its (never opened) template all the same, :155
    P.rname.push_back(1); P.strand.push_back(0); P.start.push_back(trim5 + 1); src.push_back(nullptr); len.push_back(0);
  }
  buf.release();

  lap("pack");
  // ---- templid := 0..N-1 ; setorder(rname, start) -- stable (R/internal.R:193-195) ----
  const size_t n = P.rname.size();
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  {
    auto less = [&](uint32_t a, uint32_t b) {
      if (P.rname[a] != P.rname[b]) return P.rname[a] < P.rname[b];
      return P.start[a] < P.start[b];
    };
    // stable: ranges sorted by the threads, then merged pairwise (std::inplace_merge keeps equal keys in order)
    size_t K = 1;
    while (K * 2 <= (size_t)(opt.nthreads > 16 ? 16 : opt.nthreads) && n / (K * 2) >= 65536) K *= 2;
    std::vector<size_t> cut(K + 1);
    for (size_t k = 0; k <= K; k++) cut[k] = n * k / K;
    {
      std::vector<std::thread> th;
      for (size_t k = 1; k < K; k++) th.emplace_back([&, k]() { std::stable_sort(order.begin() + cut[k], order.begin() + cut[k + 1], less); });
      std::stable_sort(order.begin() + cut[0], order.begin() + cut[1], less);
      for (auto &t : th) t.join();
    }
    for (size_t w = 1; w < K; w *= 2) {                      // merge runs of w ranges
      std::vector<std::thread> th;
      for (size_t k = 0; k + w < K; k += 2 * w) {
        const size_t lo = cut[k], mid = cut[k + w], hi = cut[k + 2 * w < K ? k + 2 * w : K];
        th.emplace_back([&, lo, mid, hi]() { std::inplace_merge(order.begin() + lo, order.begin() + mid, order.begin() + hi, less); });
      }
      for (auto &t : th) t.join();
    }
  }
  size_t nbytes = 0;
  for (size_t i = 0; i < n; i++) nbytes += (size_t)len[i];
  size_t cap = (nbytes + 15) / 16 * 16 + 64;
  void *xmp = nullptr;
#ifdef EPI_HOST_ONLY
  xmp = malloc(cap); out->pinned = 0;                      // host-only sanitizer build (`make asan`)
#else
  pin.wait();
  // the buffer pinned while the file was being read -- unless it turned out too small, or more than a quarter larger
  // than needed (page-locked memory stays pinned for the life of the templates: then the exact size is allocated)
  if (pin.ok && pin.cap >= cap && pin.cap <= cap + cap / 4 + ((size_t)4 << 20)) {
    xmp = pin.p; cap = pin.cap; out->pinned = 1;
    pin.p = nullptr; pin.ok = false;
  } else {
    pin.release();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipHostMalloc(&xmp, cap, hipHostMallocDefault) == hipSuccess) out->pinned = 1;
    else { (void)hipGetLastError(); xmp = malloc(cap); out->pinned = 0; }
  }
#endif
  out->xm = (uint8_t *)xmp;
  out->off = (int64_t *)malloc((n + 1) * sizeof(int64_t));
  out->rname = (int32_t *)malloc((n + 1) * sizeof(int32_t));
  out->strand = (int32_t *)malloc((n + 1) * sizeof(int32_t));
  out->start = (int32_t *)malloc((n + 1) * sizeof(int32_t));
  out->target_names = (char **)calloc(names.size() + 1, sizeof(char *));
  if (!out->xm || !out->off || !out->rname || !out->strand || !out->start || !out->target_names) {
    epi_templates_free(out);
    return fail(EPI_ERR_NOMEM, "epi_preprocess_bam: out of host memory");
  }
  int64_t w = 0;
  for (size_t i = 0; i < n; i++) { out->off[i] = w; w += len[order[i]]; }
  out->off[n] = w;
  {                                                        // the ordered copy, by ranges of output rows
    size_t K = opt.nthreads > 1 ? (size_t)(opt.nthreads > 16 ? 16 : opt.nthreads) : 1;
    if (n < 4096) K = 1;
    auto run = [&](size_t k) {
      for (size_t i = n * k / K; i < n * (k + 1) / K; i++) {
        const uint32_t t = order[i];
        if (len[t]) memcpy(out->xm + out->off[i], src[t], (size_t)len[t]);
        out->rname[i] = P.rname[t]; out->strand[i] = P.strand[t]; out->start[i] = P.start[t];
      }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < K; k++) th.emplace_back(run, k);
    run(0);
    for (auto &t : th) t.join();
  }
  // 0xFB padding up to the 16-byte boundary the kernels may read to, plus a little (a buffer pinned ahead can be much
  // larger than the templates: its tail stays untouched and is not part of xm_capacity)
  const size_t padded = ((size_t)w + 15) / 16 * 16 + 64;
  memset(out->xm + w, 0xFB, padded - (size_t)w);
  cap = padded;
  lap("sort+copy");
  out->n = (int64_t)n;
  out->nbytes = w;
  out->xm_capacity = (int64_t)cap;
  out->nrecs = (int64_t)nrecs_total;
  out->paired = paired ? 1 : 0;
  out->n_targets = (int32_t)names.size();
  for (size_t i = 0; i < names.size(); i++) out->target_names[i] = strdup(names[i].c_str());
  return EPI_OK;
}
