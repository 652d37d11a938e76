// Host-side producer: BAM file -> sorted, packed SoA templates in pinned host memory.
//
// Replaces, for short-read XG/XM alignments, the reference's
//   rcpp_check_bam            src/rcpp_check_bam.cpp:19-60   + .checkBam  R/internal.R:75-128
//   rcpp_read_bam_paired      src/rcpp_read_bam.cpp:19-192
//   rcpp_read_bam_single      src/rcpp_read_bam.cpp:199-343
//   .readBam (templid, sort)  R/internal.R:154-199
// without HTSlib: BGZF is a series of gzip members whose compressed size is in the
// BC extra field (SAM spec 4.1), so the blocks are located without inflating and
// inflated in parallel by worker threads straight into one buffer; BAM records have a
// fixed layout (SAM spec 4.2).  The packed byte per reference position is
// (nt16 << 4) | ctx_to_idx(XM) (src/epialleleR.h:28-35), filler 0xFB (N,'-').
// Output is what the GPU engine consumes: one contiguous byte stream in (rname,start)
// order + offsets + int32 columns, allocated with hipHostMalloc when a HIP device is
// usable (so it can be streamed to HBM with hipMemcpyAsync) and with malloc otherwise.
// Long-read MM/ML alignments (rcpp_read_bam_mm_single) are not handled yet.
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <numeric>
#include <string>
#include <thread>
#include <vector>
#include "common.hpp"

using namespace epi;

namespace {

struct Block { size_t cpos, clen; size_t upos, ulen; };   // compressed payload / uncompressed placement

inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int read_file(const char *path, std::vector<uint8_t> &buf) {
  FILE *f = fopen(path, "rb");
  if (!f) return fail(EPI_ERR_ARG, "Unable to open BAM file for reading");   // src/rcpp_read_bam.cpp:34
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize(sz > 0 ? (size_t)sz : 0);
  size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, f) : 0;
  fclose(f);
  if (got != buf.size()) return fail(EPI_ERR_ARG, "Unable to read BAM file");
  return EPI_OK;
}

// Locate the BGZF blocks (no inflation), then inflate them in parallel.
int bgzf_inflate(const std::vector<uint8_t> &in, int nthreads, std::vector<uint8_t> &out) {
  std::vector<Block> blocks;
  size_t p = 0, total = 0;
  while (p + 18 <= in.size()) {
    const uint8_t *h = in.data() + p;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail(EPI_ERR_ARG, "not a BGZF/BAM file");
    const unsigned xlen = rd16(h + 10);
    size_t q = p + 12, xend = p + 12 + xlen;
    int bsize = -1;
    while (q + 4 <= xend && xend <= in.size()) {
      const unsigned slen = rd16(in.data() + q + 2);
      if (in[q] == 'B' && in[q + 1] == 'C' && slen == 2) bsize = rd16(in.data() + q + 4);
      q += 4 + slen;
    }
    if (bsize < 0 || p + (size_t)bsize + 1 > in.size()) return fail(EPI_ERR_ARG, "truncated BGZF block");
    const size_t blen = (size_t)bsize + 1;
    Block b;
    b.cpos = xend;
    b.clen = blen - (xend - p) - 8;
    b.ulen = rd32(in.data() + p + blen - 4);
    b.upos = total;
    total += b.ulen;
    blocks.push_back(b);
    p += blen;
  }
  out.resize(total);
  std::atomic<size_t> next(0);
  std::atomic<int> bad(0);
  auto work = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= blocks.size()) break;
      const Block &b = blocks[i];
      if (b.ulen == 0) continue;
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; continue; }
      zs.next_in = const_cast<Bytef *>(in.data() + b.cpos);
      zs.avail_in = (uInt)b.clen;
      zs.next_out = out.data() + b.upos;
      zs.avail_out = (uInt)b.ulen;
      const int rc = inflate(&zs, Z_FINISH);
      if (rc != Z_STREAM_END || zs.avail_out != 0) bad = 1;
      inflateEnd(&zs);
    }
  };
  int nt = nthreads > 0 ? nthreads : 1;
  if ((size_t)nt > blocks.size()) nt = blocks.empty() ? 1 : (int)blocks.size();
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

struct Packed {
  std::vector<int32_t> rname, strand, start;
  std::vector<int64_t> off;          // template t owns bytes [off[t], off[t+1])
  std::vector<uint8_t> bytes;
};

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
  if (t->xm) { if (t->pinned) (void)hipHostFree(t->xm); else free(t->xm); }
  free(t->off); free(t->rname); free(t->strand); free(t->start);
  if (t->target_names) { for (int32_t i = 0; i < t->n_targets; i++) free(t->target_names[i]); free(t->target_names); }
  memset(t, 0, sizeof(*t));
}

int epi_preprocess_bam(const char *path, const epi_bam_options *opt_in, epi_templates *out) {
  if (!path || !out) return fail(EPI_ERR_ARG, "epi_preprocess_bam: NULL argument");
  memset(out, 0, sizeof(*out));
  epi_bam_options opt;
  if (opt_in) opt = *opt_in;
  else { memset(&opt, 0, sizeof(opt)); opt.skip_secondary = opt.skip_qcfail = opt.skip_supplementary = 1; opt.paired = -1; opt.nthreads = 1; }
  if (opt.trim5 < 0 || opt.trim3 < 0) return fail(EPI_ERR_ARG, "trim must be non-negative");

  std::vector<uint8_t> file, bam;
  EPI_TRY(read_file(path, file));
  EPI_TRY(bgzf_inflate(file, opt.nthreads, bam));
  file.clear(); file.shrink_to_fit();
  if (bam.size() < 12 || memcmp(bam.data(), "BAM\1", 4) != 0) return fail(EPI_ERR_ARG, "Unable to read BAM header");
  size_t p = 8 + (size_t)rd32(bam.data() + 4);
  if (p + 4 > bam.size()) return fail(EPI_ERR_ARG, "Unable to read BAM header");
  const uint32_t n_ref = rd32(bam.data() + p);
  p += 4;
  std::vector<std::string> names;
  for (uint32_t i = 0; i < n_ref; i++) {
    if (p + 4 > bam.size()) return fail(EPI_ERR_ARG, "Unable to read BAM header");
    const uint32_t l = rd32(bam.data() + p);
    if (p + 4 + l + 4 > bam.size()) return fail(EPI_ERR_ARG, "Unable to read BAM header");
    names.emplace_back((const char *)bam.data() + p + 4);
    p += 4 + l + 4;
  }
  // record index
  std::vector<Rec> recs;
  while (p + 4 <= bam.size()) {
    const uint32_t bs = rd32(bam.data() + p);
    if (bs < 32 || p + 4 + bs > bam.size()) return fail(EPI_ERR_ARG, "truncated BAM record");
    const uint8_t *b = bam.data() + p + 4;
    Rec r;
    r.tid = (int32_t)rd32(b); r.pos = (int32_t)rd32(b + 4);
    const uint32_t l_qname = b[8];
    r.mapq = b[9];
    r.n_cigar = rd16(b + 12); r.flag = rd16(b + 14);
    r.l_seq = (int32_t)rd32(b + 16); r.mtid = (int32_t)rd32(b + 20); r.mpos = (int32_t)rd32(b + 24); r.isize = (int32_t)rd32(b + 28);
    r.qname = (const char *)b + 32;
    r.cigar = b + 32 + l_qname;
    r.seq = r.cigar + 4 * (size_t)r.n_cigar;
    r.qual = r.seq + ((size_t)r.l_seq + 1) / 2;
    r.aux = r.qual + (size_t)r.l_seq;
    r.end = b + bs;
    if (r.aux > r.end) return fail(EPI_ERR_ARG, "corrupt BAM record");
    recs.push_back(r);
    p += 4 + (size_t)bs;
  }

  // ---- .checkBam over the first 1024 records (src/rcpp_check_bam.cpp:40-50, R/internal.R:82-120) ----
  size_t nrecs = 0, npaired = 0, ntempls = 0;
  bool tXG = false, tXM = false, tYD = false, tZS = false, tMM = false;
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
  const bool paired = npaired * 2 > nrecs;
  const bool sorted = ntempls > 0 && (ntempls >= nrecs / 2 || ntempls >= npaired / 2);
  if (nrecs == 0) return fail(EPI_ERR_ARG, "Empty file provided! Exiting");
  if (!tXG && tYD) return fail(EPI_ERR_ARG, "No XG tags found (though YD tags are there)! BWA-meth alignment? If so, make methylation calls using epialleleR::callMethylation. Exiting");
  if (!tXG && tZS) return fail(EPI_ERR_ARG, "No XG tags found (though ZS tags are there)! BSMAP alignment? If so, make methylation calls using epialleleR::callMethylation. Exiting");
  if (!tXM && tXG) return fail(EPI_ERR_ARG, "No XM tags found! Was methylation called successfully? If not, make methylation calls using epialleleR::callMethylation. Exiting");
  if (tMM) return fail(EPI_ERR_ARG, "long-read MM/ML alignment detected: not supported by this producer yet");
  if (!(tXG && tXM)) return fail(EPI_ERR_ARG, "No known methylation tags found! Exiting");
  if (paired && !sorted) return fail(EPI_ERR_ARG, "BAM file seems to be paired-end but not sorted by name! Please sort using 'samtools sort -n -o out.bam in.bam'. Exiting");
  if (opt.paired >= 0 && (opt.paired != 0) != paired) return fail(EPI_ERR_ARG, "Expected endness is different from detected! Exiting");

  // ---- .readBam: skip flags (R/internal.R:173-177) and the packers ----
  uint16_t skip_flags = 4;
  if (opt.skip_secondary) skip_flags |= 256;
  if (opt.skip_qcfail) skip_flags |= 512;
  if (opt.skip_duplicates) skip_flags |= 1024;
  if (opt.skip_supplementary) skip_flags |= 2048;
  Packed P;
  P.off.push_back(0);
  const int trim5 = opt.trim5, trim3 = opt.trim3;
  if (paired) {
    skip_flags |= 8;
    const uint8_t q0 = (uint8_t)(opt.min_baseq - (opt.min_baseq > 0 ? 1 : 0));   // src/rcpp_read_bam.cpp:30,57
    std::vector<uint8_t> tq(8192, q0), ts(8192, 0xFB);
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
    for (const Rec &r : recs) {
      if ((r.flag & skip_flags) || !(r.flag & 0x2) || (int)r.mapq < opt.min_mapq) continue;   // :76-78
      bool pg, pm;
      const char *xg = aux_z(r, 'X', 'G', &pg), *xm = aux_z(r, 'X', 'M', &pm);
      if (!pg || !pm || !xg || !xm) continue;                                                   // :80-82
      if (!tname || strcmp(tname, r.qname) != 0) {                                              // :85
        if (t_strand != 0) push_template();
        tname = r.qname;
        t_rname = r.tid;
        t_start = r.pos < r.mpos ? r.pos : r.mpos;                                              // :92-93
        t_width = r.isize < 0 ? -r.isize : r.isize;                                             // :94
        t_strand = 2 - (xg[0] == 'C' ? 1 : 0);                                                  // :95
        if ((size_t)t_width > tq.size()) { tq.resize((size_t)t_width, q0); ts.resize((size_t)t_width, 0xFB); }
      }
      uint32_t dest_end = 0;
      const uint32_t dest0 = (uint32_t)(r.pos - t_start);                                       // :118
      EPI_TRY(apply_cigar(r, dest0, [&](uint32_t qpos, uint32_t dpos, uint32_t len) {
        if ((size_t)dpos + len > tq.size()) { tq.resize((size_t)dpos + len, q0); ts.resize((size_t)dpos + len, 0xFB); }
        for (uint32_t j = 0; j < len; j++) {
          if (r.qual[qpos + j] > tq[dpos + j]) {                                                // :127 strictly higher quality wins
            tq[dpos + j] = r.qual[qpos + j];
            ts[dpos + j] = (uint8_t)(seqi_shifted(r.seq, qpos + j) | ctx_idx(xm[qpos + j]));
          }
        }
      }, &dest_end));
      if (t_width < (int)dest_end) t_width = (int)dest_end;                                     // :151
    }
    push_template();                                                                            // :155
  } else {
    std::vector<uint8_t> buf;
    for (const Rec &r : recs) {
      if ((r.flag & skip_flags) || (int)r.mapq < opt.min_mapq) continue;                        // :240-241
      bool pg, pm;
      const char *xg = aux_z(r, 'X', 'G', &pg), *xm = aux_z(r, 'X', 'M', &pm);
      if (!pg || !pm || !xg || !xm) continue;
      uint32_t width = 0;                                                                       // bam_cigar2rlen, :255
      for (uint32_t i = 0; i < r.n_cigar; i++) {
        const uint32_t c = rd32(r.cigar + 4 * i), op = c & 0xF;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) width += c >> 4;
      }
      buf.assign(width, 0xFB);                                                                  // :265
      uint32_t dest_end = 0;
      EPI_TRY(apply_cigar(r, 0, [&](uint32_t qpos, uint32_t dpos, uint32_t len) {
        for (uint32_t j = 0; j < len; j++)
          if ((int)r.qual[qpos + j] >= opt.min_baseq)                                           // :278
            buf[dpos + j] = (uint8_t)(seqi_shifted(r.seq, qpos + j) | ctx_idx(xm[qpos + j]));
      }, &dest_end));
      P.rname.push_back(r.tid + 1);                                                             // :303-306
      P.strand.push_back(xg[0] == 'C' ? 1 : 2);
      P.start.push_back(r.pos + trim5 + 1);
      const int keep = (int)dest_end - (trim5 + trim3);
      if (keep > 0) P.bytes.insert(P.bytes.end(), buf.begin() + trim5, buf.begin() + trim5 + keep);
      P.off.push_back((int64_t)P.bytes.size());
    }
  }

  // ---- templid := 0..N-1 ; setorder(rname, start) -- stable (R/internal.R:193-195) ----
  const size_t n = P.rname.size();
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    if (P.rname[a] != P.rname[b]) return P.rname[a] < P.rname[b];
    return P.start[a] < P.start[b];
  });
  const size_t nbytes = P.bytes.size(), cap = (nbytes + 15) / 16 * 16 + 64;
  void *xmp = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipHostMalloc(&xmp, cap, hipHostMallocDefault) == hipSuccess) out->pinned = 1;
  else { (void)hipGetLastError(); xmp = malloc(cap); out->pinned = 0; }
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
  for (size_t i = 0; i < n; i++) {
    const uint32_t t = order[i];
    const int64_t len = P.off[t + 1] - P.off[t];
    out->off[i] = w;
    if (len) memcpy(out->xm + w, P.bytes.data() + P.off[t], (size_t)len);
    w += len;
    out->rname[i] = P.rname[t]; out->strand[i] = P.strand[t]; out->start[i] = P.start[t];
  }
  out->off[n] = w;
  memset(out->xm + w, 0xFB, cap - (size_t)w);
  out->n = (int64_t)n;
  out->nbytes = w;
  out->xm_capacity = (int64_t)cap;
  out->nrecs = (int64_t)recs.size();
  out->paired = paired ? 1 : 0;
  out->n_targets = (int32_t)names.size();
  for (size_t i = 0; i < names.size(); i++) out->target_names[i] = strdup(names[i].c_str());
  return EPI_OK;
}

}  // extern "C"
