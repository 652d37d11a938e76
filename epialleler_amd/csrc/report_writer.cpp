// .writeReport (R/internal.R:274-287): the report table as a tab-separated file with a header line, optionally
// gzip-compressed -- what data.table::fwrite(report, quote=FALSE, sep="\t", col.names=TRUE, compress=...) writes:
// integers in decimal, factors as their labels, doubles with up to 15 significant digits, NA (and NaN) as an empty
// field.  Host-only code: rows are formatted in chunks by `nthreads` threads and written in order; with gzip every
// chunk becomes one member of the (multi-member, RFC 1952) gzip file.
#include "common.hpp"
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

using epi::fail;

inline char *put_i32(char *p, int32_t v) {
  uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
  if (v < 0) *p++ = '-';
  char tmp[12];
  int n = 0;
  do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  while (n) *p++ = tmp[--n];
  return p;
}

inline char *put_f64(char *p, double x) {
  if (isnan(x)) return p;                                  // NA / NaN: na = ""
  if (isinf(x)) { if (x < 0) *p++ = '-'; memcpy(p, "Inf", 3); return p + 3; }
  if (fabs(x) < 1e15 && x == (double)(int64_t)x) {         // integral values print without a fraction (as %.15g does);
                                                           // the magnitude test comes first: the cast is undefined at 2^63 and beyond
    int64_t v = (int64_t)x;
    uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    if (v < 0) *p++ = '-';
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
  }
  return p + snprintf(p, 32, "%.15g", x);
}

struct Job {                                               // rows [lo, hi) of the table -> `out` (text, or one gzip member)
  int64_t lo = 0, hi = 0;
  std::vector<char> out;
  int rc = 0;
  bool done = false;
};

size_t field_bound(const epi_report_column &c) {
  if (c.kind == EPI_COL_F64) return 32;
  if (c.kind == EPI_COL_FACTOR) {
    size_t m = 12;
    for (int32_t i = 0; i < c.nlevels; i++) { const size_t l = c.levels && c.levels[i] ? strlen(c.levels[i]) : 0; if (l > m) m = l; }
    return m;
  }
  return 12;
}

int format_rows(const epi_report_column *cols, int32_t ncol, int64_t lo, int64_t hi, size_t row_bound, std::vector<char> &text) {
  text.resize((size_t)(hi - lo) * row_bound + 16);
  char *p = text.data();
  for (int64_t r = lo; r < hi; r++) {
    for (int32_t c = 0; c < ncol; c++) {
      const epi_report_column &col = cols[c];
      if (c) *p++ = '\t';
      if (col.kind == EPI_COL_F64) {
        p = put_f64(p, static_cast<const double *>(col.data)[r]);
      } else {
        const int32_t v = static_cast<const int32_t *>(col.data)[r];
        if (v == INT32_MIN) continue;                      // NA_integer_
        if (col.kind == EPI_COL_FACTOR) {
          if (v >= 1 && v <= col.nlevels && col.levels && col.levels[v - 1]) {
            const size_t l = strlen(col.levels[v - 1]);
            memcpy(p, col.levels[v - 1], l);
            p += l;
          }                                                // a code outside the levels is NA
        } else {
          p = put_i32(p, v);
        }
      }
    }
    *p++ = '\n';
  }
  text.resize((size_t)(p - text.data()));
  return EPI_OK;
}

int gzip_member(const std::vector<char> &text, std::vector<char> &out) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return EPI_ERR_NOMEM;
  out.resize(deflateBound(&zs, (uLong)text.size()) + 32);
  zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(text.data()));
  zs.avail_in = (uInt)text.size();
  zs.next_out = reinterpret_cast<Bytef *>(out.data());
  zs.avail_out = (uInt)out.size();
  const int rc = deflate(&zs, Z_FINISH);
  out.resize(zs.total_out);
  deflateEnd(&zs);
  return rc == Z_STREAM_END ? EPI_OK : EPI_ERR_STATE;
}

}  // namespace

extern "C" int epi_write_report(const char *path, const epi_report_column *cols, int32_t ncol, int64_t nrow, int32_t gzip,
                                int32_t nthreads) {
  if (!path || ncol < 0 || nrow < 0 || (ncol > 0 && !cols)) return fail(EPI_ERR_ARG, "epi_write_report: bad arguments");
  for (int32_t c = 0; c < ncol; c++) {
    if (!cols[c].name || (nrow > 0 && !cols[c].data) || cols[c].kind < EPI_COL_I32 || cols[c].kind > EPI_COL_FACTOR)
      return fail(EPI_ERR_ARG, "epi_write_report: bad column %d", (int)c);
  }
  try {
    FILE *f = fopen(path, "wb");
    if (!f) return fail(EPI_ERR_ARG, "epi_write_report: cannot open %s for writing", path);
    std::vector<char> head;
    for (int32_t c = 0; c < ncol; c++) {
      if (c) head.push_back('\t');
      head.insert(head.end(), cols[c].name, cols[c].name + strlen(cols[c].name));
    }
    head.push_back('\n');
    size_t row_bound = 2;
    for (int32_t c = 0; c < ncol; c++) row_bound += field_bound(cols[c]) + 1;
    const int64_t chunk = 1 << 16;                         // rows per job: a few MiB of text
    const int64_t njobs = (nrow + chunk - 1) / chunk;
    int nt = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads);
    if ((int64_t)nt > njobs) nt = njobs > 0 ? (int)njobs : 1;
    int rc = EPI_OK;
    {                                                      // header (its own gzip member when compressing)
      std::vector<char> hz;
      const std::vector<char> *w = &head;
      if (gzip) { rc = gzip_member(head, hz); w = &hz; }
      if (rc == EPI_OK && fwrite(w->data(), 1, w->size(), f) != w->size()) rc = EPI_ERR_STATE;
    }
    // a window of 4 * nt jobs in flight: workers take jobs in order, the caller's thread writes them in order
    const int64_t window = 4 * (int64_t)nt;
    std::vector<Job> jobs((size_t)(njobs < window ? njobs : window));
    std::mutex mu;
    std::condition_variable cv_done, cv_free;
    std::atomic<int64_t> next{0};
    int64_t written = 0;
    bool stop = false;
    auto worker = [&]() {
      for (;;) {
        const int64_t j = next.fetch_add(1);
        if (j >= njobs) return;
        Job &job = jobs[(size_t)(j % (int64_t)jobs.size())];
        {
          std::unique_lock<std::mutex> lk(mu);
          cv_free.wait(lk, [&]() { return stop || j < written + (int64_t)jobs.size(); });   // the slot's previous job is on disk
          if (stop) return;
        }
        std::vector<char> text;
        job.lo = j * chunk;
        job.hi = job.lo + chunk < nrow ? job.lo + chunk : nrow;
        int jrc;
        try {
          jrc = format_rows(cols, ncol, job.lo, job.hi, row_bound, text);
          if (jrc == EPI_OK && gzip) jrc = gzip_member(text, job.out); else job.out.swap(text);
        } catch (const std::bad_alloc &) { jrc = EPI_ERR_NOMEM; }
        std::lock_guard<std::mutex> lk(mu);
        job.rc = jrc;
        job.done = true;
        cv_done.notify_all();
      }
    };
    std::vector<std::thread> th;
    if (rc == EPI_OK) for (int i = 0; i < nt; i++) th.emplace_back(worker);
    for (; rc == EPI_OK && written < njobs;) {
      Job &job = jobs[(size_t)(written % (int64_t)jobs.size())];
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&]() { return job.done; });
      lk.unlock();
      if (job.rc != EPI_OK) rc = job.rc;
      else if (fwrite(job.out.data(), 1, job.out.size(), f) != job.out.size()) rc = EPI_ERR_STATE;
      lk.lock();
      job.done = false;
      std::vector<char>().swap(job.out);
      written++;
      cv_free.notify_all();
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
      next.store(njobs);
      cv_free.notify_all();
    }
    for (auto &t : th) t.join();
    if (fclose(f) != 0 && rc == EPI_OK) rc = EPI_ERR_STATE;
    if (rc == EPI_ERR_NOMEM) return fail(rc, "epi_write_report: out of memory");
    if (rc != EPI_OK) return fail(rc, "epi_write_report: writing %s failed", path);
    return EPI_OK;
  } catch (const std::bad_alloc &) {
    return fail(EPI_ERR_NOMEM, "epi_write_report: out of memory");
  } catch (...) {
    return fail(EPI_ERR_STATE, "epi_write_report: unexpected failure");
  }
}
