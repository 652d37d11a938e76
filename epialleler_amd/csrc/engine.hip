// Engine plumbing: errors, device buffers, engine/batch lifetime, pinned
// double-buffered upload, small synchronous read-backs, HIP-event profiling.
#include "common.hpp"
#include <algorithm>
#include <sys/mman.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace epi {

int DevBuf::ensure(size_t bytes) {
  if (bytes <= cap && p) return EPI_OK;
  if (bytes == 0) bytes = 16;
  size_t want = bytes + (bytes >> 3) + 256;   // a little slack so steady-state calls never reallocate
  if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
  EPI_HIP(hipMalloc(&p, want));
  cap = want;
  return EPI_OK;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}

// A NULL `stream` is the HIP null stream (what torch's default stream is), NOT the engine's
// private stream: work must stay ordered with whatever the caller queued before.
hipStream_t pick_stream(epi_batch *b, void *stream) {
  (void)b;
  return reinterpret_cast<hipStream_t>(stream);
}

int read_scalars(epi_batch *b, hipStream_t s, const void *d_src, size_t bytes, void *h_dst) {
  if (bytes > 512) return fail(EPI_ERR_ARG, "read_scalars: too large");
  EPI_HIP(hipMemcpyAsync(b->eng->h_scalars, d_src, bytes, hipMemcpyDeviceToHost, s));
  EPI_HIP(hipStreamSynchronize(s));
  memcpy(h_dst, b->eng->h_scalars, bytes);
  return EPI_OK;
}

// ---- profiling ----------------------------------------------------------------
static bool g_prof_on = false;
static std::mutex g_prof_mu;
struct Pending { std::string name; hipEvent_t a, b; };
static std::vector<Pending> g_pending;
static std::map<std::string, ProfEntry> g_prof;
static std::map<std::string, hipEvent_t> g_open;

void prof_begin(const char *name, hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_open[name] = e;
}

void prof_end(const char *name, hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  auto it = g_open.find(name);
  if (it == g_open.end()) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_pending.push_back({name, it->second, e});
  g_open.erase(it);
}

static void prof_drain() {
  for (auto &p : g_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      g_prof[p.name].ms += ms;
      g_prof[p.name].n += 1;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  g_pending.clear();
}

}  // namespace epi

using namespace epi;

extern "C" {

int epi_version(void) { return 100; }

void epi_prof_enable(int on) { g_prof_on = on != 0; }

int epi_prof_get(const char *name, double *ms_total, int64_t *launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_drain();
  auto it = g_prof.find(name);
  if (it == g_prof.end()) { if (ms_total) *ms_total = 0; if (launches) *launches = 0; return EPI_OK; }
  if (ms_total) *ms_total = it->second.ms;
  if (launches) *launches = it->second.n;
  return EPI_OK;
}

void epi_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_drain();
  g_prof.clear();
}

int epi_engine_create(int device, epi_engine **out) {
  if (!out) return fail(EPI_ERR_ARG, "epi_engine_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(EPI_ERR_NODEVICE, "no HIP device available (%s); libepihip has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorName(e));
  if (device < 0 || device >= ndev) return fail(EPI_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
  EPI_HIP(hipSetDevice(device));
  epi_engine *eng = new epi_engine();
  eng->device = device;
  EPI_HIP(hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking));
  EPI_HIP(hipStreamCreateWithFlags(&eng->copy_stream, hipStreamNonBlocking));
  EPI_HIP(hipStreamCreateWithFlags(&eng->aux_stream, hipStreamNonBlocking));
  EPI_HIP(hipHostMalloc(reinterpret_cast<void **>(&eng->h_scalars), 512, hipHostMallocDefault));
  *out = eng;
  return EPI_OK;
}

void epi_engine_destroy(epi_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (int i = 0; i < 2; i++) {
    if (e->pinned[i]) (void)hipHostFree(e->pinned[i]);
    if (e->pinned_done[i]) (void)hipEventDestroy(e->pinned_done[i]);
  }
  for (int i = 0; i < 2; i++) {
    if (e->dev_stage[i]) (void)hipFree(e->dev_stage[i]);
    if (e->up_done[i]) (void)hipEventDestroy(e->up_done[i]);
    if (e->stage_free[i]) (void)hipEventDestroy(e->stage_free[i]);
  }
  if (e->h_scalars) (void)hipHostFree(e->h_scalars);
  if (e->aux_stream) (void)hipStreamDestroy(e->aux_stream);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  delete e;
}

int epi_engine_device(const epi_engine *e) { return e ? e->device : -1; }

// Host -> HBM through two pinned staging buffers: the CPU fills buffer k+1
// while the DMA engine drains buffer k (hipMemcpyAsync on the copy stream).
// memcpy between pageable memory and a pinned staging buffer with a few threads (one core copies ~12 GB/s, the link takes
// ~55; first-touch page faults of a fresh destination spread over the threads too).  The workers are a small persistent
// pool: spawning threads per 8 MiB piece cost more than the copies.
namespace {
struct CopyPool {
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> workers;
  char *dst = nullptr;
  const char *src = nullptr;
  size_t len = 0, part = 0;
  unsigned parts = 0, next = 0, pending = 0;
  uint64_t epoch = 0;
  bool stop = false;

  void run_part(unsigned i) {
    const size_t o = part * i;
    if (o < len) memcpy(dst + o, src + o, o + part > len ? len - o : part);
  }
  void worker() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [&]() { return stop || (epoch != seen && next < parts); });
      if (stop) return;
      while (next < parts) {
        const unsigned i = next++;
        lk.unlock();
        run_part(i);
        lk.lock();
        if (--pending == 0) cv_done.notify_all();
      }
      seen = epoch;
    }
  }
  void copy(void *d, const void *s, size_t n, unsigned k) {
    std::unique_lock<std::mutex> lk(mu);
    while (workers.size() + 1 < k) workers.emplace_back([this]() { worker(); });
    dst = static_cast<char *>(d); src = static_cast<const char *>(s); len = n;
    part = ((n + k - 1) / k + 4095) & ~(size_t)4095;       // k * part >= n
    parts = k; next = 0; pending = k; epoch++;
    cv_work.notify_all();
    while (next < parts) {                                  // the caller takes parts too
      const unsigned i = next++;
      lk.unlock();
      run_part(i);
      lk.lock();
      --pending;
    }
    cv_done.wait(lk, [&]() { return pending == 0; });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_work.notify_all();
    for (auto &t : workers) t.join();
  }
};
}  // namespace

static void parallel_memcpy(void *dst, const void *src, size_t len) {
  static const unsigned hw = std::thread::hardware_concurrency();
  const unsigned k = len < (2u << 20) ? 1u : (hw >= 16 ? 8u : hw >= 8 ? 4u : hw >= 4 ? 2u : 1u);
  if (k == 1) { memcpy(dst, src, len); return; }
  static CopyPool pool;                                     // (one copy at a time: the engine's calls are serialised by contract)
  static std::mutex one;
  std::lock_guard<std::mutex> g(one);
  pool.copy(dst, src, len, k);
}

// true when h_src is page-locked host memory the runtime knows (hipHostMalloc / hipHostRegister, e.g. the producer's
// epi_templates::xm or a torch pinned tensor): the DMA engine can read it directly
static bool is_pinned_host(const void *h_src) {
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, h_src) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

static int staged_upload(epi_engine *eng, void *d_dst, const void *h_src, size_t bytes) {
  const size_t chunk = 64u << 20;
  if (bytes && is_pinned_host(h_src)) {
    // pinned source (what the producer hands over): no staging copy, hipMemcpyAsync straight from the caller's buffer
    EPI_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, eng->copy_stream));
    return EPI_OK;
  }
  if (!eng->pinned[0]) {
    for (int i = 0; i < 2; i++) {
      EPI_HIP(hipHostMalloc(&eng->pinned[i], chunk, hipHostMallocDefault));
      EPI_HIP(hipEventCreateWithFlags(&eng->pinned_done[i], hipEventDisableTiming));
    }
    eng->pinned_bytes = chunk;
  }
  for (int i = 0; i < 2; i++) EPI_HIP(hipEventSynchronize(eng->pinned_done[i]));   // (a device-to-host copy may have used them)
  size_t done = 0;
  int k = 0;
  while (done < bytes) {
    size_t len = bytes - done < chunk ? bytes - done : chunk;
    EPI_HIP(hipEventSynchronize(eng->pinned_done[k]));   // buffer k free again (no-op before first use)
    parallel_memcpy(eng->pinned[k], static_cast<const char *>(h_src) + done, len);
    EPI_HIP(hipMemcpyAsync(static_cast<char *>(d_dst) + done, eng->pinned[k], len, hipMemcpyHostToDevice, eng->copy_stream));
    EPI_HIP(hipEventRecord(eng->pinned_done[k], eng->copy_stream));
    done += len;
    k ^= 1;
  }
  return EPI_OK;                                           // (the caller synchronises the copy stream once, after all columns)
}

// The packed bytes of an upload, laid out position-congruently while they arrive (layout.hip): the source arena crosses the
// link in pieces of 64 MB into two device-side staging pieces; as soon as a piece is there, a kernel on aux_stream moves the
// rows (and parts of rows) it holds to their places in the batch's arena while the next piece is on the link.  What the
// layout costs an upload is the last piece's kernel -- not a second pass over the arena, and no second arena.
static int upload_rows_congruent(epi_engine *eng, const uint8_t *h_xm, const int64_t *h_off, int64_t n, int64_t nbytes,
                                 const int64_t *d_src_off, const int32_t *d_len, const int64_t *d_dst_off, uint8_t *arena) {
  const bool pinned_src = is_pinned_host(h_xm);
  // piece size: a pageable source goes through the engine's 64 MB pinned buffers; a pinned one is read by the DMA engine
  // directly, in an eighth of the arena at a time (16 .. 256 MB: ~17 us of gap per piece on the link)
  size_t chunk = 64u << 20;
  if (pinned_src) {
    chunk = ((size_t)nbytes / 8 + ((1u << 20) - 1)) & ~(size_t)((1u << 20) - 1);
    if (chunk < (16u << 20)) chunk = 16u << 20;
    if (chunk > (256u << 20)) chunk = 256u << 20;
  }
  if (eng->dev_stage_bytes < chunk) {
    for (int i = 0; i < 2; i++) {
      if (eng->dev_stage[i]) { EPI_HIP(hipFree(eng->dev_stage[i])); eng->dev_stage[i] = nullptr; }
      EPI_HIP(hipMalloc(&eng->dev_stage[i], chunk));
    }
    eng->dev_stage_bytes = chunk;
  }
  for (int i = 0; i < 2; i++) {
    if (!eng->up_done[i]) EPI_HIP(hipEventCreateWithFlags(&eng->up_done[i], hipEventDisableTiming));
    if (!eng->stage_free[i]) EPI_HIP(hipEventCreateWithFlags(&eng->stage_free[i], hipEventDisableTiming));
  }
  if (!pinned_src && !eng->pinned[0]) {
    for (int i = 0; i < 2; i++) {
      EPI_HIP(hipHostMalloc(&eng->pinned[i], 64u << 20, hipHostMallocDefault));
      EPI_HIP(hipEventCreateWithFlags(&eng->pinned_done[i], hipEventDisableTiming));
    }
    eng->pinned_bytes = 64u << 20;
  }
  if (!pinned_src) for (int i = 0; i < 2; i++) EPI_HIP(hipEventSynchronize(eng->pinned_done[i]));
  const int g = layout_group(nbytes, n);
  int k = 0;
  for (int64_t c0 = 0; c0 < nbytes; k ^= 1) {
    const int64_t c1 = nbytes - c0 < (int64_t)chunk ? nbytes : c0 + (int64_t)chunk;
    const void *src = h_xm + c0;
    if (!pinned_src) {
      EPI_HIP(hipEventSynchronize(eng->pinned_done[k]));   // host buffer k drained
      parallel_memcpy(eng->pinned[k], src, (size_t)(c1 - c0));
      src = eng->pinned[k];
    }
    EPI_HIP(hipStreamWaitEvent(eng->copy_stream, eng->stage_free[k], 0));   // the kernel that emptied device piece k two pieces ago
    EPI_HIP(hipMemcpyAsync(eng->dev_stage[k], src, (size_t)(c1 - c0), hipMemcpyHostToDevice, eng->copy_stream));
    if (!pinned_src) EPI_HIP(hipEventRecord(eng->pinned_done[k], eng->copy_stream));
    EPI_HIP(hipEventRecord(eng->up_done[k], eng->copy_stream));
    // rows with a byte, or their end, in (c0, c1]: from the first row that ends behind c0 (every row in the first piece) to the
    // last one that starts at or before c1 (the kernel clips; a row that only touches c1 does nothing)
    const int64_t ra = c0 == 0 ? 0 : (std::upper_bound(h_off + 1, h_off + n + 1, c0) - (h_off + 1));
    const int64_t rb = (std::upper_bound(h_off, h_off + n, c1) - h_off) - 1;
    EPI_HIP(hipStreamWaitEvent(eng->aux_stream, eng->up_done[k], 0));
    if (rb >= ra)
      EPI_TRY(layout_copy_range(static_cast<const uint8_t *>(eng->dev_stage[k]), c0, c1, d_src_off, d_len, d_dst_off, ra, rb - ra + 1, g, arena,
                                eng->aux_stream));
    EPI_HIP(hipEventRecord(eng->stage_free[k], eng->aux_stream));
    c0 = c1;
  }
  return EPI_OK;
}

}  // extern "C"

namespace epi {

static int ensure_staging(epi_engine *eng) {
  const size_t chunk = 64u << 20;
  if (!eng->pinned[0]) {
    for (int i = 0; i < 2; i++) {
      EPI_HIP(hipHostMalloc(&eng->pinned[i], chunk, hipHostMallocDefault));
      EPI_HIP(hipEventCreateWithFlags(&eng->pinned_done[i], hipEventDisableTiming));
    }
    eng->pinned_bytes = chunk;
  }
  return EPI_OK;
}

// A large pageable destination that has not been touched yet (a column the caller has just allocated) is faulted in by the
// copy below: with transparent huge pages in `madvise` mode the hint turns ~40 000 4 KiB faults of a 10 M-row table into ~80.
static void advise_huge(void *p, size_t n) {
  if (n < (4u << 20) || options().no_hugepage) return;
  const uintptr_t two = (uintptr_t)2 << 20;
  const uintptr_t lo = (reinterpret_cast<uintptr_t>(p) + two - 1) & ~(two - 1), hi = (reinterpret_cast<uintptr_t>(p) + n) & ~(two - 1);
  if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
}

int copy_parts_to_host(epi_engine *eng, const CopyPart *parts, int nparts, hipStream_t s) {
  const size_t piece = 8u << 20;                           // (the staging buffers hold 64 MiB each; 8 MiB pieces keep the
                                                           // first copy-out early and the last one short)
  struct Pending { void *dst; size_t len; int buf; };
  Pending pend = {nullptr, 0, 0};
  bool have = false, staged = false;
  int k = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  auto drain = [&]() -> int {                              // wait for the piece in flight, copy it out of its staging buffer
    if (!have) return EPI_OK;
    EPI_HIP(hipEventSynchronize(ev[pend.buf]));
    parallel_memcpy(pend.dst, eng->pinned[pend.buf], pend.len);
    have = false;
    return EPI_OK;
  };
  for (int i = 0; i < nparts; i++) {
    const CopyPart &p = parts[i];
    if (!p.bytes) continue;
    if (is_pinned_host(p.dst)) { EPI_HIP(hipMemcpyAsync(p.dst, p.src, p.bytes, hipMemcpyDeviceToHost, s)); continue; }
    advise_huge(p.dst, p.bytes);
    if (!staged) {
      EPI_TRY(ensure_staging(eng));
      for (int j = 0; j < 2; j++) EPI_HIP(hipEventCreateWithFlags(&ev[j], hipEventDisableTiming));
      staged = true;
    }
    for (size_t done = 0; done < p.bytes; done += piece) {
      const size_t len = p.bytes - done < piece ? p.bytes - done : piece;
      // buffer k was drained two pieces ago (drain() below runs before it is reused)
      hipError_t e1 = hipMemcpyAsync(eng->pinned[k], static_cast<const char *>(p.src) + done, len, hipMemcpyDeviceToHost, s);
      hipError_t e2 = e1 == hipSuccess ? hipEventRecord(ev[k], s) : e1;
      int rc = drain();                                    // the previous piece leaves its buffer while this one is on the link
      if (e2 != hipSuccess || rc != EPI_OK) {
        (void)hipStreamSynchronize(s);
        for (int j = 0; j < 2; j++) if (ev[j]) (void)hipEventDestroy(ev[j]);
        return rc != EPI_OK ? rc : fail(EPI_ERR_HIP, "device to host copy failed: %s", hipGetErrorName(e2));
      }
      pend = {static_cast<char *>(p.dst) + done, len, k};
      have = true;
      k ^= 1;
    }
  }
  int rc = drain();
  hipError_t e = hipStreamSynchronize(s);
  for (int j = 0; j < 2; j++) if (ev[j]) (void)hipEventDestroy(ev[j]);
  if (rc != EPI_OK) return rc;
  if (e != hipSuccess) return fail(EPI_ERR_HIP, "device to host copy failed: %s", hipGetErrorName(e));
  return EPI_OK;
}

int copy_to_host(epi_engine *eng, void *h_dst, const void *d_src, size_t bytes, hipStream_t s) {
  const CopyPart p = {h_dst, d_src, bytes};
  return copy_parts_to_host(eng, &p, 1, s);
}

}  // namespace epi

extern "C" {

int epi_batch_upload(epi_engine *e, const uint8_t *xm, const int64_t *off, const int32_t *rname,
                     const int32_t *strand, const int32_t *start, int64_t n, epi_batch **out) {
  if (!e || !out) return fail(EPI_ERR_ARG, "epi_batch_upload: NULL engine/out");
  *out = nullptr;
  if (n < 0 || n > 0x7FFFFFF0LL) return fail(EPI_ERR_ARG, "epi_batch_upload: n=%lld out of range", (long long)n);
  if (!off) return fail(EPI_ERR_ARG, "epi_batch_upload: off is NULL");
  if (n > 0 && (!rname || !strand || !start)) return fail(EPI_ERR_ARG, "epi_batch_upload: NULL column");
  const int64_t nbytes = off[n] - off[0];
  if (off[0] != 0) return fail(EPI_ERR_ARG, "epi_batch_upload: off[0] must be 0");
  if (nbytes < 0 || (nbytes > 0 && !xm)) return fail(EPI_ERR_ARG, "epi_batch_upload: bad xm/off");
  EPI_HIP(hipSetDevice(e->device));
  epi_batch *b = new epi_batch();
  b->eng = e;
  b->n = n;
  b->nbytes = nbytes;
  b->owns = true;
  int rc = EPI_OK;
  const int modulus = (n > 0 && nbytes > 0) ? options().realign : 0;   // EPIHIP_REALIGN=0: rows stay as uploaded
  DevBuf new_off;
  do {
    if ((rc = b->own_off.ensure((size_t)(n + 1) * 8))) break;
    if ((rc = b->own_rname.ensure((size_t)n * 4 + 4))) break;
    if ((rc = b->own_strand.ensure((size_t)n * 4 + 4))) break;
    if ((rc = b->own_start.ensure((size_t)n * 4 + 4))) break;
    if ((rc = staged_upload(e, b->own_off.p, off, (size_t)(n + 1) * 8))) break;
    if (n) {
      if ((rc = staged_upload(e, b->own_rname.p, rname, (size_t)n * 4))) break;
      if ((rc = staged_upload(e, b->own_strand.p, strand, (size_t)n * 4))) break;
      if ((rc = staged_upload(e, b->own_start.p, start, (size_t)n * 4))) break;
    }
    b->off = b->own_off.as<int64_t>();
    b->rname = b->own_rname.as<int32_t>();
    b->strand = b->own_strand.as<int32_t>();
    b->start = b->own_start.as<int32_t>();
    // row lengths and statistics behind the columns, on the copy stream (the first report reads the verdict)
    if ((rc = launch_row_stats(b, e->copy_stream))) break;
    if (modulus) {
      // The batch owns its arena: rows go where the tile kernels read them fastest -- at offsets congruent to their start
      // position (layout.hip) -- while the bytes arrive.  The offsets need the columns only.
      unsigned long long h_end = 0;
      uint32_t head = 0;
      if ((rc = layout_offsets(b, modulus, e->copy_stream, &new_off, &h_end, &head))) break;
      const size_t cap = ((size_t)h_end + 15) / 16 * 16 + 64;
      if ((rc = b->own_xm.ensure(cap))) break;
      uint8_t *arena = b->own_xm.as<uint8_t>();
      if ((head && hipMemsetAsync(arena, 0xFB, head, e->aux_stream) != hipSuccess) ||
          hipMemsetAsync(arena + h_end, 0xFB, cap - (size_t)h_end, e->aux_stream) != hipSuccess) { rc = fail(EPI_ERR_HIP, "memset failed"); break; }
      if ((rc = upload_rows_congruent(e, xm, off, n, nbytes, b->off, b->len, new_off.as<int64_t>(), arena))) break;
      if (hipStreamSynchronize(e->copy_stream) != hipSuccess || hipStreamSynchronize(e->aux_stream) != hipSuccess) {
        rc = fail(EPI_ERR_HIP, "upload failed: %s", hipGetErrorName(hipGetLastError())); break;
      }
      std::swap(b->own_off, new_off);
      b->off = b->own_off.as<int64_t>();
      b->nbytes = (int64_t)h_end;
      b->congruent = modulus;
    } else {
      const size_t cap = ((size_t)nbytes + 15) / 16 * 16 + 64;
      if ((rc = b->own_xm.ensure(cap))) break;
      if (hipMemsetAsync(static_cast<char *>(b->own_xm.p) + nbytes, 0xFB, cap - nbytes, e->copy_stream) != hipSuccess) {
        rc = fail(EPI_ERR_HIP, "memset failed"); break;
      }
      if (nbytes && (rc = staged_upload(e, b->own_xm.p, xm, (size_t)nbytes))) break;
      if (hipStreamSynchronize(e->copy_stream) != hipSuccess) { rc = fail(EPI_ERR_HIP, "upload failed: %s", hipGetErrorName(hipGetLastError())); break; }
    }
  } while (0);
  new_off.release();
  if (rc) {
    (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamSynchronize(e->aux_stream);   // nothing of the batch is in flight when it goes
    epi_batch_free(b);
    return rc;
  }
  b->xm = b->own_xm.as<uint8_t>();
  b->cols_owned = true;
  *out = b;
  return EPI_OK;
}

int epi_batch_adopt(epi_engine *e, const uint8_t *d_xm, int64_t xm_capacity, int64_t nbytes,
                    const int64_t *d_off, const int32_t *d_rname, const int32_t *d_strand,
                    const int32_t *d_start, int64_t n, epi_batch **out) {
  if (!e || !out) return fail(EPI_ERR_ARG, "epi_batch_adopt: NULL engine/out");
  *out = nullptr;
  if (n < 0 || n > 0x7FFFFFF0LL) return fail(EPI_ERR_ARG, "epi_batch_adopt: n out of range");
  if (!d_off || (n > 0 && (!d_rname || !d_strand || !d_start))) return fail(EPI_ERR_ARG, "epi_batch_adopt: NULL column");
  if (nbytes < 0 || (nbytes > 0 && !d_xm)) return fail(EPI_ERR_ARG, "epi_batch_adopt: bad xm");
  if ((reinterpret_cast<uintptr_t>(d_xm) & 15) != 0) return fail(EPI_ERR_ARG, "epi_batch_adopt: d_xm must be 16-byte aligned");
  if (xm_capacity < (nbytes + 15) / 16 * 16)
    return fail(EPI_ERR_ARG, "epi_batch_adopt: xm_capacity %lld < nbytes rounded up to 16 (%lld)",
                (long long)xm_capacity, (long long)((nbytes + 15) / 16 * 16));
  epi_batch *b = new epi_batch();
  b->eng = e;
  b->n = n;
  b->nbytes = nbytes;
  b->owns = false;
  b->xm = d_xm;
  b->off = d_off;
  b->rname = d_rname;
  b->strand = d_strand;
  b->start = d_start;
  // length / order / strand statistics of the (from now on immutable) columns, queued on the null stream: the
  // caller's columns must be complete as seen from that stream, as for every later call with stream = NULL
  EPI_HIP(hipSetDevice(e->device));
  const int rc = launch_row_stats(b, nullptr);
  if (rc) { epi_batch_free(b); return rc; }
  *out = b;
  return EPI_OK;
}

void epi_batch_free(epi_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->eng->device);
  DevBuf *bufs[] = {&b->own_xm, &b->own_off, &b->own_len, &b->own_rname, &b->own_strand, &b->own_start, &b->stats,
                    &b->scan_tmp, &b->tiles, &b->tile_nrow, &b->tile_base,
                    &b->tile_out, &b->pool_key, &b->pool_a, &b->pool_b, &b->pool_c, &b->pool_d, &b->pool_e, &b->pool_f,
                    &b->misc, &b->mhl_m, &b->mhl_h, &b->mhl_blk, &b->mhl_cont, &b->mhl_cur, &b->d_shared_keys, &b->d_shared_owned, &b->heavy_list, &b->heavy_slab, &b->heavy_sums, &b->deep_list, &b->mhlf_fold_slab, &b->diag, &b->d_slot_tile, &b->pass_tmp, &b->thr_tab, &b->mhl_keep_tab, &b->host_io, &b->tile_bsum[0], &b->tile_bsum[1], &b->tile_bsum[2], &b->tile_bsum[3], &b->own_slab, &b->own_slab2, &b->tiles_nt_dev};
  for (DevBuf *d : bufs) d->release();
  if (b->stats_done) (void)hipEventDestroy(b->stats_done);
  delete b;
}

int64_t epi_batch_nrows(const epi_batch *b) { return b ? b->n : -1; }

void epi_cx_table_free(epi_cx_table *t) {                // (the columns are one allocation, capi.hip: rname is its start)
  if (!t) return;
  epi::table_block_release(t->rname);
  memset(t, 0, sizeof(*t));
}

void epi_mhl_table_free(epi_mhl_table *t) {              // (one allocation: length is its start)
  if (!t) return;
  epi::table_block_release(t->length);
  memset(t, 0, sizeof(*t));
}

}  // extern "C"
