// Row-range sharded reports over the GPUs of one node with RCCL called DIRECTLY behind the C ABI: a host in
// any language -- the R shim has no torch -- creates one epi_comm per process / GPU and calls one entry point per report.
//
//   epi_comm_unique_id (one rank)  ->  the 128 bytes travel by whatever the host has (a file, a socket, MPI, torch)
//   epi_comm_create (every rank)   ->  ncclCommInitRank
//   epi_batch_cytosine_report_sharded / epi_batch_mhl_report_sharded (every rank, its own contiguous range of the
//   globally sorted rows, rank order = row order):
//     1. ncclAllGather of every rank's (first, last) tile key -- once per batch, tile grid and communicator, remembered;
//     2. every rank derives the same sorted list of shared tile keys and their owners (lowest rank that reaches the key);
//     3. the tile kernels run as for a single GPU, except that shared tiles dump their raw sums into a slab;
//     4. ncclAllReduce(sum) of the slab(s) on the report's stream -- the one data-path collective, 64-128 KiB per shared tile;
//     5. owners apply the rule to their shared tiles, rows are ordered: this rank's rows in table order.
//   The rows of rank r precede those of rank r + 1 in the reference's table; fetch them with epi_batch_cx_fetch_* /
//   epi_batch_mhl_fetch_* as after a single-GPU report.
//
// librccl is loaded with dlopen on first use: the library itself (and every single-GPU caller) does not depend on it.
#include "common.hpp"
#include <dlfcn.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <rccl/rccl.h>

namespace epi {

struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

static Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.h, "ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.h, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.h, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather && r.GetErrorString;
  });
  return r;
}

#define EPI_NCCL(expr)                                                                                     \
  do {                                                                                                     \
    ncclResult_t _r = (expr);                                                                              \
    if (_r != ncclSuccess) return ::epi::fail(EPI_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(_r)); \
  } while (0)

// Keys shared by at least two ranks and their owners, from every rank's (first, last) key range (first > last: the rank
// holds no rows).  Rows are globally sorted, so two ranges can only overlap inside one reference sequence.
static int shared_keys_of(const std::vector<int64_t> &ranges, int world, std::vector<int64_t> *keys, std::vector<int32_t> *owner) {
  std::map<int64_t, int32_t> m;
  for (int i = 0; i < world; i++) {
    const int64_t fi = ranges[2 * i], li = ranges[2 * i + 1];
    if (fi > li) continue;
    for (int j = i + 1; j < world; j++) {
      const int64_t fj = ranges[2 * j], lj = ranges[2 * j + 1];
      if (fj > lj) continue;
      const int64_t lo = std::max(fi, fj), hi = std::min(li, lj);
      if (lo > hi) continue;
      if ((lo >> 32) != (hi >> 32))
        return fail(EPI_ERR_ARG, "rank ranges overlap across reference sequences: the shards are not contiguous ranges of one sorted row stream");
      if (hi - lo > (1 << 22)) return fail(EPI_ERR_ARG, "more than 4 M shared tiles between two ranks");
      for (int64_t k = lo; k <= hi; k++) m.emplace(k, (int32_t)i);   // (i ascends: the first insert is the lowest rank)
    }
  }
  keys->clear(); owner->clear();
  for (const auto &kv : m) { keys->push_back(kv.first); owner->push_back(kv.second); }
  return EPI_OK;
}

}  // namespace epi

struct epi_comm {
  epi_engine *eng = nullptr;
  ncclComm_t nccl = nullptr;
  int rank = 0, world = 1;
  int test_shared = 0;                        // test hook (world size 1): this many tiles in the middle of the batch are treated as shared
  epi::DevBuf d_gather;                       // all-gather buffers (a few int64 per rank)
  int64_t last_bytes = 0;                     // bytes this rank handed to the last report's all-reduce(s)
  uint64_t serial = 0;                        // identity for the ranges remembered on a batch
};

using namespace epi;

namespace {

std::atomic<uint64_t> g_comm_serial{1};

// all ranks' values (nvals int64 per rank) -> host
int all_gather_i64(epi_comm *c, const int64_t *mine, int nvals, hipStream_t s, std::vector<int64_t> *all) {
  all->assign((size_t)nvals * c->world, 0);
  if (!c->nccl) { std::copy(mine, mine + nvals, all->begin()); return EPI_OK; }   // (world size 1 without a communicator)
  EPI_TRY(c->d_gather.ensure((size_t)nvals * 8 * (c->world + 1)));
  int64_t *d_send = c->d_gather.as<int64_t>(), *d_recv = d_send + nvals;
  EPI_HIP(hipMemcpyAsync(d_send, mine, (size_t)nvals * 8, hipMemcpyHostToDevice, s));
  EPI_NCCL(rccl().AllGather(d_send, d_recv, (size_t)nvals, ncclInt64, c->nccl, s));
  EPI_HIP(hipMemcpyAsync(all->data(), d_recv, (size_t)nvals * 8 * c->world, hipMemcpyDeviceToHost, s));
  EPI_HIP(hipStreamSynchronize(s));
  return EPI_OK;
}

// world size 1 with the test hook: `n` consecutive tile keys in the middle of the rank's range, inside one reference sequence
void forced_keys(int64_t first, int64_t last, int n, std::vector<int64_t> *keys, std::vector<int32_t> *owner) {
  keys->clear(); owner->clear();
  if (first > last || n <= 0) return;
  int64_t mid = first + (last - first) / 2;
  if ((mid >> 32) != (first >> 32)) mid = first;            // (the range spans reference sequences: start at its first tile)
  for (int64_t k = mid; k < mid + n && k <= last && (k >> 32) == (mid >> 32); k++) { keys->push_back(k); owner->push_back(0); }
}

}  // namespace

struct epi_shard_plan {                         // what a (batch, tile grid, communicator) triple needs per report; remembered on the batch
  uint64_t comm_serial = 0;
  int T = 0, kind = 0;
  std::vector<int64_t> keys;
  std::vector<int32_t> owned;
  bool fused = false;                           // lMHL: all ranks can take the one-pass kernel
};

extern "C" {

// The shared tiles of a set of key ranges (pure host logic; what every rank derives from the all-gathered ranges).
int epi_shared_tile_keys(const int64_t *ranges, int32_t world, int64_t *keys_out, int32_t *owner_out, int32_t cap, int32_t *n_out) {
  if (!ranges || world < 1 || !n_out || (cap > 0 && (!keys_out || !owner_out))) return fail(EPI_ERR_ARG, "epi_shared_tile_keys: bad arguments");
  std::vector<int64_t> all(ranges, ranges + 2 * (size_t)world), keys;
  std::vector<int32_t> owner;
  EPI_TRY(shared_keys_of(all, world, &keys, &owner));
  *n_out = (int32_t)keys.size();
  if ((int64_t)keys.size() > cap) return cap > 0 ? fail(EPI_ERR_ARG, "epi_shared_tile_keys: %zu keys do not fit %d", keys.size(), cap) : EPI_OK;
  std::copy(keys.begin(), keys.end(), keys_out);
  std::copy(owner.begin(), owner.end(), owner_out);
  return EPI_OK;
}

int epi_comm_unique_id(void *id_out) {
  if (!id_out) return fail(EPI_ERR_ARG, "epi_comm_unique_id: NULL argument");
  if (!rccl().ok) return fail(EPI_ERR_NODEVICE, "librccl could not be loaded: sharded reports need RCCL");
  static_assert(EPI_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId id;
  EPI_NCCL(rccl().GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return EPI_OK;
}

int epi_comm_create(epi_engine *eng, const void *id, int rank, int world, epi_comm **out) {
  if (!eng || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return fail(EPI_ERR_ARG, "epi_comm_create: bad arguments");
  *out = nullptr;
  EPI_HIP(hipSetDevice(eng->device));
  epi_comm *c = new epi_comm();
  c->eng = eng; c->rank = rank; c->world = world;
  c->serial = g_comm_serial.fetch_add(1);
  if (id) {                                                  // (world size 1 without an id: no communicator, nothing to exchange)
    if (!rccl().ok) { delete c; return fail(EPI_ERR_NODEVICE, "librccl could not be loaded: sharded reports need RCCL"); }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    const ncclResult_t r = rccl().CommInitRank(&c->nccl, world, uid, rank);
    if (r != ncclSuccess) { delete c; return fail(EPI_ERR_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(r)); }
  }
  *out = c;
  return EPI_OK;
}

void epi_comm_free(epi_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->eng->device);
  if (c->nccl) (void)rccl().CommDestroy(c->nccl);
  c->d_gather.release();
  delete c;
}

int epi_comm_rank(const epi_comm *c) { return c ? c->rank : -1; }
int epi_comm_world(const epi_comm *c) { return c ? c->world : 0; }
int64_t epi_comm_last_exchange_bytes(const epi_comm *c) { return c ? c->last_bytes : 0; }
void epi_comm_set_test_shared(epi_comm *c, int ntiles) { if (c) { c->test_shared = ntiles > 0 ? ntiles : 0; c->serial = g_comm_serial.fetch_add(1); } }

int epi_batch_cytosine_report_sharded(epi_batch *b, epi_comm *c, const char *ctx_meth, const char *ctx_unmeth, const char *ooctx_meth,
                                      const char *ooctx_unmeth, uint32_t min_n_ctx, double min_ctx_meth_frac, double max_ooctx_meth_frac,
                                      const int32_t *d_pass, const char *ctx, int32_t *d_pass_out, void *stream, int64_t *nrow_out) {
  if (!b || !c || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_cytosine_report_sharded: NULL argument");
  if (b->eng != c->eng) return fail(EPI_ERR_ARG, "epi_batch_cytosine_report_sharded: batch and communicator live on different engines");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  *nrow_out = 0;
  c->last_bytes = 0;
  const int T = epi_cx_tile_positions(ctx);
  // 1, 2: shared tiles -- a property of the shards, the tile grid and the communicator: exchanged once, remembered on the batch
  epi_shard_plan *plan = nullptr;
  for (auto &p : b->shard_plans) if (p->comm_serial == c->serial && p->T == T && p->kind == 0) plan = p.get();
  if (!plan) {
    int64_t mine[2] = {0, -1};
    EPI_TRY(epi_batch_tile_key_range_for(b, T, s, &mine[0], &mine[1]));
    std::vector<int64_t> all;
    EPI_TRY(all_gather_i64(c, mine, 2, s, &all));
    std::shared_ptr<epi_shard_plan> np(new epi_shard_plan());
    np->comm_serial = c->serial; np->T = T; np->kind = 0;
    std::vector<int32_t> owner;
    if (c->world == 1 && c->test_shared > 0) forced_keys(mine[0], mine[1], c->test_shared, &np->keys, &owner);
    else EPI_TRY(shared_keys_of(all, c->world, &np->keys, &owner));
    np->owned.resize(owner.size());
    for (size_t i = 0; i < owner.size(); i++) np->owned[i] = owner[i] == c->rank ? 1 : 0;
    b->shard_plans.push_back(std::move(np));
    plan = b->shard_plans.back().get();
  }
  const int32_t nshared = (int32_t)plan->keys.size();
  // 3: accumulate (shared tiles into the slab)
  const size_t slab_bytes = (size_t)nshared * kCxPlanes * T * 4;
  if (nshared > 0) {
    EPI_TRY(b->own_slab.ensure(slab_bytes));
    EPI_HIP(hipMemsetAsync(b->own_slab.p, 0, slab_bytes, s));
  }
  EPI_TRY(epi_batch_cx_set_shared(b, plan->keys.data(), plan->owned.data(), nshared, nshared ? b->own_slab.as<int32_t>() : nullptr));
  int rc;
  int64_t nrow = 0;
  auto first_half = [&]() {
    if (ctx_meth)
      return epi_batch_cytosine_report_dev(b, ctx_meth, ctx_unmeth, ooctx_meth, ooctx_unmeth, min_n_ctx, min_ctx_meth_frac, max_ooctx_meth_frac,
                                           ctx, d_pass_out, s, &nrow);
    return epi_batch_cx_report_dev(b, d_pass, ctx, s, &nrow);
  };
  // ONE host synchronisation per report where the batch allows it (an earlier report found no ultra-deep tile the host
  // would have to finish before the slab travels): the first half only queues its kernels, the all-reduce and the owners'
  // emit are queued behind them, and everything is read back and checked once, in the second half.
  b->cx_defer = true;
  rc = first_half();
  b->cx_defer = false;
  if (rc == EPI_OK && nshared > 0) {
    // 4: the one data-path collective, queued on the report's stream behind the tile kernels
    if (c->nccl) {
      const ncclResult_t r = rccl().AllReduce(b->own_slab.p, b->own_slab.p, slab_bytes / 4, ncclInt32, ncclSum, c->nccl, s);
      if (r != ncclSuccess) rc = fail(EPI_ERR_HIP, "ncclAllReduce failed: %s", rccl().GetErrorString(r));
      c->last_bytes = (int64_t)slab_bytes;
    }
    // 5: owners emit their shared tiles; rows ordered
    if (rc == EPI_OK) rc = epi_batch_cx_finish_shared(b, ctx, s, &nrow);
    if (rc == EPI_RETRY_POOL) {
      // (deferred synchronisation only) the row pool was too small: the first half again, with its own synchronisation --
      // it grows the pool -- into a scratch slab; the shared tiles are then emitted from the slab that is already reduced
      rc = b->own_slab2.ensure(slab_bytes);
      if (rc == EPI_OK && hipMemsetAsync(b->own_slab2.p, 0, slab_bytes, s) != hipSuccess) rc = fail(EPI_ERR_HIP, "hipMemsetAsync failed");
      if (rc == EPI_OK) rc = epi_batch_cx_set_shared(b, plan->keys.data(), plan->owned.data(), nshared, b->own_slab2.as<int32_t>());
      if (rc == EPI_OK) rc = first_half();
      if (rc == EPI_OK) rc = epi_batch_cx_set_shared(b, plan->keys.data(), plan->owned.data(), nshared, b->own_slab.as<int32_t>());
      if (rc == EPI_OK) rc = epi_batch_cx_finish_shared(b, ctx, s, &nrow);
    }
  }
  (void)epi_batch_cx_set_shared(b, nullptr, nullptr, 0, nullptr);     // later single-GPU calls on this batch emit every tile
  if (rc != EPI_OK) return rc;
  *nrow_out = nrow;
  return EPI_OK;
}

int epi_batch_mhl_report_sharded(epi_batch *b, epi_comm *c, const char *ctx, int hmax, int hmin, double max_ooctx_meth_frac, void *stream,
                                 int64_t *nrow_out) {
  if (!b || !c || !ctx || !nrow_out) return fail(EPI_ERR_ARG, "epi_batch_mhl_report_sharded: NULL argument");
  if (b->eng != c->eng) return fail(EPI_ERR_ARG, "epi_batch_mhl_report_sharded: batch and communicator live on different engines");
  EPI_HIP(hipSetDevice(b->eng->device));
  hipStream_t s = pick_stream(b, stream);
  *nrow_out = 0;
  c->last_bytes = 0;
  // the plan depends on the context string only through "can every rank take the one-pass kernel": keyed by its hash
  int kind = 1;
  for (const unsigned char *p = reinterpret_cast<const unsigned char *>(ctx); *p; p++) kind = kind * 31 + *p;
  kind |= 1 << 30;
  epi_shard_plan *plan = nullptr;
  for (auto &p : b->shard_plans) if (p->comm_serial == c->serial && p->kind == kind) plan = p.get();
  if (!plan) {
    // both tile grids' ranges and whether this rank's rows allow the one-pass kernel: one all-gather decides path and tiles
    int32_t ok = 0;
    EPI_TRY(epi_batch_mhl_fused_ok(b, ctx, s, &ok));
    int64_t mine[5] = {0, -1, 0, -1, ok};
    EPI_TRY(epi_batch_tile_key_range_for(b, epi_mhl_tile_positions(), s, &mine[0], &mine[1]));
    if (ok) EPI_TRY(epi_batch_tile_key_range_for(b, epi_mhl_fused_tile_positions(), s, &mine[2], &mine[3]));
    std::vector<int64_t> all;
    EPI_TRY(all_gather_i64(c, mine, 5, s, &all));
    bool fused = true;
    for (int r = 0; r < c->world; r++) fused = fused && all[5 * r + 4] != 0;
    std::vector<int64_t> ranges((size_t)2 * c->world);
    for (int r = 0; r < c->world; r++) { ranges[2 * r] = all[5 * r + (fused ? 2 : 0)]; ranges[2 * r + 1] = all[5 * r + (fused ? 3 : 1)]; }
    std::shared_ptr<epi_shard_plan> np(new epi_shard_plan());
    np->comm_serial = c->serial; np->kind = kind; np->fused = fused;
    np->T = fused ? epi_mhl_fused_tile_positions() : epi_mhl_tile_positions();
    std::vector<int32_t> owner;
    if (c->world == 1 && c->test_shared > 0) forced_keys(ranges[0], ranges[1], c->test_shared, &np->keys, &owner);
    else EPI_TRY(shared_keys_of(ranges, c->world, &np->keys, &owner));
    np->owned.resize(owner.size());
    for (size_t i = 0; i < owner.size(); i++) np->owned[i] = owner[i] == c->rank ? 1 : 0;
    b->shard_plans.push_back(std::move(np));
    plan = b->shard_plans.back().get();
  }
  const int32_t nshared = (int32_t)plan->keys.size();
  const int T = plan->T;
  const size_t cnt_bytes = plan->fused ? (size_t)nshared * 4 * T * 4 : (size_t)nshared * 16 * T * 4;
  const size_t sum_bytes = plan->fused ? (size_t)nshared * 6 * T * 8 : (size_t)nshared * (size_t)epi_mhl_slab_sums() * 8;
  if (nshared > 0) {
    EPI_TRY(b->own_slab.ensure(cnt_bytes));
    EPI_TRY(b->own_slab2.ensure(sum_bytes));
    EPI_HIP(hipMemsetAsync(b->own_slab.p, 0, cnt_bytes, s));
    EPI_HIP(hipMemsetAsync(b->own_slab2.p, 0, sum_bytes, s));
  }
  if (plan->fused)
    EPI_TRY(epi_batch_mhl_set_shared_fused(b, plan->keys.data(), plan->owned.data(), nshared, nshared ? b->own_slab.as<int32_t>() : nullptr,
                                           nshared ? b->own_slab2.as<int64_t>() : nullptr));
  else
    EPI_TRY(epi_batch_mhl_set_shared(b, plan->keys.data(), plan->owned.data(), nshared, nshared ? b->own_slab.as<int32_t>() : nullptr,
                                     nshared ? b->own_slab2.as<int64_t>() : nullptr));
  int64_t nrow = 0;
  int rc = epi_batch_mhl_report_dev(b, ctx, hmax, hmin, max_ooctx_meth_frac, s, &nrow);
  if (rc == EPI_OK && nshared > 0) {
    if (c->nccl) {
      ncclResult_t r = rccl().AllReduce(b->own_slab.p, b->own_slab.p, cnt_bytes / 4, ncclInt32, ncclSum, c->nccl, s);
      if (r == ncclSuccess) r = rccl().AllReduce(b->own_slab2.p, b->own_slab2.p, sum_bytes / 8, ncclInt64, ncclSum, c->nccl, s);
      if (r != ncclSuccess) rc = fail(EPI_ERR_HIP, "ncclAllReduce failed: %s", rccl().GetErrorString(r));
      c->last_bytes = (int64_t)(cnt_bytes + sum_bytes);
    }
    if (rc == EPI_OK) rc = epi_batch_mhl_finish_shared(b, s, &nrow);
  }
  (void)epi_batch_mhl_set_shared(b, nullptr, nullptr, 0, nullptr, nullptr);
  if (rc != EPI_OK) return rc;
  *nrow_out = nrow;
  return EPI_OK;
}

}  // extern "C"
