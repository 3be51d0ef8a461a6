// draco-sharp_amd/csrc/dsa_pool.h
// Multi-GPU submit behind the C-ABI (include/draco_mi355x.h, dsa_pool_*): one context + one worker thread per listed
// device, a job's streams sorted by compressed length and handed out chunk by chunk through one atomic counter
// (SURVEY.md section 8e).  Meshes are independent (DracoDecoder.cs:19-42 creates everything per call), so there is no
// collective and no peer traffic; a chunk is an ordinary dsa_batch on the worker's context.  Included by dsa_api.hip.
#pragma once
#include <algorithm>
#include <atomic>
#include <mutex>
#include <numeric>
#include <thread>

struct dsa_pool {
  std::vector<dsa_context *> ctx;
  uint32_t chunk = 256;
  std::string err;
  std::mutex busy;                       // one dsa_pool_decode at a time
  std::atomic<uint32_t> live_jobs{0};    // jobs not yet freed: their batches point at this pool's contexts
  bool doomed = false;                   // dsa_pool_destroy was called while jobs were alive: the last job frees the pool
};

struct dsa_pool_job {
  struct Where { uint32_t chunk, index; };
  dsa_pool *pool = nullptr;
  std::vector<dsa_batch *> batches;      // one per chunk
  std::vector<uint32_t> worker;          // which context decoded the chunk
  std::vector<Where> where;              // per stream of the job
};

extern "C" {

uint32_t dsa_pool_plan(uint32_t n, const size_t *lengths, uint32_t chunk_meshes, uint32_t *order, uint32_t *chunk_begin) {
  if (!lengths || !order || !chunk_begin || n == 0) { if (chunk_begin) chunk_begin[0] = 0; return 0; }
  if (chunk_meshes == 0) chunk_meshes = 1;
  std::iota(order, order + n, 0u);
  std::stable_sort(order, order + n, [&](uint32_t a, uint32_t b) { return lengths[a] > lengths[b]; });   // longest first, ties by index
  uint32_t chunks = 0;
  for (uint32_t at = 0; at < n; at += chunk_meshes) chunk_begin[chunks++] = at;
  chunk_begin[chunks] = n;
  return chunks;
}

static dsa_status pool_create(const int *devices, uint32_t num_devices, uint32_t chunk_meshes, dsa_pool **out) {
  dsa_pool *p = new dsa_pool();
  p->chunk = std::min<uint32_t>(chunk_meshes, 65535u);      // 0: chosen per job (pool_decode)
  for (uint32_t i = 0; i < num_devices; ++i) {
    dsa_context *c = nullptr;
    dsa_status st = dsa_context_create(devices[i], nullptr, &c);
    if (st != DSA_OK) { for (dsa_context *x : p->ctx) dsa_context_destroy(x); delete p; return st; }
    p->ctx.push_back(c);
  }
  *out = p;
  return DSA_OK;
}
dsa_status dsa_pool_create(const int *devices, uint32_t num_devices, uint32_t chunk_meshes, dsa_pool **out) {
  if (!out) return DSA_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!devices || num_devices == 0 || num_devices > 64) return DSA_ERR_INVALID_ARGUMENT;
  DSA_GUARD((dsa_context *)nullptr, pool_create(devices, num_devices, chunk_meshes, out));
}

static void pool_really_destroy(dsa_pool *pool) {
  for (dsa_context *c : pool->ctx) dsa_context_destroy(c);
  delete pool;
}
// Jobs hold batches that live on the pool's contexts: a pool destroyed before its jobs stays alive until the last of them is
// freed (the order of the two calls does not matter to the caller).
void dsa_pool_destroy(dsa_pool *pool) {
  if (!pool) return;
  bool now;
  { std::lock_guard<std::mutex> g(pool->busy); pool->doomed = true; now = pool->live_jobs.load() == 0; }
  if (now) pool_really_destroy(pool);
}
uint32_t dsa_pool_size(const dsa_pool *pool) { return pool ? (uint32_t)pool->ctx.size() : 0; }
const char *dsa_pool_last_error(const dsa_pool *pool) { return pool ? pool->err.c_str() : "null pool"; }

static void job_release(dsa_pool_job *job) {
  for (dsa_batch *b : job->batches) if (b) dsa_batch_free(b);
  delete job;
}
void dsa_pool_job_free(dsa_pool_job *job) {
  if (!job) return;
  dsa_pool *pool = job->pool;
  job_release(job);
  if (!pool) return;
  bool last;
  { std::lock_guard<std::mutex> g(pool->busy); last = pool->live_jobs.fetch_sub(1) == 1 && pool->doomed; }
  if (last) pool_really_destroy(pool);
}
uint32_t dsa_pool_job_chunks(const dsa_pool_job *job) { return job ? (uint32_t)job->batches.size() : 0; }

// Joins every thread it holds when it goes out of scope, whatever path leaves the function (a std::thread destroyed while
// joinable calls std::terminate).
struct JoinAll {
  std::vector<std::thread> threads;
  ~JoinAll() { for (std::thread &t : threads) if (t.joinable()) t.join(); }
};
struct JobOwner {       // frees the job (and the batches it holds) unless released
  dsa_pool_job *job;
  ~JobOwner() { if (job) job_release(job); }        // never counted among the pool's live jobs
  dsa_pool_job *release() { dsa_pool_job *j = job; job = nullptr; return j; }
};

// One decode at a time per pool (the contexts' streams and events are not shared between concurrent jobs): pool->busy.
static dsa_status pool_decode(dsa_pool *pool, uint32_t n, const uint8_t *const *streams, const size_t *lengths, dsa_pool_job **out) {
  std::lock_guard<std::mutex> one_job(pool->busy);
  JobOwner owner{new dsa_pool_job()};
  dsa_pool_job *job = owner.job;
  job->pool = pool;
  std::vector<uint32_t> order(n), begin(n + 1);
  // chunk_meshes == 0 at creation: sized per job -- one chunk per device while a device's share is at most 4096 meshes (a batch
  // below a few hundred meshes takes as long as one of a thousand: the per-mesh serial chains set its time), 4096-mesh chunks
  // pulled from the queue beyond that
  uint32_t chunk = pool->chunk;
  if (chunk == 0) {
    const uint64_t per_device = ((uint64_t)n + pool->ctx.size() - 1) / pool->ctx.size();
    chunk = (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(256, per_device));
  }
  const uint32_t chunks = dsa_pool_plan(n, lengths, chunk, order.data(), begin.data());
  job->batches.assign(chunks, nullptr);
  job->worker.assign(chunks, 0);
  job->where.resize(n);
  for (uint32_t c = 0; c < chunks; ++c)
    for (uint32_t k = begin[c]; k < begin[c + 1]; ++k) job->where[order[k]] = {c, k - begin[c]};
  std::atomic<uint32_t> next{0};
  std::atomic<int> failed{DSA_OK};
  std::vector<std::string> errs(pool->ctx.size());
  // A worker keeps two chunks in flight on its context: while the kernels of chunk k run, the streams of chunk k + 1 are parsed,
  // staged and uploaded (dsa_batch_create) and its kernels queued behind; then chunk k is collected.  No exception leaves a worker:
  // a failed host allocation or anything else becomes the job's status.
  auto work = [&](uint32_t w) noexcept {
    try {
      dsa_context *ctx = pool->ctx[w];
      std::vector<const uint8_t *> ptrs;
      std::vector<size_t> lens;
      dsa_batch *inflight = nullptr;
      auto collect = [&]() {
        if (!inflight) return;
        const dsa_status st = dsa_batch_wait(inflight);
        inflight = nullptr;
        if (st != DSA_OK) { errs[w] = dsa_last_error(ctx); int ok = DSA_OK; failed.compare_exchange_strong(ok, st); }
      };
      for (;;) {
        const uint32_t c = next.fetch_add(1, std::memory_order_relaxed);
        if (c >= chunks || failed.load(std::memory_order_relaxed) != DSA_OK) break;
        ptrs.clear(); lens.clear();
        for (uint32_t k = begin[c]; k < begin[c + 1]; ++k) { ptrs.push_back(streams[order[k]]); lens.push_back(lengths[order[k]]); }
        dsa_batch *b = nullptr;
        dsa_status st = dsa_batch_create(ctx, (uint32_t)ptrs.size(), ptrs.data(), lens.data(), &b);
        job->batches[c] = b;                 // owned by the job from here on, whatever happens next
        job->worker[c] = w;
        if (st == DSA_OK) st = dsa_batch_decode(b);
        if (st != DSA_OK) { errs[w] = dsa_last_error(ctx); int ok = DSA_OK; failed.compare_exchange_strong(ok, st); break; }
        collect();                           // the previous chunk, whose kernels ran while this one was prepared
        inflight = b;
      }
      collect();
    } catch (const std::bad_alloc &) {
      int ok = DSA_OK; failed.compare_exchange_strong(ok, DSA_ERR_OUT_OF_MEMORY);
    } catch (...) {
      int ok = DSA_OK; failed.compare_exchange_strong(ok, DSA_ERR_DEVICE);
    }
  };
  {
    JoinAll workers;
    workers.threads.reserve(pool->ctx.size());
    try {
      for (uint32_t w = 1; w < pool->ctx.size(); ++w) workers.threads.emplace_back(work, w);
    } catch (...) {                          // a thread could not be started: the ones that run finish the job between them
      if (workers.threads.empty() && pool->ctx.size() > 1) pool->err = "worker threads could not be started; decoding on one device";
    }
    work(0);                                 // the calling thread is worker 0
  }                                          // joined here
  if (failed.load() != DSA_OK) {
    pool->err = "a worker failed";
    for (const std::string &e : errs) if (!e.empty()) { pool->err = e; break; }
    return (dsa_status)failed.load();        // the owner frees the job and its batches
  }
  pool->live_jobs.fetch_add(1, std::memory_order_relaxed);
  *out = owner.release();
  return DSA_OK;
}
dsa_status dsa_pool_decode(dsa_pool *pool, uint32_t n, const uint8_t *const *streams, const size_t *lengths, dsa_pool_job **out) {
  if (!pool || !out || (n && (!streams || !lengths))) return DSA_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  try { return pool_decode(pool, n, streams, lengths, out); }
  catch (const std::bad_alloc &) { pool->err = "host allocation failed"; return DSA_ERR_OUT_OF_MEMORY; }
  catch (...) { pool->err = "unexpected failure inside the library"; return DSA_ERR_DEVICE; }
}

dsa_status dsa_pool_job_locate(const dsa_pool_job *job, uint32_t stream, const dsa_batch **batch, uint32_t *mesh, uint32_t *worker) {
  if (!job || stream >= job->where.size()) return DSA_ERR_INVALID_ARGUMENT;
  const dsa_pool_job::Where &w = job->where[stream];
  if (batch) *batch = job->batches[w.chunk];
  if (mesh) *mesh = w.index;
  if (worker) *worker = job->worker[w.chunk];
  return DSA_OK;
}

}  // extern "C"
