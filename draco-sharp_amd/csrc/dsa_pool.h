// draco-sharp_amd/csrc/dsa_pool.h
// Multi-GPU submit behind the C-ABI (include/draco_mi355x.h, dsa_pool_*): one context + one worker thread per listed
// device, a job's streams sorted by compressed length and handed out chunk by chunk through one atomic counter
// (SURVEY.md section 8e).  Meshes are independent (DracoDecoder.cs:19-42 creates everything per call), so there is no
// collective and no peer traffic; a chunk is an ordinary dsa_batch on the worker's context.  Included by dsa_api.hip.
#pragma once
#include <algorithm>
#include <atomic>
#include <numeric>
#include <thread>

struct dsa_pool {
  std::vector<dsa_context *> ctx;
  uint32_t chunk = 256;
  std::string err;
};

struct dsa_pool_job {
  struct Where { uint32_t chunk, index; };
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
  p->chunk = chunk_meshes ? std::min<uint32_t>(chunk_meshes, 65535u) : 256u;
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

void dsa_pool_destroy(dsa_pool *pool) {
  if (!pool) return;
  for (dsa_context *c : pool->ctx) dsa_context_destroy(c);
  delete pool;
}
uint32_t dsa_pool_size(const dsa_pool *pool) { return pool ? (uint32_t)pool->ctx.size() : 0; }
const char *dsa_pool_last_error(const dsa_pool *pool) { return pool ? pool->err.c_str() : "null pool"; }

void dsa_pool_job_free(dsa_pool_job *job) {
  if (!job) return;
  for (dsa_batch *b : job->batches) if (b) dsa_batch_free(b);
  delete job;
}
uint32_t dsa_pool_job_chunks(const dsa_pool_job *job) { return job ? (uint32_t)job->batches.size() : 0; }

static dsa_status pool_decode(dsa_pool *pool, uint32_t n, const uint8_t *const *streams, const size_t *lengths, dsa_pool_job **out) {
  dsa_pool_job *job = new dsa_pool_job();
  std::vector<uint32_t> order(n), begin(n + 1);
  const uint32_t chunks = dsa_pool_plan(n, lengths, pool->chunk, order.data(), begin.data());
  job->batches.assign(chunks, nullptr);
  job->worker.assign(chunks, 0);
  job->where.resize(n);
  for (uint32_t c = 0; c < chunks; ++c)
    for (uint32_t k = begin[c]; k < begin[c + 1]; ++k) job->where[order[k]] = {c, k - begin[c]};
  std::atomic<uint32_t> next{0};
  std::atomic<int> failed{DSA_OK};
  std::vector<std::string> errs(pool->ctx.size());
  auto work = [&](uint32_t w) {
    dsa_context *ctx = pool->ctx[w];
    std::vector<const uint8_t *> ptrs;
    std::vector<size_t> lens;
    for (;;) {
      const uint32_t c = next.fetch_add(1, std::memory_order_relaxed);
      if (c >= chunks || failed.load(std::memory_order_relaxed) != DSA_OK) break;
      ptrs.clear(); lens.clear();
      for (uint32_t k = begin[c]; k < begin[c + 1]; ++k) { ptrs.push_back(streams[order[k]]); lens.push_back(lengths[order[k]]); }
      dsa_batch *b = nullptr;
      dsa_status st = dsa_batch_create(ctx, (uint32_t)ptrs.size(), ptrs.data(), lens.data(), &b);
      if (st == DSA_OK) st = dsa_batch_decode(b);
      if (st == DSA_OK) st = dsa_batch_wait(b);
      job->batches[c] = b;
      job->worker[c] = w;
      if (st != DSA_OK) { errs[w] = dsa_last_error(ctx); failed.store(st, std::memory_order_relaxed); break; }
    }
  };
  std::vector<std::thread> threads;
  for (uint32_t w = 1; w < pool->ctx.size(); ++w) threads.emplace_back(work, w);
  work(0);                                   // the calling thread is worker 0
  for (std::thread &t : threads) t.join();
  if (failed.load() != DSA_OK) {
    for (const std::string &e : errs) if (!e.empty()) { pool->err = e; break; }
    dsa_pool_job_free(job);
    return (dsa_status)failed.load();
  }
  *out = job;
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
