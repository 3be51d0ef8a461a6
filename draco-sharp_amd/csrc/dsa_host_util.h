// draco-sharp_amd/csrc/dsa_host_util.h
// Host-side plumbing shared by the decode boundary (dsa_api.hip), the encode boundary (dsa_encode.h) and the pool
// (dsa_pool.h): a thread fan-out that cannot leak a joinable thread or an exception, owners for device / pinned
// allocations, and the pinned staging buffers every host -> device upload goes through (a pageable source makes
// hipMemcpy stage through the runtime's own bounce buffer at a few GB/s; a pinned one is a single DMA at the link rate).
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <exception>
#include <thread>
#include <vector>

namespace hostutil {

// Threads a call may start: the cores this process may run on (a rank of an 8-process job sees its share when the launcher pins
// it), at most 32.  DSA_HOST_THREADS overrides (a launcher that does not pin can divide the box between its ranks with it).
inline uint32_t host_threads() {
  static const uint32_t n = []() -> uint32_t {
    if (const char *e = getenv("DSA_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return (uint32_t)std::min(v, 256); }
    cpu_set_t set;
    uint32_t cores = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = (uint32_t)CPU_COUNT(&set);
    if (cores == 0) cores = std::thread::hardware_concurrency();
    if (cores == 0) cores = 8;
    // a container's CPU quota (cgroup v2 cpu.max, v1 cfs_quota / cfs_period): more runnable threads than the quota pays for are
    // not slower by their share, they are stopped for the rest of every period
    auto quota = []() -> uint32_t {
      long long q = -1, per = 0;
      if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char first[32] = {0};
        if (fscanf(f, "%31s %lld", first, &per) == 2 && strcmp(first, "max") != 0) q = atoll(first);
        fclose(f);
      } else {
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
      }
      return q > 0 && per > 0 ? (uint32_t)((q + per - 1) / per) : 0u;
    }();
    if (quota) cores = std::min(cores, std::max(quota, 2u));
    return std::min<uint32_t>(cores, 32u);
  }();
  return n;
}

// fn(i) for i in [0, count) on up to host_threads() threads (the caller's included).  Every started thread is joined on every
// path; an exception thrown by fn is carried to the caller (the first one wins) after the join; a thread that cannot be started
// is not an error -- the others take its share (work is handed out through one counter).
template <class Fn>
inline void parallel_for(uint32_t count, Fn fn, uint32_t grain = 1) {
  if (count == 0) return;
  std::atomic<uint32_t> next{0};
  std::exception_ptr error;
  std::atomic<bool> failed{false};
  auto body = [&]() noexcept {
    for (;;) {
      const uint32_t i0 = next.fetch_add(grain, std::memory_order_relaxed);
      if (i0 >= count || failed.load(std::memory_order_relaxed)) return;
      const uint32_t i1 = std::min<uint32_t>(count, i0 + grain);
      try { for (uint32_t i = i0; i < i1; ++i) fn(i); }
      catch (...) { if (!failed.exchange(true)) error = std::current_exception(); return; }
    }
  };
  const uint32_t want = std::min<uint32_t>(host_threads(), (count + grain - 1) / grain);
  struct Joiner { std::vector<std::thread> t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner;
  try {
    joiner.t.reserve(want);
    for (uint32_t k = 1; k < want; ++k) joiner.t.emplace_back(body);
  } catch (...) {}                       // fewer helpers than wanted
  body();
  for (auto &x : joiner.t) x.join();
  if (failed.load()) std::rethrow_exception(error);
}

// Copies `bytes` with several threads when the piece is large enough to pay for them (a single core moves 5 - 10 GB/s).
inline void parallel_memcpy(void *dst, const void *src, size_t bytes) {
  const size_t piece = 4u << 20;
  if (bytes < 4 * piece) { memcpy(dst, src, bytes); return; }
  const uint32_t pieces = (uint32_t)((bytes + piece - 1) / piece);
  parallel_for(pieces, [&](uint32_t k) {
    const size_t at = (size_t)k * piece;
    memcpy((uint8_t *)dst + at, (const uint8_t *)src + at, std::min(piece, bytes - at));
  });
}

// Owners: released on every exit path of the function that holds them.
struct DeviceBuf {
  void *p = nullptr;
  DeviceBuf() = default;
  DeviceBuf(const DeviceBuf &) = delete;
  DeviceBuf &operator=(const DeviceBuf &) = delete;
  ~DeviceBuf() { reset(); }
  void reset() { if (p) (void)hipFree(p); p = nullptr; }
  hipError_t alloc(size_t bytes) { reset(); return hipMalloc(&p, bytes ? bytes : 256); }
  template <class T> T *as() const { return (T *)p; }
};
struct PinnedBuf {
  uint8_t *p = nullptr;
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  ~PinnedBuf() { reset(); }
  void reset() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  // at least `bytes`, contents not kept; grows by half so that a sequence of slowly growing batches does not re-pin every time
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    reset();
    const size_t want = bytes + bytes / 2 + 4096;
    hipError_t e = hipHostMalloc((void **)&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; e = hipHostMalloc((void **)&p, bytes, hipHostMallocDefault); if (e == hipSuccess) cap = bytes; else p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
};

// A pinned staging buffer with the event that says when the DMA reading it has finished: the next user waits for that
// event instead of for the stream.
struct Staging {
  PinnedBuf buf;
  hipEvent_t done = nullptr;
  bool pending = false;
  ~Staging() { if (done) (void)hipEventDestroy(done); }
  // ready for refilling with `bytes`
  hipError_t acquire(size_t bytes) {
    if (!done) { hipError_t e = hipEventCreateWithFlags(&done, hipEventDisableTiming); if (e != hipSuccess) return e; }
    if (pending) { hipError_t e = hipEventSynchronize(done); if (e != hipSuccess) return e; pending = false; }
    return buf.ensure(bytes);
  }
  hipError_t submitted(hipStream_t s) { hipError_t e = hipEventRecord(done, s); if (e == hipSuccess) pending = true; return e; }
};

// Whose turn it is on the host -> device link.  The chunks of an encode batch upload in two phases: A, what the long latency-bound
// stage of a chunk needs (the faces), and B, the rest (the attribute values, which only the kernels behind that stage read).
// Phases A go in the order of the chunks (the large ones of a tapered batch first) and before every waiting phase B, so that
// every chunk's long stage is under way before the link carries anything that is not needed yet.
struct UploadTurns {
  std::mutex m;
  std::condition_variable cv;
  bool busy = false;
  uint32_t next_a = 0;
  std::vector<char> a_done;               // per chunk: its phase A is over (or will never come)
  explicit UploadTurns(uint32_t chunks) : a_done(chunks, 0) {}
  int a_waiting = 0;
  void acquire_a(uint32_t chunk) {
    std::unique_lock<std::mutex> lk(m);
    ++a_waiting;
    cv.wait(lk, [&] { return !busy && next_a == chunk; });
    --a_waiting;
    busy = true;
  }
  void acquire_b() {                      // (a chunk in front of a waiting phase A is on its way to its own phase A, never waiting here)
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return !busy && a_waiting == 0; });
    busy = true;
  }
  void finish_a(uint32_t chunk, bool held) {          // also for a chunk that never uploads (held = false)
    {
      std::lock_guard<std::mutex> lk(m);
      if (held) busy = false;
      if (chunk < a_done.size()) a_done[chunk] = 1;
      while (next_a < a_done.size() && a_done[next_a]) ++next_a;
    }
    cv.notify_all();
  }
  void release_b() { { std::lock_guard<std::mutex> lk(m); busy = false; } cv.notify_all(); }
};
struct TurnGuard {                       // a turn is given back, and a phase A that never came is struck off, on every exit path
  UploadTurns *t;
  uint32_t chunk;
  int held = 0;                          // 1: phase A, 2: phase B
  bool a_over = false;
  TurnGuard(UploadTurns *turns, uint32_t chunk_index) : t(turns), chunk(chunk_index) {}
  TurnGuard(const TurnGuard &) = delete;
  TurnGuard &operator=(const TurnGuard &) = delete;
  void acquire_a() { if (t && !held && !a_over) { t->acquire_a(chunk); held = 1; } }
  void acquire_b() { if (t && !held) { if (!a_over) { t->finish_a(chunk, false); a_over = true; } t->acquire_b(); held = 2; } }
  void release() {
    if (!t) return;
    if (held == 1) { t->finish_a(chunk, true); a_over = true; }
    else if (held == 2) t->release_b();
    held = 0;
  }
  ~TurnGuard() { release(); if (t && !a_over) t->finish_a(chunk, false); }
};

}  // namespace hostutil
