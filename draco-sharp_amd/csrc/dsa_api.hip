// draco-sharp_amd/csrc/dsa_api.hip
// Host side of the C-ABI declared in include/draco_mi355x.h: batch construction
// (header pre-parse for arena sizing, upload), kernel launches on the context's
// HIP stream, result queries and copy-out.  No torch types, no CPU decode path:
// every byte of geometry is produced by the kernels in dsa_kernels.h.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/draco_mi355x.h"
#include "dsa_kernels.h"
#include "dsa_seams.h"
#include "dsa_general.h"
#include "dsa_host_parse.h"
#include "dsa_host_util.h"

namespace {

enum { STG_LOCATE = 0, STG_CONNECTIVITY, STG_TRAVERSE, STG_OPERANDS, STG_SYMBOLS, STG_PREDICT, STG_FINALIZE, STG_TOTAL };
static const char *kStageNames[DSA_NUM_STAGES] = {"locate", "connectivity", "traverse", "para_operands",
                                                 "symbols", "predict", "finalize", "total"};
// Kernels timed one by one (dsa_batch_kernel_times): an event pair around each on the stream it is launched on, so that a duration
// here is a row of `rocprofv3 --kernel-trace --stats` (first launch of that kernel in the decode where a kernel is launched twice).
enum { KT_CHAIN = 0, KT_CONNECTIVITY, KT_TRAVERSE, KT_SYMBOLS_EARLY, KT_SYMBOLS_LATE, KT_OCT_STREAMS, KT_PREDICT_WRAP_EARLY, KT_PREDICT_WRAP_LATE, KT_FACES,
       KT_SEAM_TABLES, KT_TRAVERSE_ATT, KT_TEXCOORDS, KT_COUNT };
static const char *kKernelNames[KT_COUNT] = {"k_chain", "k_connectivity", "k_traverse", "k_symbols_reg[early]", "k_symbols_reg[late]", "k_predict_oct_streams",
                                             "k_predict_wrap[early]", "k_predict_wrap[late]", "k_faces", "k_seam_tables", "k_traverse_att", "k_texcoords"};


}  // namespace

// One lane of the encode pipeline (dsa_encode.h): a stream of its own and two pinned staging buffers, so that several chunks of a
// batch are in flight at once -- the uploads of one beside the kernels of another beside the stream layout of a third.
struct EncLane {
  int device = 0;
  hipStream_t st = nullptr;
  hostutil::Staging stage[2];
  int next = 0;
  hostutil::UploadTurns *upload_turn = nullptr;     // of the batch being coded: the chunks' uploads go over the link one after the other
  uint32_t upload_chunk = 0;                         // the chunk this lane is coding: its place in the order of the uploads
  // the walks of a chunk (k_enc_connectivity: 0.1 - 0.2 s of memory latency) on a stream of their own, so that the chunk's
  // attribute values upload and quantise behind them on `st`
  hipStream_t walk_st = nullptr;
  hipEvent_t tables_done = nullptr, walk_done = nullptr;
  // device memory of the lane, grown on demand and kept: hipMalloc / hipFree wait for every stream of the device, which would
  // put the chunks of a batch back in single file
  struct Buf {
    void *p = nullptr; uint64_t cap = 0;
    hipError_t ensure(uint64_t bytes) {
      if (bytes <= cap) return hipSuccess;
      if (p) (void)hipFree(p);
      p = nullptr; cap = 0;
      const uint64_t want = bytes + bytes / 4;
      hipError_t e = hipMalloc(&p, want);
      if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; e = hipMalloc(&p, bytes); if (e == hipSuccess) cap = bytes; else p = nullptr; return e; }
      cap = want;
      return hipSuccess;
    }
    ~Buf() { if (p) (void)hipFree(p); }
  } arena, streams, conns, packed, items;
  ~EncLane() {
    if (walk_st) { (void)hipStreamSynchronize(walk_st); (void)hipStreamDestroy(walk_st); }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (tables_done) (void)hipEventDestroy(tables_done);
    if (walk_done) (void)hipEventDestroy(walk_done);
  }
};

// The streams one decode runs on, and the events that order them.  A context has two sets, used in turn (unless the caller gave
// the context a stream of its own: then only the first): a decode queued while the previous one is still running goes to the other
// set, so that its chain and entropy decode start beside the previous batch's issue-light tail (parallelogram prediction, faces,
// dequantisation) instead of behind it on the same in-order streams.
struct StreamSet {
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;     // symbol decode runs here, concurrently with connectivity + traversal
  hipStream_t stream3 = nullptr;     // connectivity validation (link symmetry, seam streams)
  hipStream_t stream4 = nullptr;     // early attributes: symbols, prediction, dequantisation (dispatch priority)
  hipStream_t stream5 = nullptr;     // the serial bit streams of the late prediction schemes (flip bits, orientation bits): wanted late, so
                                     // they must not stand in front of anything on another stream
  hipEvent_t ev_join3 = nullptr, ev_trav = nullptr, ev_maps = nullptr, ev_early = nullptr, ev_flips = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_conn = nullptr;
  hipEvent_t ev_seambits = nullptr, ev_tables = nullptr, ev_att = nullptr;      // fast seam path: seam bits decoded, seam tables built, attribute traversals done
  hipEvent_t ev_pred = nullptr;                                                 // fast seam path: corrections of the attributes in front of the seam tables ready
  bool own_stream = false;
  hipError_t create(hipStream_t user, int least, int greatest) {
    // Dispatch priorities (numerically lower = higher): the stream of the per-mesh chain above the stream of the early attributes
    // above the late symbols, which fill whatever the others leave.
    hipError_t e = hipSuccess;
    if (user) { stream = user; own_stream = false; }
    else { e = hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest); own_stream = e == hipSuccess; }
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&stream4, hipStreamNonBlocking, (least + greatest) / 2);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, least);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream3, hipStreamNonBlocking);
    // (low priority, beside stream2: the streams of one priority class share four hardware queues, and the default class already
    // holds stream3, stream4 and the two copy streams -- a fifth there waits behind one of them)
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&stream5, hipStreamNonBlocking, least);
    for (hipEvent_t *ev : {&ev_join3, &ev_trav, &ev_maps, &ev_flips, &ev_early, &ev_fork, &ev_join, &ev_conn, &ev_seambits, &ev_tables, &ev_att, &ev_pred})
      if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    return e;
  }
  void destroy() {
    for (hipEvent_t *ev : {&ev_join3, &ev_trav, &ev_maps, &ev_flips, &ev_early, &ev_fork, &ev_join, &ev_conn, &ev_seambits, &ev_tables, &ev_att, &ev_pred}) { if (*ev) (void)hipEventDestroy(*ev); *ev = nullptr; }
    for (hipStream_t *s : {&stream2, &stream3, &stream4, &stream5}) { if (*s) (void)hipStreamDestroy(*s); *s = nullptr; }
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }
};

struct dsa_context {
  int device = 0;
  StreamSet sets[2];
  int num_sets = 0;                  // sets this context may use (1 on a caller's stream); the second is created when first wanted
  bool set1_made = false;
  int prio_least = 0, prio_greatest = 0;
  int last_set = 0;                  // the set of the most recent decode, and that decode's end
  hipEvent_t last_done = nullptr;
  hipStream_t up = nullptr;          // host -> device: compressed streams of the next batch, beside the kernels of this one
  hipStream_t down = nullptr;        // device -> host: the output block and the mesh descriptors of the previous one
  bool profiling = false;
  std::string err;
  // Uploads go through pinned staging (two buffers, used in turn: the DMA of one batch reads its buffer while the next batch is
  // staged into the other).
  hostutil::Staging stage[2];
  int stage_next = 0;
  // Arenas and pinned mirrors of freed batches are kept for the next batch of the context: hipMalloc / hipFree of tens of GB and
  // pinning GBs of host memory cost as much as the decode itself.  Three of each: two batches in flight + one being built.
  struct Spare { uint8_t *p; uint64_t bytes; };
  std::vector<Spare> spare_arenas, spare_mirrors, spare_descs, spare_packed;   // spare_descs: pinned landing zones of the mesh descriptors; spare_packed: packed blocks of compact downloads
  static constexpr size_t kSpares = 3;
  // batches point at their context: a context destroyed first lives on until its last batch is freed
  std::atomic<int> live_batches{0};
  std::atomic<bool> doomed{false};
  // The caches above, stage_next and next_set are touched by dsa_batch_create / _download / _free, which the pool calls from its
  // worker threads and from whichever thread frees a job while the next decode runs (dsa_pool.h): one lock for all of them.
  std::mutex mu;
  std::vector<std::unique_ptr<EncLane>> enc_lanes;    // kept between dsa_encode_batch calls (pinning their staging costs more than a small batch)
  // k_register_gate holds the late symbol kernel back by asking for more registers than a SIMD has left beside four chain waves and
  // three entropy decoders.  That is occupancy arithmetic on THIS build's register counts: checked once per context (gate_check),
  // and the gate is left out -- with a note here -- when the numbers no longer add up.
  bool gate_ok = false;
  std::string gate_note;
};

struct dsa_batch {
  dsa_context *ctx = nullptr;
  uint32_t n = 0;
  std::vector<MeshLayout> layouts;
  std::vector<HostMesh> host;
  std::vector<MeshDesc> descs;       // copied back by dsa_batch_wait
  uint8_t *arena = nullptr;          // [layouts | globals | streams + slack | descs | per-mesh scratch | table pool | output block]
  uint64_t arena_bytes = 0, arena_cap = 0;
  MeshLayout *d_layouts = nullptr;   // inside the arena
  MeshDesc *d_descs = nullptr;
  BatchGlobals *d_globals = nullptr;
  BatchGlobals globals = {};
  uint64_t out_base = 0, out_bytes = 0;   // the output block: faces, attribute values and point maps of every mesh, in three
  uint64_t out_values = 0, out_values_bytes = 0;   // sub-blocks (offsets of the values sub-block inside the output block)
  // Compact download: the values sub-block as it is + a packed block made on the device (faces as uint16 where every point id
  // fits, one point map per distinct map).  Layout per mesh from the header's counts; the device copy of it and the packed block
  // are made at the first compact download.
  std::vector<CompactMesh> compact;       // host copy; the device's is part of the arena's uploaded head
  CompactMesh *d_compact = nullptr;
  uint64_t packed_bytes = 0;
  uint8_t *d_packed = nullptr;            // the packed block, from the context's cache at the first compact download
  uint64_t d_packed_cap = 0;
  bool mirror_compact = false;
  uint32_t max_faces = 0, max_vertices = 0, max_atts = 0, max_att_data = 0;
  uint64_t sum_vertices = 0;
  bool any_general = false, any_valence = false, any_seamed = false, any_multipara = false;
  bool decoded = false, collected = false;
  hipEvent_t ev[DSA_NUM_STAGES + 1] = {};
  hipEvent_t ev_sym[2] = {};
  hipEvent_t ev_k[KT_COUNT][2] = {};
  bool k_timed[KT_COUNT] = {};
  bool have_events = false;
  float stage_ms[DSA_NUM_STAGES] = {};
  float kernel_ms[KT_COUNT] = {};
  // per-batch ordering between the copy streams and the kernels (a second batch may be queued on the same context meanwhile,
  // so nothing here waits for a whole stream)
  hipEvent_t ev_uploaded = nullptr, ev_done = nullptr, ev_descs = nullptr, ev_down = nullptr;
  MeshDesc *descs_pin = nullptr;     // pinned landing zone of the descriptors
  uint64_t descs_pin_bytes = 0;
  // host copy of the output block: a library-owned pinned mirror or the caller's buffer
  uint8_t *mirror = nullptr;
  uint64_t mirror_bytes = 0;
  bool mirror_owned = false, download_queued = false, downloaded = false;
  // meshes the fast kernels handed back (DSA_SITE_RETRY_GENERAL): decoded again through the general path in a batch
  // of their own by dsa_batch_wait; every per-mesh accessor follows retry_index
  bool all_general = false;
  dsa_batch *retry = nullptr;
  std::vector<int32_t> retry_index;
};

namespace {

dsa_status set_err(dsa_context *ctx, dsa_status st, const char *fmt, ...) {
  if (ctx) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctx->err = buf;
  }
  return st;
}
#define HIP_TRY(ctx, call)                                                                         \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return set_err((ctx), e_ == hipErrorOutOfMemory ? DSA_ERR_OUT_OF_MEMORY : DSA_ERR_DEVICE, \
                                         "%s failed: %s", #call, hipGetErrorString(e_));          \
  } while (0)

// Arena / mirror caches of a context (see dsa_context).
uint8_t *take_spare(dsa_context *ctx, std::vector<dsa_context::Spare> &spares, uint64_t need, uint64_t *got) {
  std::lock_guard<std::mutex> g(ctx->mu);
  int best = -1;
  for (size_t k = 0; k < spares.size(); ++k)
    if (spares[k].bytes >= need && (best < 0 || spares[k].bytes < spares[(size_t)best].bytes)) best = (int)k;
  if (best < 0) return nullptr;
  uint8_t *p = spares[(size_t)best].p;
  *got = spares[(size_t)best].bytes;
  spares.erase(spares.begin() + best);
  return p;
}
template <class FreeFn>
void give_spare(dsa_context *ctx, std::vector<dsa_context::Spare> &spares, uint8_t *p, uint64_t bytes, FreeFn release) {
  if (!p) return;
  std::lock_guard<std::mutex> g(ctx->mu);
  try { spares.push_back({p, bytes}); } catch (...) { release(p); return; }
  while (spares.size() > dsa_context::kSpares) {       // drop the smallest
    size_t s = 0;
    for (size_t k = 1; k < spares.size(); ++k) if (spares[k].bytes < spares[s].bytes) s = k;
    release(spares[s].p);
    spares.erase(spares.begin() + (ptrdiff_t)s);
  }
}
void drop_spares(dsa_context *ctx, std::vector<dsa_context::Spare> &spares, bool pinned) {
  std::lock_guard<std::mutex> g(ctx->mu);
  for (auto &sp : spares) { if (pinned) (void)hipHostFree(sp.p); else (void)hipFree(sp.p); }
  spares.clear();
}
hipError_t arena_alloc(dsa_context *ctx, uint64_t need, uint8_t **out, uint64_t *cap) {
  if (uint8_t *p = take_spare(ctx, ctx->spare_arenas, need, cap)) { *out = p; return hipSuccess; }
  hipError_t e = hipMalloc((void **)out, need);
  if (e != hipSuccess) {     // what the context keeps idle may be what is in the way: spare arenas, then the encoder's lanes
    (void)hipGetLastError();
    drop_spares(ctx, ctx->spare_arenas, false);
    e = hipMalloc((void **)out, need);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      { std::lock_guard<std::mutex> g(ctx->mu); ctx->enc_lanes.clear(); }
      e = hipMalloc((void **)out, need);
    }
  }
  *cap = need;
  return e;
}

dsa_status build_batch(dsa_context *ctx, uint32_t n, const uint8_t *const *streams, const size_t *lengths, dsa_batch **out, bool all_general = false) {
  if (!ctx || !out || (n && (!streams || !lengths))) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "null argument");
  if (n > 65535) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "batch too large (max 65535 meshes)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dsa_batch *b = new (std::nothrow) dsa_batch();
  if (!b) return set_err(ctx, DSA_ERR_OUT_OF_MEMORY, "host allocation failed");
  struct Guard { dsa_batch *b; ~Guard() { if (b) dsa_batch_free(b); } } guard{b};     // freed on every failing path (exceptions included)
  b->ctx = ctx;
  ++ctx->live_batches;
  b->n = n;
  b->all_general = all_general;
  b->layouts.resize(n);
  b->host.resize(n);
  b->descs.resize(n);
  // ---- arena layout: [layouts | globals | streams | slack | descs | per-mesh scratch | table pool | output block]
  const uint64_t off_layouts = 0, off_globals = align_up(sizeof(MeshLayout) * (uint64_t)n, 256);
  const uint64_t off_compact = align_up(off_globals + sizeof(BatchGlobals), 256);
  uint64_t cur = align_up(off_compact + sizeof(CompactMesh) * (uint64_t)n, 256);
  for (uint32_t i = 0; i < n; ++i) {
    if (lengths[i] > 0xFFFFFF00u) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "stream %u longer than 4 GiB", i);
    MeshLayout &L = b->layouts[i];
    memset(&L, 0, sizeof(L));
    L.stream = cur;
    L.stream_len = (uint32_t)lengths[i];
    cur = align_up(cur + lengths[i], 16);
  }
  cur = align_up(cur + 1024, 256);   // window over-read slack behind the last stream
  const uint64_t upload_bytes = cur;  // everything the host provides, one transfer
  const uint64_t off_descs = cur;
  cur = align_up(cur + sizeof(MeshDesc) * (uint64_t)(n ? n : 1), 256);
  const uint64_t scratch_begin = cur;
  hostutil::parallel_for(n, [&](uint32_t i) { host_parse(streams[i], lengths[i], b->host[i], all_general); }, 16);
  // Regions of every mesh behind the streams.  When the arena does not fit the device, the mesh with the largest
  // claim is set aside (per-mesh DSA_ERR_OUT_OF_MEMORY: a header may claim far more elements than its stream can
  // carry) and the rest is laid out again, so that one stream cannot take the batch down.
  hipError_t e = hipSuccess;
  std::vector<uint64_t> claim(n, 0);
  for (;;) {
    cur = scratch_begin;
    OutCursors ocur;
    b->max_faces = b->max_vertices = b->max_atts = b->max_att_data = 0;
    b->sum_vertices = 0;
    b->any_general = false;
    b->any_valence = false;
    b->any_seamed = false; b->any_multipara = false;
    for (uint32_t i = 0; i < n; ++i) {
      HostMesh &h = b->host[i];
      MeshLayout &L = b->layouts[i];
      if (h.status != 0) { h.faces = 0; h.enc_vertices = 0; h.split_symbols = 0; h.splits = 0; h.atts.clear(); h.general = false; }
      const uint64_t F = h.faces, V = (uint64_t)h.enc_vertices + h.split_symbols, before = cur, obefore = ocur.faces + ocur.values + ocur.maps;
      const uint64_t stream_off = L.stream;
      const uint32_t stream_len = L.stream_len;
      memset(&L, 0, sizeof(L));
      L.stream = stream_off; L.stream_len = stream_len;
      cur = layout_mesh(h, lengths[i], L, cur, 16, nullptr, &ocur);
      claim[i] = (cur - before) + (ocur.faces + ocur.values + ocur.maps - obefore);
      b->max_faces = std::max<uint32_t>(b->max_faces, (uint32_t)F);
      b->max_vertices = std::max<uint32_t>(b->max_vertices, (uint32_t)V);
      b->sum_vertices += V;
      b->max_atts = std::max<uint32_t>(b->max_atts, (uint32_t)h.atts.size());
      b->max_att_data = std::max<uint32_t>(b->max_att_data, h.num_att_data);
      b->any_general = b->any_general || h.general;
      b->any_valence = b->any_valence || (h.valence && !h.general);
      b->any_seamed = b->any_seamed || (h.seamed && !h.general);
      b->any_multipara = b->any_multipara || L.mp_att != 0;
    }
    // pool for the cumulative tables of large-alphabet streams (bump-allocated by k_locate)
    {
      uint64_t streams_total = 0;
      for (uint32_t i = 0; i < n; ++i) streams_total += b->layouts[i].cap_attributes;
      b->globals.pool = cur;
      b->globals.pool_bytes = (64ull << 20) + 8192ull * streams_total;
      cur = align_up(cur + b->globals.pool_bytes, 256);
    }
    // the output block behind everything else; the offsets layout_mesh handed out were relative to it
    b->out_base = align_up(cur, 4096);
    b->out_values = align_up(ocur.faces, 4096);
    b->out_values_bytes = ocur.values;
    const uint64_t out_maps = align_up(b->out_values + ocur.values, 4096);
    b->out_bytes = out_maps + ocur.maps;
    for (uint32_t i = 0; i < n; ++i) {
      MeshLayout &L = b->layouts[i];
      L.faces += b->out_base;
      for (uint32_t a = 0; a < L.cap_attributes; ++a) { L.out[a] += b->out_base + b->out_values; L.map[a] += b->out_base + out_maps; }
    }
    b->arena_bytes = b->out_base + align_up(b->out_bytes, 256) + 256;
    e = arena_alloc(ctx, b->arena_bytes, &b->arena, &b->arena_cap);
    if (e == hipSuccess) break;
    (void)hipGetLastError();
    b->arena = nullptr;
    uint32_t worst = 0;
    for (uint32_t i = 1; i < n; ++i) if (claim[i] > claim[worst]) worst = i;
    if (n == 0 || b->host[worst].status != 0 || claim[worst] <= (64ull << 20)) break;   // nothing left to set aside: the batch itself is too large
    b->host[worst].status = DSA_ERR_OUT_OF_MEMORY;
  }
  if (e != hipSuccess) return set_err(ctx, DSA_ERR_OUT_OF_MEMORY, "hipMalloc of %llu-byte arena failed: %s", (unsigned long long)b->arena_bytes, hipGetErrorString(e));
  b->d_layouts = (MeshLayout *)(b->arena + off_layouts);
  b->d_globals = (BatchGlobals *)(b->arena + off_globals);
  b->d_compact = (CompactMesh *)(b->arena + off_compact);
  {   // the packed block of a compact download, from the header's counts
    b->compact.assign(n, CompactMesh());
    uint64_t pcur = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const HostMesh &h = b->host[i];
      const MeshLayout &L = b->layouts[i];
      CompactMesh &c = b->compact[i];
      c.u16 = L.cap_points <= 65536u ? 1u : 0u; c.pad = 0;
      c.faces = pcur;
      pcur = align_up(pcur + (uint64_t)L.cap_faces * (c.u16 ? 6 : 12), 64);
      const bool identity = h.faces == 0 && !h.general;          // point clouds: linear sequencing, point i = entry i
      int key_of[DSA_MAX_ATT];
      for (uint32_t a = 0; a < DSA_MAX_ATT; ++a) c.map[a] = ~0ull;
      for (uint32_t a = 0; a < L.cap_attributes && a < DSA_MAX_ATT; ++a) {
        if (identity) continue;
        // attributes decoded in one traversal order share their map: all vertex attributes of a fast-path mesh, the attributes of
        // one corner-attribute decoder; a general-path mesh keeps one map per attribute
        key_of[a] = h.general ? 1000 + (int)a : (h.atts[a].corner ? 1 + (int)h.atts[a].dec : 0);
        uint32_t rep = a;
        for (uint32_t k = 0; k < a; ++k) if (key_of[k] == key_of[a]) { rep = k; break; }
        if (rep != a) { c.map[a] = c.map[rep]; continue; }
        c.map[a] = pcur;
        pcur = align_up(pcur + 4ull * L.cap_points, 64);
      }
    }
    b->packed_bytes = pcur;
  }
  b->d_descs = (MeshDesc *)(b->arena + off_descs);
  HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_uploaded, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_descs, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_down, hipEventDisableTiming));
  {   // pinned landing zone of the descriptors, from the context's cache (hipHostMalloc / hipHostFree wait for the device: with
      // another batch in flight they would serialise the pipeline)
    const uint64_t need = sizeof(MeshDesc) * (uint64_t)(n ? n : 1);
    uint8_t *p = take_spare(ctx, ctx->spare_descs, need, &b->descs_pin_bytes);
    if (!p) { HIP_TRY(ctx, hipHostMalloc((void **)&p, need + need / 2, hipHostMallocDefault)); b->descs_pin_bytes = need + need / 2; }
    b->descs_pin = (MeshDesc *)p;
  }
  // ---- upload: layouts, globals and all streams staged in pinned memory (host threads), one DMA on the upload stream.  The
  // caller's buffers are not referenced once this function returns; the kernels wait for ev_uploaded, the host does not.
  {
    int turn;
    { std::lock_guard<std::mutex> g(ctx->mu); turn = ctx->stage_next; ctx->stage_next ^= 1; }
    hostutil::Staging &stg = ctx->stage[turn];
    HIP_TRY(ctx, stg.acquire(upload_bytes));
    uint8_t *h = stg.buf.p;
    if (n) memcpy(h + off_layouts, b->layouts.data(), sizeof(MeshLayout) * (size_t)n);
    memset(h + sizeof(MeshLayout) * (size_t)n, 0, (size_t)(off_globals - sizeof(MeshLayout) * (uint64_t)n));
    memcpy(h + off_globals, &b->globals, sizeof(BatchGlobals));
    const uint64_t first = n ? b->layouts[0].stream : upload_bytes;
    memset(h + off_globals + sizeof(BatchGlobals), 0, (size_t)(first - off_globals - sizeof(BatchGlobals)));
    if (n) memcpy(h + off_compact, b->compact.data(), sizeof(CompactMesh) * (size_t)n);
    hostutil::parallel_for(n, [&](uint32_t i) {       // stream i and the zero gap up to the next one
      const uint64_t at = b->layouts[i].stream, end = at + lengths[i], next = i + 1 < n ? b->layouts[i + 1].stream : upload_bytes;
      if (lengths[i]) memcpy(h + at, streams[i], lengths[i]);
      memset(h + end, 0, (size_t)(next - end));
    }, 8);
    HIP_TRY(ctx, hipMemcpyAsync(b->arena, h, upload_bytes, hipMemcpyHostToDevice, ctx->up));
    HIP_TRY(ctx, stg.submitted(ctx->up));
    HIP_TRY(ctx, hipEventRecord(b->ev_uploaded, ctx->up));
  }
  guard.b = nullptr;
  *out = b;
  return DSA_OK;
}

// The register arithmetic k_register_gate rests on (dsa_kernels.h): a SIMD's 512 vector registers hold four
// k_chain waves and three k_symbols_reg waves, and what is left is less than one k_register_gate wave asks for -- so the gate's idle
// waves find room only when decoders leave.  Allocation granularity 8 registers.
void gate_check(dsa_context *c) {
  hipFuncAttributes chain{}, reg{}, gate{};
  if (hipFuncGetAttributes(&chain, (const void *)dsa::k_chain) != hipSuccess || hipFuncGetAttributes(&reg, (const void *)dsa::k_symbols_reg) != hipSuccess ||
      hipFuncGetAttributes(&gate, (const void *)dsa::k_register_gate) != hipSuccess) {
    (void)hipGetLastError();
    c->gate_ok = false; c->gate_note = "k_register_gate left out: kernel attributes not available";
    return;
  }
  auto up8 = [](int r) { return (r + 7) / 8 * 8; };
  const int rc = up8(chain.numRegs), rr = up8(reg.numRegs), rg = up8(gate.numRegs), simd = 512;
  const int left = simd - 4 * rc - 3 * rr;                 // beside four chain waves and three decoders
  // the gate must not fit there, nor a fourth decoder (else the machine is not the one the schedule was measured on), but must
  // fit once the decoders have left
  c->gate_ok = left >= 0 && rg > left && rr > left && rg <= left + 3 * rr;
  char buf[256];
  snprintf(buf, sizeof(buf), "k_register_gate %s: registers k_chain %d, k_symbols_reg %d, k_register_gate %d; %d left beside 4 + 3 waves of a SIMD",
           c->gate_ok ? "in use" : "left out", chain.numRegs, reg.numRegs, gate.numRegs, left);
  c->gate_note = buf;
}

}  // namespace

extern "C" {

int dsa_abi_version(void) { return DSA_ABI_VERSION; }

int dsa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

dsa_status dsa_context_create(int device, void *stream, dsa_context **out) {
  if (!out) return DSA_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return DSA_ERR_DEVICE;
  dsa_context *c = new (std::nothrow) dsa_context();
  if (!c) return DSA_ERR_OUT_OF_MEMORY;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return DSA_ERR_DEVICE; }
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  // a caller's stream: every decode runs in its order; otherwise two stream sets used in turn (see StreamSet)
  c->num_sets = stream ? 1 : 2;
  c->prio_least = least; c->prio_greatest = greatest;
  if (c->sets[0].create((hipStream_t)stream, least, greatest) != hipSuccess) { dsa_context_destroy(c); return DSA_ERR_DEVICE; }
  if (hipStreamCreateWithFlags(&c->up, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->down, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming) != hipSuccess) { dsa_context_destroy(c); return DSA_ERR_DEVICE; }
  gate_check(c);
  *out = c;
  return DSA_OK;
}

void dsa_context_destroy(dsa_context *ctx) {
  if (!ctx) return;
  {   // dsa_batch_free of the last batch comes back here
    std::lock_guard<std::mutex> g(ctx->mu);
    if (ctx->live_batches.load() > 0) { ctx->doomed = true; return; }
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->up) { (void)hipStreamSynchronize(ctx->up); (void)hipStreamDestroy(ctx->up); }
  if (ctx->down) { (void)hipStreamSynchronize(ctx->down); (void)hipStreamDestroy(ctx->down); }
  drop_spares(ctx, ctx->spare_arenas, false);
  drop_spares(ctx, ctx->spare_packed, false);
  drop_spares(ctx, ctx->spare_mirrors, true);
  drop_spares(ctx, ctx->spare_descs, true);
  for (StreamSet &set : ctx->sets) set.destroy();
  if (ctx->last_done) (void)hipEventDestroy(ctx->last_done);
  delete ctx;
}

const char *dsa_last_error(const dsa_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

dsa_status dsa_context_set_profiling(dsa_context *ctx, int enabled) {
  if (!ctx) return DSA_ERR_INVALID_ARGUMENT;
  ctx->profiling = enabled != 0;
  return DSA_OK;
}

// No exception crosses the C boundary: a failed host allocation is DSA_ERR_OUT_OF_MEMORY, anything else DSA_ERR_DEVICE.
#define DSA_GUARD(ctx, expr)                                                                              \
  try { return (expr); }                                                                                   \
  catch (const std::bad_alloc &) { return set_err((ctx), DSA_ERR_OUT_OF_MEMORY, "host allocation failed"); } \
  catch (...) { return set_err((ctx), DSA_ERR_DEVICE, "unexpected failure inside the library"); }

dsa_status dsa_batch_create(dsa_context *ctx, uint32_t n, const uint8_t *const *streams, const size_t *lengths, dsa_batch **out) {
  DSA_GUARD(ctx, build_batch(ctx, n, streams, lengths, out));
}

static dsa_status create_packed(dsa_context *ctx, uint32_t n, const uint8_t *blob, const uint64_t *offsets, dsa_batch **out) {
  std::vector<const uint8_t *> ptrs(n);
  std::vector<size_t> lens(n);
  for (uint32_t i = 0; i < n; ++i) {
    if (offsets[i + 1] < offsets[i]) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    ptrs[i] = blob + offsets[i];
    lens[i] = (size_t)(offsets[i + 1] - offsets[i]);
  }
  return build_batch(ctx, n, ptrs.data(), lens.data(), out);
}
dsa_status dsa_batch_create_packed(dsa_context *ctx, uint32_t n, const uint8_t *blob, const uint64_t *offsets, dsa_batch **out) {
  if (!ctx || !out || (n && (!blob || !offsets))) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "null argument");
  DSA_GUARD(ctx, create_packed(ctx, n, blob, offsets, out));
}

dsa_status dsa_batch_decode(dsa_batch *b) {
  if (!b) return DSA_ERR_INVALID_ARGUMENT;
  dsa_context *ctx = b->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint32_t n = b->n;
  const bool was_decoded = b->decoded;
  b->decoded = true;
  b->collected = false;
  if (b->retry) { dsa_batch_free(b->retry); b->retry = nullptr; }
  b->retry_index.clear();
  // Which stream set: the first, unless the context's previous decode is still running -- then the other one, so that this decode
  // does not stand behind that one's tail on the same in-order streams.  (The second set is made when first wanted: its streams
  // share the hardware queues of their priority classes with the first set's, which a context that decodes one batch at a time
  // should not pay for.)
  int set_turn = 0;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    if (ctx->num_sets > 1 && ctx->last_done && hipEventQuery(ctx->last_done) == hipErrorNotReady) {
      set_turn = ctx->last_set ^ 1;
      if (set_turn == 1 && !ctx->set1_made) {
        if (ctx->sets[1].create(nullptr, ctx->prio_least, ctx->prio_greatest) == hipSuccess) ctx->set1_made = true;
        else { (void)hipGetLastError(); ctx->sets[1].destroy(); set_turn = 0; }
      }
    }
    (void)hipGetLastError();
    ctx->last_set = set_turn;
  }
  StreamSet &S = ctx->sets[set_turn];
  hipStream_t st = S.stream;
  if (n == 0) {
    HIP_TRY(ctx, hipEventRecord(b->ev_done, st));
    HIP_TRY(ctx, hipEventRecord(b->ev_descs, st));
    return DSA_OK;
  }
  const bool prof = ctx->profiling;
  if (prof && !b->have_events) {
    for (int i = 0; i <= DSA_NUM_STAGES; ++i) HIP_TRY(ctx, hipEventCreate(&b->ev[i]));
    for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipEventCreate(&b->ev_sym[i]));
    for (int k = 0; k < KT_COUNT; ++k) for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipEventCreate(&b->ev_k[k][i]));
    b->have_events = true;
  }
  for (int k = 0; k < KT_COUNT; ++k) b->k_timed[k] = false;
  // an event pair around one kernel, on its own stream
  auto k_begin = [&](int k, hipStream_t s) { if (prof && !b->k_timed[k]) (void)hipEventRecord(b->ev_k[k][0], s); };
  auto k_end = [&](int k, hipStream_t s) { if (prof && !b->k_timed[k]) { (void)hipEventRecord(b->ev_k[k][1], s); b->k_timed[k] = true; } };
  int evi = 0;
  auto mark = [&]() -> hipError_t { return prof ? hipEventRecord(b->ev[evi++], st) : hipSuccess; };
  if (b->download_queued && !b->downloaded) HIP_TRY(ctx, hipStreamWaitEvent(st, b->ev_down, 0));   // a download of the previous decode still reads the arena
  if (was_decoded) HIP_TRY(ctx, hipStreamWaitEvent(st, b->ev_done, 0));   // this batch's previous decode may still be running on the other stream set
  b->download_queued = false; b->downloaded = false;
  HIP_TRY(ctx, hipStreamWaitEvent(st, b->ev_uploaded, 0));           // the streams and layouts are in the arena
  HIP_TRY(ctx, hipMemsetAsync(b->d_descs, 0, sizeof(MeshDesc) * n, st));
  HIP_TRY(ctx, hipMemsetAsync(&b->d_globals->pool_cursor, 0, sizeof(unsigned long long) * 2, st));   // the table pool starts empty
  HIP_TRY(ctx, mark());
  {
    uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((b->max_faces + 16383) / 16384, 4));
    hipLaunchKernelGGL(dsa::k_init, dim3(gx, n), dim3(256), 0, st, b->arena, b->d_layouts, n);
  }
  hipLaunchKernelGGL(dsa::k_locate, dim3(n), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n, b->d_globals);
  // tagged symbol streams: a round of {tag stream on a wave of its own, the walk taken up behind it} per attribute a mesh can
  // have (what follows a tagged attribute is only found by decoding its tags); nothing to do without them
  for (uint32_t r = 0; r < std::max<uint32_t>(1, b->max_atts); ++r) {
    hipLaunchKernelGGL(dsa::k_tags, dim3(n), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n);
    hipLaunchKernelGGL(dsa::k_locate_resume, dim3(n), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n, b->d_globals);
  }
  // valence-coded connectivity: the six context lists of every mesh on waves of their own (the register-table decoder), in front
  // of the connectivity waves, which would otherwise decode them one after the other at a third of the speed each
  if (b->any_valence) hipLaunchKernelGGL(dsa::k_valence_lists, dim3(n, 6), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n);
  HIP_TRY(ctx, mark());
  const uint32_t na = std::max<uint32_t>(1, b->max_atts);
  // fork: entropy decode of every attribute stream on the second stream (it only needs k_locate's offsets)
  // DSA_SERIAL=1 (diagnostics): everything on the main stream, so that stage times are stand-alone kernel times
  static const bool serial = getenv("DSA_SERIAL") != nullptr;
  // Everything else the schedule could vary by is fixed in the product library; a -DDSA_EXPERIMENTS build (csrc/Makefile, EXTRA=)
  // reads the switches profiles/README.md reports on from the environment: the schedules and the lane-per-chain kernels that
  // were measured and lost.
#ifdef DSA_EXPERIMENTS
  auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
  static const int fuse_choice = env_int("DSA_FUSE_OPERANDS", -1), chain_choice = env_int("DSA_CHAIN", -1);
  static const uint32_t lane_flags = (uint32_t)env_int("DSA_LANES", (int)PW_FLAG);
  static const bool run_window = env_int("DSA_TRAV_NO_WINDOW", 0) == 0;
  static const int trav_split = env_int("DSA_TRAV_SPLIT", 1), split_mode = env_int("DSA_SYM_SPLIT", 1);
  static const uint32_t wide_flag = env_int("DSA_SYM_WIDE", 1) ? SYM_WIDE : 0u;
  static const bool early_fuse_on = env_int("DSA_EARLY_FUSE", 0) != 0;
  static const uint32_t oct_lpw = (uint32_t)env_int("DSA_OCT_LPW", 16);
  static const bool handoff_on = env_int("DSA_LATE_HANDOFF", 0) != 0;
#else
  const bool handoff_on = false;      // measured: 40.0 - 40.9 -> 44.5 ms (profiles/README.md)
  const int fuse_choice = -1, chain_choice = -1, trav_split = 1, split_mode = 1;
  const uint32_t lane_flags = PW_FLAG, wide_flag = SYM_WIDE;
  const bool run_window = true, early_fuse_on = false;
#endif
  // The octahedral delta one lane per stream (k_predict_oct_streams) when the batch fills the machine anyway -- measured on the
  // bench meshes: 0.5 - 0.8 ms slower up to 3072 meshes (nothing waits for the instructions it saves, and its waves share SIMDs
  // with chain waves), 1.5 - 2 ms faster at 4096 -- and no mesh is much larger than the rest (the kernel takes as long as the
  // longest stream's chain).  DSA_OCT_STREAMS=0|1 (diagnostics, read per decode so that a test can set it) overrides the rule.
  const int oct_choice = []() { const char *e = getenv("DSA_OCT_STREAMS"); return e ? atoi(e) : -1; }();
  const bool oct_rule = n >= 3584 && (uint64_t)b->max_vertices * n <= 4 * b->sum_vertices;
  const uint32_t oct_flag = (oct_choice >= 0 ? oct_choice != 0 : oct_rule) && !(lane_flags & (LN_FLAG_OCT | LN_FLAG_PREDICT)) ? OS_FLAG : 0u;
  hipStream_t st2 = serial ? st : S.stream2, st3 = serial ? st : S.stream3;
  HIP_TRY(ctx, hipEventRecord(S.ev_fork, st));
  HIP_TRY(ctx, hipStreamWaitEvent(st2, S.ev_fork, 0));
  if (b->any_general) {                 // whole-mesh serial decode of the general-path meshes, beside the fast kernels
    HIP_TRY(ctx, hipStreamWaitEvent(st3, S.ev_fork, 0));
    hipLaunchKernelGGL(dsa::k_general, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    hipLaunchKernelGGL(dsa::k_general_tables, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    hipLaunchKernelGGL(dsa::k_general_attributes<0>, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    hipLaunchKernelGGL(dsa::k_general_attributes<1>, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    if (n > 2048) hipLaunchKernelGGL(dsa::k_general_values_crowded, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    else hipLaunchKernelGGL(dsa::k_general_attributes<2>, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    hipLaunchKernelGGL(dsa::k_general_attributes<3>, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    HIP_TRY(ctx, hipEventRecord(S.ev_join3, st3));        // k_finalize(1) dequantises what these kernels decoded
  }
  // ---- the chain of the mesh itself first: its waves must find their slots before the entropy decoders fill the machine
  // the seam streams are checked on the third stream from the start: the check needs k_locate's offsets only (k_seal
  // compares what it finds with the connectivity's edge count)
  if (!b->any_general) HIP_TRY(ctx, hipStreamWaitEvent(st3, S.ev_fork, 0));
  {
    uint32_t lpm = 1;                                  // lanes per mesh >= attribute data per mesh (<= DSA_MAX_ATT_DATA = 7)
    while (lpm < b->max_att_data) lpm *= 2;
    const uint32_t per_wave = WAVE / lpm;
    hipLaunchKernelGGL(dsa::k_conn_checks, dim3((n + per_wave - 1) / per_wave), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n, lpm);
  }
  {
    // the flip bits of GeometricNormal attributes (one serial rABS stream per attribute), beside everything else
    uint32_t lpm = 1;
    while (lpm < na) lpm *= 2;
    const uint32_t per_wave = WAVE / lpm;
    hipStream_t st5 = serial ? st : S.stream5;
    HIP_TRY(ctx, hipStreamWaitEvent(st5, S.ev_fork, 0));
    hipLaunchKernelGGL(dsa::k_flip_bits, dim3((n + per_wave - 1) / per_wave), dim3(WAVE), 0, st5, b->arena, b->d_layouts, b->d_descs, n, lpm, 0u);
    // and the orientation bits of TexCoordsPortable attributes, the same way
    hipLaunchKernelGGL(dsa::k_orient_bits, dim3((n + per_wave - 1) / per_wave), dim3(WAVE), 0, st5, b->arena, b->d_layouts, b->d_descs, n, lpm, 0u);
    // and the crease flags of ConstrainedMultiParallelogram attributes: four streams each (only where the host parse saw the scheme)
    if (b->any_multipara) hipLaunchKernelGGL(dsa::k_crease_bits, dim3((n + WAVE / (4 * lpm) - 1) / (WAVE / (4 * lpm))), dim3(WAVE), 0, st5, b->arena, b->d_layouts, b->d_descs, n, lpm);
    HIP_TRY(ctx, hipEventRecord(S.ev_flips, st5));
  }
  // parallelogram operands: by the traversal waves themselves when the batch keeps the machine busy anyway, by an
  // element-parallel kernel behind the traversal when it does not
  const bool fuse_operands = fuse_choice >= 0 ? fuse_choice != 0 : n >= 2048;
  // late attributes predicted by the second of their producers to finish (dsa_kernels.h: late_handoff) -- with the operands written
  // by the traversal waves, on the wave-per-mesh kernels, corrections and order on one connectivity
  const bool handoff = fuse_operands && !serial && (lane_flags & PW_FLAG) && !(lane_flags & (LN_FLAG_SYMBOLS | LN_FLAG_PREDICT)) && !b->any_seamed && handoff_on;
  const uint32_t trav_flags = (fuse_operands ? 1u : 0u) | (run_window ? 2u : 0u) | (handoff ? 8u : 0u);   // bit 1: adaptive run window
  // connectivity and traversal of a mesh by one wave (k_chain) unless DSA_CHAIN=0 asks for the two kernels: as two kernels,
  // the slots the connectivity waves leave go to waiting entropy-decode waves and most traversal waves start late
  // (a small batch leaves slots free anyway, and is quicker with the faces converted beside the traversal)
  // (a batch with corner attributes runs the two kernels: the tables of the seamed attributes are built between them, so that
  // their symbols and their traversals start beside the position traversal instead of behind it)
  const bool chain = (chain_choice >= 0 ? chain_choice != 0 : n > 2048) && !b->any_seamed;       // measured: equal at 2048, 1.3 ms slower at 1024, 4 ms faster at 4096
  auto launch_faces = [&]() -> hipError_t {      // faces as point ids + link census need the connectivity: third stream
    hipError_t e = hipEventRecord(S.ev_trav, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(st3, S.ev_trav, 0);
    const uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((b->max_faces + 16383) / 16384, 4));
    k_begin(KT_FACES, st3);
    hipLaunchKernelGGL(dsa::k_faces, dim3(gx, n), dim3(256), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    k_end(KT_FACES, st3);
    if (e == hipSuccess) e = hipEventRecord(S.ev_maps, st3);
    return e;
  };
  // (k_seam_maps, 5 - 7 ms, was also measured behind k_texcoords_prepare -- beside the chain of the texture coordinates -- and in
  // front of the GeometricNormal kernels: within 2 ms of this placement either way, better for one dialect and worse for another)
  auto launch_seam_maps = [&]() -> hipError_t {
    hipError_t e = hipEventRecord(S.ev_seambits, st);       // (the event is free by now: the position traversal is queued)
    if (e == hipSuccess) e = hipStreamWaitEvent(st3, S.ev_seambits, 0);
    const uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((b->max_faces + 16383) / 16384, 4));
    hipLaunchKernelGGL(dsa::k_seam_maps, dim3(gx, n), dim3(256), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
    if (e == hipSuccess) e = hipEventRecord(S.ev_maps, st3);
    return e;
  };
  const bool chain_launched = chain;
  if (chain) {
    HIP_TRY(ctx, mark());                              // the connectivity stage has no time of its own
    k_begin(KT_CHAIN, st);
    hipLaunchKernelGGL(dsa::k_chain, dim3(n), dim3(WAVE), CN_LDS_WORDS * 4, st, b->arena, b->d_layouts, b->d_descs, n, trav_flags);
    k_end(KT_CHAIN, st);
  } else {
    k_begin(KT_CONNECTIVITY, st);
    hipLaunchKernelGGL(dsa::k_connectivity, dim3(n), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n);
    k_end(KT_CONNECTIVITY, st);
    if (!b->any_seamed) HIP_TRY(ctx, launch_faces());  // beside the traversal
    if (b->any_seamed) {
      // corner attributes: seam edges, attribute vertices and points per corner from the connectivity and the seam bits; then, beside
      // the position traversal, the traversal of every seamed attribute on its own table and its symbols (below).  Both on the third
      // stream, so that the position traversal starts at once on the main stream -- and in FRONT of k_faces there (which only takes
      // the census of such meshes: on the loaded machine it needs 12 ms that the seam tables must not wait for)
      HIP_TRY(ctx, hipEventRecord(S.ev_trav, st));
      HIP_TRY(ctx, hipStreamWaitEvent(st3, S.ev_trav, 0));
      k_begin(KT_SEAM_TABLES, st3);
      hipLaunchKernelGGL(dsa::k_seam_tables, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
      k_end(KT_SEAM_TABLES, st3);
      // a corner attribute whose extent is its entry count (tagged symbols, uncompressed integers) stopped the walk of its mesh: it
      // is taken up here, as many rounds of {tag stream, walk} as a mesh has attributes (nothing to do for most batches)
      for (uint32_t r = 0; r < std::max<uint32_t>(1, b->max_atts); ++r) {
        hipLaunchKernelGGL(dsa::k_tags, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
        hipLaunchKernelGGL(dsa::k_locate_resume, dim3(n), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n, b->d_globals);
      }
      HIP_TRY(ctx, hipEventRecord(S.ev_tables, st3));
      k_begin(KT_TRAVERSE_ATT, st3);
      hipLaunchKernelGGL(dsa::k_traverse_att, dim3(n, std::max<uint32_t>(1, b->max_att_data)), dim3(WAVE), 0, st3, b->arena, b->d_layouts, b->d_descs, n);
      k_end(KT_TRAVERSE_ATT, st3);
      HIP_TRY(ctx, hipEventRecord(S.ev_att, st3));
      HIP_TRY(ctx, launch_faces());
      {   // the flip bits of GeometricNormal attributes with seams: as many as the attribute has entries, which the seam tables counted
        uint32_t lpm = 1;
        while (lpm < na) lpm *= 2;
        const uint32_t per_wave = WAVE / lpm;
        hipStream_t st5 = serial ? st : S.stream5;
        HIP_TRY(ctx, hipStreamWaitEvent(st5, S.ev_tables, 0));
        hipLaunchKernelGGL(dsa::k_flip_bits, dim3((n + per_wave - 1) / per_wave), dim3(WAVE), 0, st5, b->arena, b->d_layouts, b->d_descs, n, lpm, 1u);
        hipLaunchKernelGGL(dsa::k_orient_bits, dim3((n + per_wave - 1) / per_wave), dim3(WAVE), 0, st5, b->arena, b->d_layouts, b->d_descs, n, lpm, 1u);
        HIP_TRY(ctx, hipEventRecord(S.ev_flips, st5));
      }
    }
    HIP_TRY(ctx, mark());
    const uint32_t per = (n + (uint32_t)trav_split - 1) / (uint32_t)trav_split;
    // (seamed batches: the position traversal at low issue priority beside k_seam_tables measured no different, and started behind
    // the tables it halves them (31 -> 18 ms) but runs beside the attribute traversals, 30 + 33 ms instead of 24 + 28: 1 - 2 ms)
    k_begin(KT_TRAVERSE, st);
    for (uint32_t m0 = 0; m0 < n; m0 += per) {
      const uint32_t cnt = std::min(per, n - m0);
      hipLaunchKernelGGL(dsa::k_traverse, dim3(cnt), dim3(WAVE), 0, st, b->arena, b->d_layouts + m0, b->d_descs + m0, cnt, trav_flags);
    }
    k_end(KT_TRAVERSE, st);
    if (b->any_seamed) {
      // point -> entry maps of the meshes with corner attributes, from the corners: behind both kinds of traversal, on the third
      // stream beside the late prediction (k_seal waits for ev_maps)
      HIP_TRY(ctx, launch_seam_maps());
    }
  }
  HIP_TRY(ctx, mark());
  if (prof) HIP_TRY(ctx, hipEventRecord(b->ev_sym[0], st2));
  // DSA_LANES (diagnostics): bit 0 = raw rANS streams one lane per stream (k_symbols_lanes), bit 1 = prediction one lane per
  // attribute (k_predict_lanes); both measured slower than the wave-per-stream kernels on this workload and off by
  // default (profiles/README.md).  bit 2 = wrap prediction by k_predict_wrap (default on), bit 3 = octahedral delta
  // one lane per stream (k_predict_oct_lanes: frees 2 G scalar + 2 G vector instructions per step, but its own chain is
  // longer than the wave-per-stream kernel's and the traversal beside it does not speed up: measured 2 ms slower, off)
#ifdef DSA_EXPERIMENTS
  if (lane_flags & LN_FLAG_SYMBOLS) {
    const uint32_t groups = (n + WAVE - 1) / WAVE;
    hipLaunchKernelGGL(dsa::lanes::k_symbols_lanes<LN_T2_SYMS>, dim3(groups, na), dim3(WAVE), 0, st2, b->arena, b->d_layouts, b->d_descs, n, lane_flags);
    hipLaunchKernelGGL(dsa::lanes::k_symbols_lanes<LN_T1_SYMS>, dim3(groups, na), dim3(WAVE), 0, st2, b->arena, b->d_layouts, b->d_descs, n, lane_flags);
    hipLaunchKernelGGL(dsa::lanes::k_symbols_lanes<LN_T0_SYMS>, dim3(groups, na), dim3(WAVE), 0, st2, b->arena, b->d_layouts, b->d_descs, n, lane_flags);
  }
#endif
  // Attributes whose prediction does not wait for the traversal ("early": difference, octahedral delta) are decoded,
  // predicted and dequantised on a stream of their own with dispatch priority, beside the traversal; the symbols of the
  // parallelogram attributes ("late") follow on the second stream.  DSA_SYM_SPLIT=0, DSA_SERIAL and the lane-per-chain
  // options keep the single symbol launch.
  const bool sym_split = !serial && !(lane_flags & (LN_FLAG_SYMBOLS | LN_FLAG_PREDICT)) && split_mode != 0;
  hipStream_t st4 = sym_split ? S.stream4 : st2;
  // (gate: k_register_gate in front of the 12-bit kernel, see the crowded-batch schedule below)
  auto launch_symbols = [&](hipStream_t s, uint32_t fl, bool gate = false) {
    fl |= wide_flag;
    const uint32_t tier_blocks = (uint32_t)std::min<uint64_t>((uint64_t)n * na, SYM_TIER_BLOCKS);
    // The kernels for everything but 12-bit-precision streams go first: for most batches they find nothing to do, which takes
    // them microseconds while the machine is still filling and a millisecond and a half once every slot is held by a decoder
    // (they used to follow k_symbols_reg: 1.7 ms of empty launches in front of the early attributes' prediction).
    if (wide_flag) hipLaunchKernelGGL(dsa::k_symbols_wide, dim3(n, na), dim3(WAVE), 0, s, b->arena, b->d_layouts, b->d_descs, n, fl);
    hipLaunchKernelGGL(dsa::k_symbols<1>, dim3(tier_blocks), dim3(WAVE), dsa::sym_tier_lds_bytes(1), s, b->arena, b->d_layouts, b->d_descs, n, na, fl);
    hipLaunchKernelGGL(dsa::k_symbols<0>, dim3(tier_blocks), dim3(WAVE), dsa::sym_tier_lds_bytes(0), s, b->arena, b->d_layouts, b->d_descs, n, na, fl);
    hipLaunchKernelGGL(dsa::k_symbols<2>, dim3(tier_blocks), dim3(WAVE), dsa::sym_tier_lds_bytes(2), s, b->arena, b->d_layouts, b->d_descs, n, na, fl);
    if (gate) hipLaunchKernelGGL(dsa::k_register_gate, dim3(SYM_TIER_BLOCKS), dim3(WAVE), 0, s);
    const int kt = (fl & SYM_EARLY_ONLY) ? KT_SYMBOLS_EARLY : KT_SYMBOLS_LATE;
    k_begin(kt, s);
    hipLaunchKernelGGL(dsa::k_symbols_reg, dim3(n, na), dim3(WAVE), 0, s, b->arena, b->d_layouts, b->d_descs, n, fl);
    k_end(kt, s);
  };
  {
    // The identity maps of point clouds (nothing to do for meshes) go first on the symbol stream, and the symbol kernels of
    // both streams behind them: those few microseconds let every wave of k_chain, which became ready at the same moment,
    // take its slot (4 per SIMD) before the entropy decoders fill the rest; a chain wave that finds its SIMD full waits
    // for a decoder to finish, 5 ms or more (tools/wave_times.py).
    const uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((b->max_vertices + 16383) / 16384, 4));
    hipLaunchKernelGGL(dsa::k_point_maps, dim3(gx, n), dim3(256), 0, st2, b->arena, b->d_layouts, b->d_descs, n);
    HIP_TRY(ctx, hipEventRecord(S.ev_conn, st2));
  }
  // DSA_EARLY_FUSE=1 (measured 1.7 ms slower, off): the wave that decoded an early attribute's symbols also predicts and
  // dequantises it -- the octahedral prediction leaves the tail (7.5 -> 4.7 ms) but its waves keep 80-register slots twice as
  // long and the late symbols end 4.7 ms later
  const uint32_t early_fuse = early_fuse_on && !(lane_flags & (LN_FLAG_PREDICT | LN_FLAG_OCT)) ? SYM_EARLY_FUSE : 0u;
  const uint32_t hand = handoff ? LATE_HANDOFF : 0u;
  if (sym_split && split_mode == 2) {          // early attributes first, then the late ones beside the early prediction
    launch_symbols(st2, lane_flags | SYM_EARLY_ONLY | early_fuse);
    HIP_TRY(ctx, hipEventRecord(S.ev_conn, st2));
    HIP_TRY(ctx, hipStreamWaitEvent(st4, S.ev_conn, 0));
    launch_symbols(st2, lane_flags | SYM_LATE_ONLY | hand);
  } else if (sym_split && oct_flag) {
    // The early attributes have the longer tail behind their symbols when the octahedral delta runs one lane per stream (a
    // chain of 10 - 14 ms), so they get a head start: the late attributes' 12-bit kernel stands behind k_register_gate, idle
    // waves that ask for 136 registers each -- which a SIMD holding four chain waves and three decoders only has once decoders
    // leave and no block of the early kernel is waiting to take their place, i.e. when the early kernel is dispatched to its
    // last block (9 ms into the decode).  The early symbols then end at 17 ms of 33 instead of 27, and the stream kernel in the
    // shadow of the chain.  (Late after ALL early symbols: a millisecond slower.  Early symbols in two launches with the late
    // ones behind the first: the second launch shares the freed slots with the late kernel block for block and ends at 28 ms.
    // Until the registers of the 16 KB tier were put right its idle waves did this by accident.)
    HIP_TRY(ctx, hipStreamWaitEvent(st4, S.ev_conn, 0));
    launch_symbols(st4, lane_flags | SYM_EARLY_ONLY | early_fuse);
    launch_symbols(st2, lane_flags | SYM_LATE_ONLY | hand, ctx->gate_ok);
  } else if (sym_split) {                      // both at once, the early ones on the stream with priority
    HIP_TRY(ctx, hipStreamWaitEvent(st4, S.ev_conn, 0));
    launch_symbols(st4, lane_flags | SYM_EARLY_ONLY | early_fuse);
    launch_symbols(st2, lane_flags | SYM_LATE_ONLY | hand);
  } else launch_symbols(st2, lane_flags | early_fuse | hand);
  if (b->any_seamed) {       // the symbols of corner attributes: their entry counts are k_seam_tables'
    HIP_TRY(ctx, hipEventRecord(S.ev_pred, st2));           // (the corrections of everything else are ready)
    HIP_TRY(ctx, hipStreamWaitEvent(st2, S.ev_tables, 0));
    launch_symbols(st2, lane_flags | SYM_CORNER);
  }
  if (prof) HIP_TRY(ctx, hipEventRecord(b->ev_sym[1], st2));
  HIP_TRY(ctx, hipEventRecord(S.ev_join, st2));           // corrections of the late attributes (without the split: of every attribute) are ready
  // attributes whose prediction needs no traversal data (difference, octahedral delta) are finished on this stream,
  // beside the traversal and the parallelogram attributes; joined before k_seal
#ifdef DSA_EXPERIMENTS
  if (lane_flags & LN_FLAG_PREDICT) hipLaunchKernelGGL(dsa::lanes::k_predict_lanes<16>, dim3((n + 15) / 16, na), dim3(WAVE), 0, st4, b->arena, b->d_layouts, b->d_descs, n, 0u);
  else
#endif
  {
    // the octahedral delta of a crowded batch one lane per stream (64 streams per wave: 1/30 of the instructions; the kernel takes
    // as long as the longest stream's chain whatever the batch size, which a small batch does not want)
    if (oct_flag) {
      k_begin(KT_OCT_STREAMS, st4);
      hipLaunchKernelGGL(dsa::k_predict_oct_streams, dim3((n + WAVE - 1) / WAVE, na), dim3(WAVE), OS_RING * 1024u, st4, b->arena, b->d_layouts, b->d_descs, n);
      k_end(KT_OCT_STREAMS, st4);
    }
    hipLaunchKernelGGL(dsa::k_predict, dim3(n, na), dim3(WAVE), 0, st4, b->arena, b->d_layouts, b->d_descs, n, 0u, lane_flags | oct_flag);
    if (lane_flags & PW_FLAG) {
      k_begin(KT_PREDICT_WRAP_EARLY, st4);
      hipLaunchKernelGGL(dsa::k_predict_wrap, dim3(n, na), dim3(WAVE), 0, st4, b->arena, b->d_layouts, b->d_descs, n, 0u, lane_flags);
      k_end(KT_PREDICT_WRAP_EARLY, st4);
    }
#ifdef DSA_EXPERIMENTS
    if (lane_flags & LN_FLAG_OCT) {
      if (oct_lpw == 8) hipLaunchKernelGGL(dsa::lanes::k_predict_oct_lanes<8>, dim3((n + 7) / 8, na), dim3(WAVE), 0, st4, b->arena, b->d_layouts, b->d_descs, n, lane_flags);
      else if (oct_lpw == 32) hipLaunchKernelGGL(dsa::lanes::k_predict_oct_lanes<32>, dim3((n + 31) / 32, na), dim3(WAVE), 0, st4, b->arena, b->d_layouts, b->d_descs, n, lane_flags);
      else if (oct_lpw == 64) hipLaunchKernelGGL(dsa::lanes::k_predict_oct_lanes<64>, dim3((n + 63) / 64, na), dim3(WAVE), 0, st4, b->arena, b->d_layouts, b->d_descs, n, lane_flags);
      else hipLaunchKernelGGL(dsa::lanes::k_predict_oct_lanes<16>, dim3((n + 15) / 16, na), dim3(WAVE), 0, st4, b->arena, b->d_layouts, b->d_descs, n, lane_flags);
    }
#endif
  }
  {
    uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((3 * b->max_faces + 65535) / 65536, 4));
    hipLaunchKernelGGL(dsa::k_finalize, dim3(gx, n, na), dim3(256), 0, st4, b->arena, b->d_layouts, b->d_descs, n, 0u, lane_flags);
  }
  HIP_TRY(ctx, hipEventRecord(S.ev_early, st4));
  if (chain_launched) HIP_TRY(ctx, launch_faces());      // behind the chain, beside the parallelogram prediction
  if (!fuse_operands) {
    uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((b->max_vertices + 8191) / 8192, 4));
    hipLaunchKernelGGL(dsa::k_para_operands, dim3(gx, n), dim3(256), 0, st, b->arena, b->d_layouts, b->d_descs, n);
  }
  HIP_TRY(ctx, mark());
  bool split_prediction = false;
  if (b->any_seamed && !b->any_multipara && !(lane_flags & LN_FLAG_PREDICT)) {
    // A batch with corner attributes: the attributes on the position connectivity are predicted behind the position traversal,
    // while the seam tables and the attribute traversals (20 - 30 ms more) are still under way -- what comes behind those then
    // finds the positions final (the TexCoordsPortable and GeometricNormal predictors read them)
    HIP_TRY(ctx, hipStreamWaitEvent(st, S.ev_pred, 0));
    if (lane_flags & PW_FLAG) hipLaunchKernelGGL(dsa::k_predict_wrap, dim3(n, na), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n, 1u, lane_flags | PRED_FRONT);
    hipLaunchKernelGGL(dsa::k_predict, dim3(n, na), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n, 1u, lane_flags | PRED_FRONT);
    // (k_vertex_positions stays behind the second prediction launch: it leaves the positions by vertex in the operand region of the
    // position connectivity, which an attribute of that connectivity located only behind the seam tables still reads there)
    split_prediction = true;
  }
  const uint32_t behind = split_prediction ? PRED_BEHIND : 0u;
  HIP_TRY(ctx, hipStreamWaitEvent(st, S.ev_join, 0));   // join: corrections are ready
  if (b->any_seamed) HIP_TRY(ctx, hipStreamWaitEvent(st, S.ev_att, 0));      // join: orders and operands of the seamed attributes
  if (b->any_general) HIP_TRY(ctx, hipStreamWaitEvent(st, S.ev_join3, 0));   // join: the general path's integers are ready
  HIP_TRY(ctx, mark());
#ifdef DSA_EXPERIMENTS
  if (lane_flags & LN_FLAG_PREDICT) hipLaunchKernelGGL(dsa::lanes::k_predict_lanes<32>, dim3((n + 31) / 32, na), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n, 1u);
  else
#endif
  {
    if (lane_flags & PW_FLAG) {
      k_begin(KT_PREDICT_WRAP_LATE, st);
      hipLaunchKernelGGL(dsa::k_predict_wrap, dim3(n, na), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n, 1u, lane_flags | behind);
      k_end(KT_PREDICT_WRAP_LATE, st);
    }
    hipLaunchKernelGGL(dsa::k_predict, dim3(n, na), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n, 1u, lane_flags | behind);
  }
  HIP_TRY(ctx, mark());
  {
    // GeometricNormal attributes: from the final positions (of whichever strand) and the flip bits
    HIP_TRY(ctx, hipStreamWaitEvent(st, S.ev_early, 0));
    HIP_TRY(ctx, hipStreamWaitEvent(st, S.ev_flips, 0));
    const uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((b->max_vertices + 1023) / 1024, 32));     // four entries per thread: 11.6 -> 9.7 ms against sixteen
    const uint32_t gv = std::max<uint32_t>(1, std::min<uint32_t>((b->max_vertices + 1023) / 1024, 64));     // two dependent gathers per vertex: many short threads
    // ConstrainedMultiParallelogram attributes: the parallelograms of every entry at once, then the chain -- in front of the
    // predictors that read the positions
    if (b->any_multipara) {
      hipLaunchKernelGGL(dsa::k_multipara_prepare, dim3(gx, n, na), dim3(256), 0, st, b->arena, b->d_layouts, b->d_descs, n);
      hipLaunchKernelGGL(dsa::k_multipara, dim3((n + 3) / 4, na), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n, b->d_globals);
    }
    hipLaunchKernelGGL(dsa::k_vertex_positions, dim3(gv, n), dim3(256), 0, st, b->arena, b->d_layouts, b->d_descs, n);
    hipLaunchKernelGGL(dsa::k_predict_geometric, dim3(gx, n, na), dim3(256), 0, st, b->arena, b->d_layouts, b->d_descs, n);
    // TexCoordsPortable attributes: what depends on the mesh and the positions for every entry at once, then the chain over the
    // decoded texture coordinates, two lanes per attribute.  (The GeometricNormal kernels beside the chain, on another stream: the
    // chain's 128 waves, placed while the machine is full, share SIMDs among themselves -- 16 -> 27 ms even when those kernels
    // find nothing to do; kept one to a CU by an LDS allocation they still take 24 - 25 ms beside real GeometricNormal work,
    // whose memory traffic outlasts the chain's request distance: 75 against 73 ms.  One after the other.)
    hipLaunchKernelGGL(dsa::k_texcoords_prepare, dim3(gx, n, na), dim3(256), 0, st, b->arena, b->d_layouts, b->d_descs, n);
    k_begin(KT_TEXCOORDS, st);
    hipLaunchKernelGGL(dsa::k_texcoords, dim3((2 * n + WAVE - 1) / WAVE, na), dim3(WAVE), 0, st, b->arena, b->d_layouts, b->d_descs, n);      // two lanes per attribute
    k_end(KT_TEXCOORDS, st);
  }
  {
    uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((3 * b->max_faces + 65535) / 65536, 4));
    hipLaunchKernelGGL(dsa::k_finalize, dim3(gx, n, na), dim3(256), 0, st, b->arena, b->d_layouts, b->d_descs, n, 1u, lane_flags);
  }
  HIP_TRY(ctx, hipStreamWaitEvent(st, S.ev_maps, 0));
  hipLaunchKernelGGL(dsa::k_seal, dim3((n + 255) / 256), dim3(256), 0, st, b->d_descs, n);
  HIP_TRY(ctx, mark());
  HIP_TRY(ctx, hipGetLastError());
  // every other stream has joined `st` by now: this event is the whole decode.  The descriptors follow on the download stream.
  HIP_TRY(ctx, hipEventRecord(b->ev_done, st));
  if (ctx->last_done) HIP_TRY(ctx, hipEventRecord(ctx->last_done, st));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->down, b->ev_done, 0));
  HIP_TRY(ctx, hipMemcpyAsync(b->descs_pin, b->d_descs, sizeof(MeshDesc) * n, hipMemcpyDeviceToHost, ctx->down));
  HIP_TRY(ctx, hipEventRecord(b->ev_descs, ctx->down));
  return DSA_OK;
}

static dsa_status batch_wait(dsa_batch *b);
dsa_status dsa_batch_wait(dsa_batch *b) {
  if (!b) return DSA_ERR_INVALID_ARGUMENT;
  DSA_GUARD(b->ctx, batch_wait(b));
}
static dsa_status batch_wait(dsa_batch *b) {
  dsa_context *ctx = b->ctx;
  if (!b->decoded) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "dsa_batch_decode was not called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (b->collected) {                  // results are in; what can still be outstanding is a download queued since
    if (b->download_queued && !b->downloaded) { HIP_TRY(ctx, hipEventSynchronize(b->ev_down)); b->downloaded = true; }
    if (b->retry) return dsa_batch_wait(b->retry);
    return DSA_OK;
  }
  // this batch's own events, not the streams: another batch of the context may be queued behind it
  HIP_TRY(ctx, hipEventSynchronize(b->ev_descs));
  if (b->download_queued) { HIP_TRY(ctx, hipEventSynchronize(b->ev_down)); b->downloaded = true; }
  if (b->n) memcpy(b->descs.data(), b->descs_pin, sizeof(MeshDesc) * (size_t)b->n);
  if (ctx->profiling && b->have_events && b->n) {
    for (int i = 0; i < STG_TOTAL; ++i) HIP_TRY(ctx, hipEventElapsedTime(&b->stage_ms[i], b->ev[i], b->ev[i + 1]));
    // the symbol stage runs on the second stream: its own event pair (the slot on the main stream is the join wait)
    HIP_TRY(ctx, hipEventElapsedTime(&b->stage_ms[STG_SYMBOLS], b->ev_sym[0], b->ev_sym[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&b->stage_ms[STG_TOTAL], b->ev[0], b->ev[STG_TOTAL]));
    for (int k = 0; k < KT_COUNT; ++k) {
      b->kernel_ms[k] = 0.f;
      if (b->k_timed[k]) HIP_TRY(ctx, hipEventElapsedTime(&b->kernel_ms[k], b->ev_k[k][0], b->ev_k[k][1]));
    }
  }
  // a stream the sizing parse set aside as beyond the device path's limits (more than DSA_MAX_ATT attributes) got no
  // regions, so the kernels stop at the first capacity check: the verdict is the host's
  for (uint32_t i = 0; i < b->n; ++i)
    if (b->host[i].status == ST_NOTIMPL && b->descs[i].status != ST_OK) { b->descs[i].status = ST_NOTIMPL; b->descs[i].detail = 129; }
    else if (b->host[i].status == DSA_ERR_OUT_OF_MEMORY) { b->descs[i].status = DSA_ERR_OUT_OF_MEMORY; b->descs[i].detail = 0; }   // set aside by build_batch
  b->collected = true;
  // second chance: meshes whose prediction schemes need the general path's tables (the fast kernels find that out
  // only behind the symbol streams, where the host parse does not go)
  if (!b->all_general) {
    std::vector<uint32_t> again;
    for (uint32_t i = 0; i < b->n; ++i)
      if (b->descs[i].status == ST_NOTIMPL && b->descs[i].detail == DSA_SITE_RETRY_GENERAL) again.push_back(i);
    if (!again.empty()) {
      std::vector<std::vector<uint8_t>> copies(again.size());
      std::vector<const uint8_t *> ptrs(again.size());
      std::vector<size_t> lens(again.size());
      for (size_t k = 0; k < again.size(); ++k) {
        const MeshLayout &L = b->layouts[again[k]];
        copies[k].resize(L.stream_len);
        if (L.stream_len) HIP_TRY(ctx, hipMemcpy(copies[k].data(), b->arena + L.stream, L.stream_len, hipMemcpyDeviceToHost));
        ptrs[k] = copies[k].data(); lens[k] = L.stream_len;
      }
      dsa_batch *rb = nullptr;
      dsa_status st = build_batch(ctx, (uint32_t)again.size(), ptrs.data(), lens.data(), &rb, true);
      if (st == DSA_OK) st = dsa_batch_decode(rb);
      if (st == DSA_OK && b->download_queued) st = dsa_batch_download(rb, nullptr, 0);      // block 1 of the download
      if (st == DSA_OK) st = dsa_batch_wait(rb);
      if (st != DSA_OK) { if (rb) dsa_batch_free(rb); b->collected = false; return st; }
      b->retry = rb;
      b->retry_index.assign(b->n, -1);
      for (size_t k = 0; k < again.size(); ++k) b->retry_index[again[k]] = (int32_t)k;
    }
  }
  return DSA_OK;
}

void dsa_batch_free(dsa_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->ctx->device);
  if (b->retry) { dsa_batch_free(b->retry); b->retry = nullptr; }
  if (b->have_events) for (int i = 0; i <= DSA_NUM_STAGES; ++i) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
  if (b->have_events) for (int i = 0; i < 2; ++i) if (b->ev_sym[i]) (void)hipEventDestroy(b->ev_sym[i]);
  if (b->have_events) for (int k = 0; k < KT_COUNT; ++k) for (int i = 0; i < 2; ++i) if (b->ev_k[k][i]) (void)hipEventDestroy(b->ev_k[k][i]);
  // nothing of this batch may still be in flight when its arena goes back to the cache (a caller may free without waiting)
  if (b->ev_uploaded) { (void)hipEventSynchronize(b->ev_uploaded); (void)hipEventDestroy(b->ev_uploaded); }
  if (b->ev_done) { if (b->decoded) (void)hipEventSynchronize(b->ev_done); (void)hipEventDestroy(b->ev_done); }
  if (b->ev_descs) { if (b->decoded) (void)hipEventSynchronize(b->ev_descs); (void)hipEventDestroy(b->ev_descs); }
  if (b->ev_down) { if (b->download_queued) (void)hipEventSynchronize(b->ev_down); (void)hipEventDestroy(b->ev_down); }
  give_spare(b->ctx, b->ctx->spare_descs, (uint8_t *)b->descs_pin, b->descs_pin_bytes, [](uint8_t *p) { (void)hipHostFree(p); });
  if (b->mirror && b->mirror_owned) give_spare(b->ctx, b->ctx->spare_mirrors, b->mirror, b->mirror_bytes, [](uint8_t *p) { (void)hipHostFree(p); });
  give_spare(b->ctx, b->ctx->spare_arenas, b->arena, b->arena_cap, [](uint8_t *p) { (void)hipFree(p); });
  give_spare(b->ctx, b->ctx->spare_packed, b->d_packed, b->d_packed_cap, [](uint8_t *p) { (void)hipFree(p); });
  dsa_context *ctx = b->ctx;
  delete b;
  bool last;
  { std::lock_guard<std::mutex> g(ctx->mu); last = --ctx->live_batches == 0 && ctx->doomed.load(); }
  if (last) dsa_context_destroy(ctx);
}

uint32_t dsa_batch_size(const dsa_batch *b) { return b ? b->n : 0; }
uint64_t dsa_batch_arena_bytes(const dsa_batch *b) { return b ? b->arena_bytes : 0; }

uint64_t dsa_batch_algorithmic_bytes(const dsa_batch *b) {
  if (!b || !b->collected) return 0;
  uint64_t total = 0;
  for (uint32_t i = 0; i < b->n; ++i) {
    const MeshDesc &D = b->descs[i];
    if (D.status != ST_OK) continue;
    total += b->layouts[i].stream_len + 12ull * D.num_faces;
    for (uint32_t a = 0; a < D.num_attributes; ++a) {
      const AttrDesc &A = D.att[a];
      total += (uint64_t)A.num_entries * A.nc * dt_len(A.data_type) + 4ull * D.num_points;
    }
  }
  if (b->retry) total += dsa_batch_algorithmic_bytes(b->retry);
  return total;
}

// a mesh that was decoded again lives in the retry batch
#define FOLLOW_RETRY(b, mesh, call)                                                                  \
  if ((b) && (b)->retry && (mesh) < (b)->n && (b)->retry_index[mesh] >= 0) {                         \
    const dsa_batch *rb_ = (b)->retry;                                                               \
    const uint32_t rm_ = (uint32_t)(b)->retry_index[mesh];                                           \
    (void)rb_; (void)rm_;                                                                            \
    return call;                                                                                     \
  }
#define CHECK_MESH(b, mesh)                                                                         \
  if (!(b)) return DSA_ERR_INVALID_ARGUMENT;                                                        \
  if (!(b)->collected) return set_err((b)->ctx, DSA_ERR_INVALID_ARGUMENT, "results not collected: call dsa_batch_wait"); \
  if ((mesh) >= (b)->n) return set_err((b)->ctx, DSA_ERR_INVALID_ARGUMENT, "mesh index %u out of range", (unsigned)(mesh));
#define CHECK_ATTR(b, mesh, a)                                                                      \
  CHECK_MESH(b, mesh)                                                                               \
  if ((b)->descs[mesh].status != ST_OK) return set_err((b)->ctx, (dsa_status)(b)->descs[mesh].status, "mesh %u failed to decode", (unsigned)(mesh)); \
  if ((a) >= (b)->descs[mesh].num_attributes) return set_err((b)->ctx, DSA_ERR_INVALID_ARGUMENT, "attribute index %u out of range", (unsigned)(a));

dsa_status dsa_batch_mesh_info(const dsa_batch *b, uint32_t mesh, dsa_mesh_info *out) {
  FOLLOW_RETRY(b, mesh, dsa_batch_mesh_info(rb_, rm_, out));
  CHECK_MESH(b, mesh);
  if (!out) return DSA_ERR_INVALID_ARGUMENT;
  const MeshDesc &D = b->descs[mesh];
  memset(out, 0, sizeof(*out));
  out->status = D.status; out->detail = D.detail;
  out->major_version = D.major; out->minor_version = D.minor; out->encoder_type = D.encoder_type; out->encoder_method = D.encoder_method;
  out->flags = D.flags;
  out->drc_bytes = b->layouts[mesh].stream_len;
  out->decode_path = D.general ? (b->all_general ? 2 : 1) : 0;
  if (D.status == ST_OK) { out->num_faces = D.num_faces; out->num_points = D.num_points; out->num_attributes = D.num_attributes; }
  return DSA_OK;
}

dsa_status dsa_batch_attribute_info(const dsa_batch *b, uint32_t mesh, uint32_t a, dsa_attribute_info *out) {
  FOLLOW_RETRY(b, mesh, dsa_batch_attribute_info(rb_, rm_, a, out));
  CHECK_ATTR(b, mesh, a);
  if (!out) return DSA_ERR_INVALID_ARGUMENT;
  const AttrDesc &A = b->descs[mesh].att[a];
  memset(out, 0, sizeof(*out));
  out->attribute_type = A.att_type; out->data_type = A.data_type; out->num_components = A.nc; out->normalized = A.normalized;
  out->unique_id = A.unique_id; out->num_entries = A.num_entries; out->byte_stride = dt_len(A.data_type) * A.nc;
  out->decoder_type = A.seq_type; out->prediction_method = A.pred_method; out->prediction_transform = A.pred_transform;
  out->quantization_bits = A.q_bits; out->range = A.q_range;
  for (int c = 0; c < 4; ++c) out->min_values[c] = A.q_min[c];
  return DSA_OK;
}

static dsa_status copy_out(const dsa_batch *b, void *dst, uint64_t off, uint64_t bytes) {
  if (!dst) return DSA_ERR_INVALID_ARGUMENT;
  if (bytes == 0) return DSA_OK;
  if (b->downloaded && !b->mirror_compact && off >= b->out_base && off + bytes <= b->out_base + b->out_bytes) {     // already on the host
    memcpy(dst, b->mirror + (off - b->out_base), (size_t)bytes);
    return DSA_OK;
  }
  HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
  HIP_TRY(b->ctx, hipMemcpy(dst, b->arena + off, bytes, hipMemcpyDeviceToHost));
  return DSA_OK;
}

dsa_status dsa_batch_copy_faces(const dsa_batch *b, uint32_t mesh, int32_t *dst) {
  FOLLOW_RETRY(b, mesh, dsa_batch_copy_faces(rb_, rm_, dst));
  CHECK_MESH(b, mesh);
  if (b->descs[mesh].status != ST_OK) return set_err(b->ctx, (dsa_status)b->descs[mesh].status, "mesh %u failed to decode", mesh);
  if (b->downloaded && b->mirror_compact) {          // from the packed host copy: widened
    if (!dst) return DSA_ERR_INVALID_ARGUMENT;
    const CompactMesh &c = b->compact[mesh];
    const uint8_t *src = b->mirror + align_up(b->out_values_bytes, 4096) + c.faces;
    const size_t ncorn = 3ull * b->descs[mesh].num_faces;
    if (c.u16) for (size_t i = 0; i < ncorn; ++i) dst[i] = (int32_t)((const uint16_t *)src)[i];
    else memcpy(dst, src, 4 * ncorn);
    return DSA_OK;
  }
  return copy_out(b, dst, b->layouts[mesh].faces, 12ull * b->descs[mesh].num_faces);
}
dsa_status dsa_batch_copy_attribute_values(const dsa_batch *b, uint32_t mesh, uint32_t a, void *dst) {
  FOLLOW_RETRY(b, mesh, dsa_batch_copy_attribute_values(rb_, rm_, a, dst));
  CHECK_ATTR(b, mesh, a);
  const AttrDesc &A = b->descs[mesh].att[a];
  if (b->downloaded && b->mirror_compact) {          // the values sub-block heads the compact host copy
    if (!dst) return DSA_ERR_INVALID_ARGUMENT;
    memcpy(dst, b->mirror + (b->layouts[mesh].out[a] - b->out_base - b->out_values), (size_t)A.num_entries * A.nc * dt_len(A.data_type));
    return DSA_OK;
  }
  return copy_out(b, dst, b->layouts[mesh].out[a], (uint64_t)A.num_entries * A.nc * dt_len(A.data_type));
}
dsa_status dsa_batch_copy_point_map(const dsa_batch *b, uint32_t mesh, uint32_t a, uint32_t *dst) {
  FOLLOW_RETRY(b, mesh, dsa_batch_copy_point_map(rb_, rm_, a, dst));
  CHECK_ATTR(b, mesh, a);
  if (b->downloaded && b->mirror_compact) {
    if (!dst) return DSA_ERR_INVALID_ARGUMENT;
    const CompactMesh &c = b->compact[mesh];
    const uint32_t npts = b->descs[mesh].num_points;
    if (c.map[a] == ~0ull) for (uint32_t p = 0; p < npts; ++p) dst[p] = p;
    else memcpy(dst, b->mirror + align_up(b->out_values_bytes, 4096) + c.map[a], 4ull * npts);
    return DSA_OK;
  }
  return copy_out(b, dst, b->layouts[mesh].map[a], 4ull * b->descs[mesh].num_points);
}
dsa_status dsa_batch_copy_portable_values(const dsa_batch *b, uint32_t mesh, uint32_t a, int32_t *dst) {
  FOLLOW_RETRY(b, mesh, dsa_batch_copy_portable_values(rb_, rm_, a, dst));
  CHECK_ATTR(b, mesh, a);
  const AttrDesc &A = b->descs[mesh].att[a];
  if (A.source == SRC_BYTES) return set_err(b->ctx, DSA_ERR_INVALID_ARGUMENT, "generic attributes have no portable form");
  return copy_out(b, dst, b->layouts[mesh].work[a], 4ull * A.num_entries * A.nc_portable);
}

const int32_t *dsa_batch_device_faces(const dsa_batch *b, uint32_t mesh) {
  FOLLOW_RETRY(b, mesh, dsa_batch_device_faces(rb_, rm_));
  if (!b || mesh >= b->n || !b->collected || b->descs[mesh].status != ST_OK) return nullptr;
  return (const int32_t *)(b->arena + b->layouts[mesh].faces);
}
const void *dsa_batch_device_attribute_values(const dsa_batch *b, uint32_t mesh, uint32_t a) {
  FOLLOW_RETRY(b, mesh, dsa_batch_device_attribute_values(rb_, rm_, a));
  if (!b || mesh >= b->n || !b->collected || b->descs[mesh].status != ST_OK || a >= b->descs[mesh].num_attributes) return nullptr;
  return b->arena + b->layouts[mesh].out[a];
}
const uint32_t *dsa_batch_device_point_map(const dsa_batch *b, uint32_t mesh, uint32_t a) {
  FOLLOW_RETRY(b, mesh, dsa_batch_device_point_map(rb_, rm_, a));
  if (!b || mesh >= b->n || !b->collected || b->descs[mesh].status != ST_OK || a >= b->descs[mesh].num_attributes) return nullptr;
  return (const uint32_t *)(b->arena + b->layouts[mesh].map[a]);
}

// ---- whole-batch results on the host: one transfer of the output block
uint64_t dsa_batch_output_bytes(const dsa_batch *b) { return b ? b->out_bytes : 0; }

uint64_t dsa_batch_compact_bytes(const dsa_batch *b) { return b ? align_up(b->out_values_bytes, 4096) + b->packed_bytes : 0; }

static dsa_status batch_download(dsa_batch *b, void *dst, size_t dst_bytes, bool compact);
dsa_status dsa_batch_download(dsa_batch *b, void *dst, size_t dst_bytes) {
  if (!b) return DSA_ERR_INVALID_ARGUMENT;
  DSA_GUARD(b->ctx, batch_download(b, dst, dst_bytes, false));
}
dsa_status dsa_batch_download_compact(dsa_batch *b, void *dst, size_t dst_bytes) {
  if (!b) return DSA_ERR_INVALID_ARGUMENT;
  DSA_GUARD(b->ctx, batch_download(b, dst, dst_bytes, true));
}
static dsa_status batch_download(dsa_batch *b, void *dst, size_t dst_bytes, bool compact) {
  dsa_context *ctx = b->ctx;
  if (!b->decoded) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "dsa_batch_decode was not called");
  if (b->download_queued) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "the batch is already being downloaded");
  const uint64_t host_bytes = compact ? dsa_batch_compact_bytes(b) : b->out_bytes;
  if (dst && dst_bytes < host_bytes) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "destination smaller than the host copy (dsa_batch_output_bytes / dsa_batch_compact_bytes)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (compact && !b->d_packed && b->packed_bytes) {     // the packed block: from the context's cache, like arenas
    uint64_t cap = 0;
    uint8_t *p = take_spare(ctx, ctx->spare_packed, b->packed_bytes, &cap);
    if (!p) {
      hipError_t e = hipMalloc((void **)&p, b->packed_bytes);
      if (e != hipSuccess) { (void)hipGetLastError(); drop_spares(ctx, ctx->spare_packed, false); drop_spares(ctx, ctx->spare_arenas, false); e = hipMalloc((void **)&p, b->packed_bytes); }
      if (e != hipSuccess) { (void)hipGetLastError(); return set_err(ctx, DSA_ERR_OUT_OF_MEMORY, "packed block of %llu bytes: %s", (unsigned long long)b->packed_bytes, hipGetErrorString(e)); }
      cap = b->packed_bytes;
    }
    b->d_packed = p; b->d_packed_cap = cap;
  }
  if (b->mirror && b->mirror_owned && dst) { give_spare(ctx, ctx->spare_mirrors, b->mirror, b->mirror_bytes, [](uint8_t *p) { (void)hipHostFree(p); }); b->mirror = nullptr; }
  if (dst) { b->mirror = (uint8_t *)dst; b->mirror_bytes = dst_bytes; b->mirror_owned = false; }
  else if (!b->mirror || !b->mirror_owned || b->mirror_bytes < host_bytes) {
    if (b->mirror && b->mirror_owned) { give_spare(ctx, ctx->spare_mirrors, b->mirror, b->mirror_bytes, [](uint8_t *p) { (void)hipHostFree(p); }); b->mirror = nullptr; }
    const uint64_t need = host_bytes ? host_bytes : 256;
    uint64_t got = 0;
    uint8_t *p = take_spare(ctx, ctx->spare_mirrors, need, &got);
    if (!p) {
      hipError_t e = hipHostMalloc((void **)&p, need, hipHostMallocDefault);
      if (e != hipSuccess) { (void)hipGetLastError(); drop_spares(ctx, ctx->spare_mirrors, true); e = hipHostMalloc((void **)&p, need, hipHostMallocDefault); }
      if (e != hipSuccess) { (void)hipGetLastError(); return set_err(ctx, DSA_ERR_OUT_OF_MEMORY, "pinned host mirror of %llu bytes: %s", (unsigned long long)need, hipGetErrorString(e)); }
      got = need;
    }
    b->mirror = p; b->mirror_bytes = got; b->mirror_owned = true;
  }
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->down, b->ev_done, 0));
  // in pieces, so that a transfer of gigabytes does not hold the engine against the descriptors of the batch behind it
  const uint64_t piece = 256ull << 20;
  auto copy_down = [&](uint8_t *to, const uint8_t *from, uint64_t bytes) -> hipError_t {
    for (uint64_t at = 0; at < bytes; at += piece) {
      const hipError_t e = hipMemcpyAsync(to + at, from + at, (size_t)std::min(piece, bytes - at), hipMemcpyDeviceToHost, ctx->down);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  };
  if (!compact) HIP_TRY(ctx, copy_down(b->mirror, b->arena + b->out_base, b->out_bytes));
  else {
    // the values as they are; faces and point maps packed by a kernel on the download stream (the copy of the values runs beside it)
    if (b->n && b->packed_bytes) {
      const uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((b->max_faces + 16383) / 16384, 4));
      hipLaunchKernelGGL(dsa::k_pack_output, dim3(gx, b->n), dim3(256), 0, ctx->down, b->arena, b->d_layouts, b->d_descs, b->n, b->d_compact, b->d_packed);
      HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, copy_down(b->mirror, b->arena + b->out_base + b->out_values, b->out_values_bytes));
    HIP_TRY(ctx, copy_down(b->mirror + align_up(b->out_values_bytes, 4096), b->d_packed, b->packed_bytes));
  }
  HIP_TRY(ctx, hipEventRecord(b->ev_down, ctx->down));
  b->download_queued = true;
  b->downloaded = false;
  b->mirror_compact = compact;
  if (b->retry && !b->retry->download_queued) return dsa_batch_download(b->retry, nullptr, 0);   // block 1: the meshes decoded a second time (always the full layout)
  return DSA_OK;
}

const void *dsa_batch_host_output(const dsa_batch *b, uint32_t block) {
  if (!b || !b->downloaded) return nullptr;
  if (block == 0) return b->mirror;
  if (block == 1 && b->retry && b->retry->downloaded) return b->retry->mirror;
  return nullptr;
}

dsa_status dsa_batch_output_layout(const dsa_batch *b, uint32_t mesh, dsa_mesh_output *out) {
  CHECK_MESH(b, mesh);
  if (!out) return DSA_ERR_INVALID_ARGUMENT;
  memset(out, 0, sizeof(*out));
  const dsa_batch *src = b;
  uint32_t m = mesh;
  if (b->retry && b->retry_index[mesh] >= 0) { src = b->retry; m = (uint32_t)b->retry_index[mesh]; out->block = 1; }
  const MeshLayout &L = src->layouts[m];
  if (src->mirror_compact && src->download_queued) {       // the host copy of a compact download: [values sub-block | packed block]
    const CompactMesh &c = src->compact[m];
    const uint64_t pk = align_up(src->out_values_bytes, 4096);
    out->flags = c.u16 ? DSA_OUTPUT_FACES_U16 : 0u;
    out->faces = pk + c.faces;
    for (uint32_t a = 0; a < L.cap_attributes && a < DSA_MAX_ATTRIBUTES; ++a) {
      out->values[a] = L.out[a] - src->out_base - src->out_values;
      out->point_map[a] = c.map[a] == ~0ull ? ~0ull : pk + c.map[a];
    }
    return DSA_OK;
  }
  out->faces = L.faces - src->out_base;
  for (uint32_t a = 0; a < L.cap_attributes && a < DSA_MAX_ATTRIBUTES; ++a) { out->values[a] = L.out[a] - src->out_base; out->point_map[a] = L.map[a] - src->out_base; }
  return DSA_OK;
}

void *dsa_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void dsa_host_free(void *p) { if (p) (void)hipHostFree(p); }
dsa_status dsa_host_register(void *p, size_t bytes) {
  if (!p || !bytes) return DSA_ERR_INVALID_ARGUMENT;
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? DSA_ERR_OUT_OF_MEMORY : DSA_ERR_DEVICE; }
  return DSA_OK;
}
dsa_status dsa_host_unregister(void *p) {
  if (!p) return DSA_ERR_INVALID_ARGUMENT;
  if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return DSA_ERR_DEVICE; }
  return DSA_OK;
}

dsa_status dsa_batch_copy_metadata(const dsa_batch *b, uint32_t mesh, uint8_t *dst, size_t dst_bytes, size_t *length) {
  FOLLOW_RETRY(b, mesh, dsa_batch_copy_metadata(rb_, rm_, dst, dst_bytes, length));
  if (!b || mesh >= b->n || !length) return DSA_ERR_INVALID_ARGUMENT;
  const HostMesh &h = b->host[mesh];
  *length = h.meta_len;
  if (!dst || h.meta_len == 0) return DSA_OK;
  if (dst_bytes < h.meta_len) return set_err(b->ctx, DSA_ERR_INVALID_ARGUMENT, "destination too small");
  return copy_out(b, dst, b->layouts[mesh].stream + h.meta_off, h.meta_len);
}

dsa_status dsa_batch_copy_debug(const dsa_batch *b, uint32_t mesh, int what, void *dst, size_t dst_bytes, size_t *written) {
  FOLLOW_RETRY(b, mesh, dsa_batch_copy_debug(rb_, rm_, what, dst, dst_bytes, written));
  CHECK_MESH(b, mesh);
  const MeshDesc &D = b->descs[mesh];
  if (D.status != ST_OK && what != 4) return set_err(b->ctx, (dsa_status)D.status, "mesh %u failed to decode", mesh);
  const MeshLayout &L = b->layouts[mesh];
  uint64_t off = 0, bytes = 0;
  auto unquad = [](uint32_t c) { return c == DSA_INVALID ? c : 3u * (c >> 2) + (c & 3u); };   // internal 4f+k -> reference 3f+k
  switch (what) {
    case 0:
    case 1: {   // de-interleave the 32-byte face records into the reference's opposite[] / corner_to_vertex[] arrays
      bytes = 12ull * D.num_faces;
      if (bytes > dst_bytes) return set_err(b->ctx, DSA_ERR_INVALID_ARGUMENT, "destination too small");
      if (D.general) {   // the general path keeps the two arrays as they are, back to back
        dsa_status st = copy_out(b, dst, L.frec + (what == 0 ? 0 : bytes), bytes);
        if (st == DSA_OK && written) *written = (size_t)bytes;
        return st;
      }
      const uint32_t words = L.rec_compact ? 4u : 8u;
      std::vector<uint32_t> rec((size_t)D.num_faces * words);
      dsa_status st = copy_out(b, rec.data(), L.frec, 4ull * words * D.num_faces);
      if (st != DSA_OK) return st;
      uint32_t *o = (uint32_t *)dst;
      for (uint32_t f = 0; f < D.num_faces; ++f)
        for (uint32_t k = 0; k < 3; ++k) {
          uint32_t vtx, opp;
          if (L.rec_compact) {       // two 64-bit words of three sign-extended 21-bit fields (dsa_kernels.h, Rec<true>)
            const uint64_t vv = rec[(size_t)f * 4] | ((uint64_t)rec[(size_t)f * 4 + 1] << 32), oo = rec[(size_t)f * 4 + 2] | ((uint64_t)rec[(size_t)f * 4 + 3] << 32);
            vtx = (uint32_t)((int32_t)((uint32_t)(vv >> (21 * k)) << 11) >> 11);
            opp = (uint32_t)((int32_t)((uint32_t)(oo >> (21 * k)) << 11) >> 11);
          } else { vtx = rec[(size_t)f * 8 + k]; opp = rec[(size_t)f * 8 + 4 + k]; }
          o[3 * f + k] = what == 0 ? unquad(opp) : vtx;
        }
      if (written) *written = (size_t)bytes;
      return DSA_OK;
    }
    case 2: {
      bytes = 4ull * D.num_entries;
      if (bytes > dst_bytes) return set_err(b->ctx, DSA_ERR_INVALID_ARGUMENT, "destination too small");
      dsa_status st = copy_out(b, dst, L.d2c, bytes);
      if (st != DSA_OK) return st;
      uint32_t *o = (uint32_t *)dst;
      if (!D.general) for (uint32_t i = 0; i < D.num_entries; ++i) o[i] = unquad(o[i]);
      if (written) *written = (size_t)bytes;
      return DSA_OK;
    }
    case 3: off = L.v2d; bytes = 4ull * D.num_vertices; break;
    case 6: off = L.vstamp; bytes = std::min<uint64_t>(dst_bytes, 4ull * L.cap_vertices); break;     // traversal trace of a -DDSA_TRAV_TRACE build
    case 5: {   // per attribute: {symbol source, alphabet size, rANS precision bits, rANS payload bytes}
      size_t need = sizeof(uint32_t) * 4 * DSA_MAX_ATT;
      if (dst_bytes < need) return set_err(b->ctx, DSA_ERR_INVALID_ARGUMENT, "destination too small");
      uint32_t *o = (uint32_t *)dst;
      for (uint32_t a = 0; a < DSA_MAX_ATT; ++a) { o[4 * a] = D.att[a].source; o[4 * a + 1] = D.att[a].num_symbols; o[4 * a + 2] = D.att[a].precision_bits; o[4 * a + 3] = D.att[a].size_rans; }
      if (written) *written = need;
      return DSA_OK;
    }
    case 4:   // phase clocks recorded by the kernels (host copy)
      if (dst_bytes < 48) return set_err(b->ctx, DSA_ERR_INVALID_ARGUMENT, "destination too small");
      memcpy(dst, D.dbg, dst_bytes < sizeof(D.dbg) ? dst_bytes : sizeof(D.dbg));
      if (written) *written = dst_bytes < sizeof(D.dbg) ? dst_bytes : sizeof(D.dbg);
      return DSA_OK;
    default: return set_err(b->ctx, DSA_ERR_INVALID_ARGUMENT, "unknown debug array %d", what);
  }
  if (bytes > dst_bytes) return set_err(b->ctx, DSA_ERR_INVALID_ARGUMENT, "destination too small");
  if (written) *written = (size_t)bytes;
  return copy_out(b, dst, off, bytes);
}

dsa_status dsa_batch_kernel_times(const dsa_batch *b, float *ms, const char **names, uint32_t capacity, uint32_t *count) {
  if (!b || !count) return DSA_ERR_INVALID_ARGUMENT;
  uint32_t k_out = 0;
  for (int k = 0; k < KT_COUNT; ++k) {
    if (!b->k_timed[k] || !b->collected) continue;
    if (k_out < capacity) { if (ms) ms[k_out] = b->kernel_ms[k]; if (names) names[k_out] = kKernelNames[k]; }
    ++k_out;
  }
  *count = k_out;
  return DSA_OK;
}

const char *dsa_context_schedule_note(const dsa_context *ctx) { return ctx ? ctx->gate_note.c_str() : ""; }

dsa_status dsa_context_trim(dsa_context *ctx) {
  if (!ctx) return DSA_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  drop_spares(ctx, ctx->spare_arenas, false);
  drop_spares(ctx, ctx->spare_packed, false);
  drop_spares(ctx, ctx->spare_mirrors, true);
  drop_spares(ctx, ctx->spare_descs, true);
  { std::lock_guard<std::mutex> g(ctx->mu); ctx->enc_lanes.clear(); }      // dsa_encode_batch is synchronous: its lanes are idle between calls
  return DSA_OK;
}

dsa_status dsa_batch_stage_times(const dsa_batch *b, float ms[DSA_NUM_STAGES], const char *names[DSA_NUM_STAGES]) {
  if (!b || !ms) return DSA_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < DSA_NUM_STAGES; ++i) { ms[i] = b->stage_ms[i]; if (names) names[i] = kStageNames[i]; }
  return DSA_OK;
}

}  // extern "C"

#include "dsa_encode.h"
#include "dsa_pool.h"
