// draco-sharp_amd/csrc/dsa_encode_conn.h  (included by dsa_encode.h)
//
// Encode direction, connectivity on the device: corner table, Edgebreaker symbols,
// depth-first attribute order and parallelogram operand entries of a batch of triangle meshes, one wave per mesh.
//   corner table            CornerTable.cs:15-172 (opposites from shared edges, left-most corners, manifold checks)
//   Edgebreaker traversal   MeshEdgeBreakerEncoder.cs:38-124 (start faces), :158-183 (init face), :185-274 (symbols),
//                           :276-303 (holes), :331-361 (hole ids), :373-390 (topology splits)
//   attribute order         Traverser/DepthFirstTraverser.cs:9-99, MeshTraversalSequencer.cs:13-31
// It is the host coder of dsa_encode_host.h (CornerTable::build, EbEncoder, dfs_sequence) statement by statement on
// arrays in device memory: the table construction and the operand entries on the whole wave, the two traversals --
// sequential by nature, every step decides the next from what is visited -- on lane 0, meshes of a batch in parallel.
// The byte stream that results is the CPU coder's (tests/test_gpu_encode.py compares them).
#pragma once

namespace dsa {

struct EncConn {                   // one per mesh; device memory, mirrored on the host
  uint64_t faces;                  // u32[3F] input: vertex of every corner
  uint64_t opp;                    // u32[3F] opposite corner or INVALID
  uint64_t voff, vcur, vlist;      // u32[V+1], u32[V], u32[3F]: corners by vertex
  uint64_t vcorner;                // u32[V] left-most corner
  uint64_t fvis, vvis;             // u8[F], u8[V]
  uint64_t hole_id, hole_vis;      // i32[V], u8[V]
  uint64_t hrec;                   // uint4[3F] per corner: vertex, corners across the right / left edge, mark of the face
  uint64_t stack;                  // u32[F]
  uint64_t processed, init_corners;// u32[F] each
  uint64_t symbols;                // u8[F] OUTPUT encoder order, bit patterns 0 1 3 5 7
  uint64_t start_bits;             // u8[F] OUTPUT
  uint64_t splits;                 // u32[3 * split_cap] OUTPUT (source, split, edge)
  uint64_t d2c, v2d;               // u32[V], i32[V]
  uint64_t e2v, ops;               // u32[V], i32[3V] OUTPUT for the attribute kernels
  uint32_t F, V, split_cap;
  uint32_t fail_key;               // k_enc_table_corners: the least status code that any vertex earned (the host coder's order of checks), or ~0
  uint32_t num_symbols, num_start_bits, num_splits, num_split_symbols, num_processed, num_init, num_entries, interior_edges;   // OUTPUT
  uint32_t status, detail;         // 0 ok; else the host coder's complaint (see enc_conn_message)
};

enum { ENC_OK = 0, ENC_DEGENERATE = 1, ENC_NONMANIFOLD_EDGE = 2, ENC_RING = 3, ENC_NONMANIFOLD_VERTEX = 4, ENC_ISOLATED = 5, ENC_UNREACHED = 6, ENC_SPLITS = 7 };
static inline const char *enc_conn_message(uint32_t status) {
  switch (status) {
    case ENC_DEGENERATE: return "degenerate face in input mesh";
    case ENC_NONMANIFOLD_EDGE: return "non-manifold edge (duplicate half-edge)";
    case ENC_RING: return "vertex ring does not close";
    case ENC_NONMANIFOLD_VERTEX: return "non-manifold vertex in input mesh";
    case ENC_ISOLATED: return "isolated vertex in input mesh";
    case ENC_UNREACHED: return "traversal did not reach every vertex";
    case ENC_SPLITS: return "too many topology splits";
    default: return "connectivity coding failed";
  }
}

__device__ __forceinline__ uint32_t ec_next(uint32_t c) { return c == DSA_INVALID ? c : ((c + 1) % 3 ? c + 1 : c - 2); }
__device__ __forceinline__ uint32_t ec_prev(uint32_t c) { return c == DSA_INVALID ? c : (c % 3 ? c - 1 : c + 2); }

struct EcTable {                   // the corner table as the traversals see it
  const uint32_t *c2v, *opp, *vcorner;
  __device__ __forceinline__ uint32_t opposite(uint32_t c) const { return c == DSA_INVALID ? c : opp[c]; }
  __device__ __forceinline__ uint32_t vertex(uint32_t c) const { return c == DSA_INVALID ? DSA_INVALID : c2v[c]; }
  __device__ __forceinline__ uint32_t swing_right(uint32_t c) const { return ec_prev(opposite(ec_prev(c))); }
  __device__ __forceinline__ uint32_t swing_left(uint32_t c) const { return ec_next(opposite(ec_next(c))); }
  __device__ __forceinline__ uint32_t right_corner(uint32_t c) const { return opposite(ec_next(c)); }
  __device__ __forceinline__ uint32_t left_corner(uint32_t c) const { return opposite(ec_prev(c)); }
};

__device__ __forceinline__ void ec_fail(EncConn *E, uint32_t status, uint32_t detail) { if (atomicCAS(&E->status, 0u, status) == 0u) E->detail = detail; }
__device__ __forceinline__ void ec_sync() { __threadfence_block(); __syncthreads(); }

#ifdef DSA_ENC_CLOCKS
__device__ unsigned long long g_enc_clocks[16];
#define ENC_CLK(i) do { if (lane == 0) { const uint64_t t_ = realclk(); atomicAdd(&g_enc_clocks[i], (unsigned long long)(t_ - t_last)); t_last = t_; } } while (0)
#else
#define ENC_CLK(i)
#endif

// The corner table is built by kernels of their own, every mesh of the chunk on as many blocks as its corners ask for (grid: blocks
// per mesh x meshes): nothing in it is sequential, and on one wave per mesh -- where the walks below have to live -- it took a
// quarter of the connectivity time.  A step that needs the previous one complete for the whole mesh is a launch.
#define ENC_TABLE_PROLOGUE                                                                          \
  const uint32_t mesh = blockIdx.y;                                                                 \
  if (mesh >= n) return;                                                                            \
  EncConn *E = &conns[mesh];                                                                        \
  if (E->status != ENC_OK) return;                                                                  \
  const uint32_t F = E->F, V = E->V, NC = 3u * F;                                                   \
  const uint32_t t0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;       \
  const uint32_t *c2v = (const uint32_t *)(arena + E->faces);                                       \
  (void)F; (void)V; (void)NC; (void)t0; (void)stride; (void)c2v;

// counts start at zero; the marks of the walks too
__global__ __launch_bounds__(256) void k_enc_table_clear(uint8_t *arena, EncConn *conns, uint32_t n) {
  ENC_TABLE_PROLOGUE
  uint32_t *voff = (uint32_t *)(arena + E->voff);
  uint8_t *fvis = arena + E->fvis, *hole_vis = arena + E->hole_vis;
  int32_t *hole_id = (int32_t *)(arena + E->hole_id), *v2d = (int32_t *)(arena + E->v2d);
  for (uint32_t v = t0; v <= V; v += stride) voff[v] = 0;
  for (uint32_t f = t0; f < F; f += stride) fvis[f] = 0;
  for (uint32_t v = t0; v < V; v += stride) { hole_id[v] = -1; hole_vis[v] = 0; v2d[v] = -1; }
}

// ---- corners by vertex (counting sort: counts -> offsets -> lists), CornerTable.cs:41-67 needs them implicitly
__global__ __launch_bounds__(256) void k_enc_table_count(uint8_t *arena, EncConn *conns, uint32_t n) {
  ENC_TABLE_PROLOGUE
  uint32_t *voff = (uint32_t *)(arena + E->voff);
  for (uint32_t c = t0; c < NC; c += stride) {
    const uint32_t a = c2v[ec_next(c)], b = c2v[ec_prev(c)];
    if (a == b || a == c2v[c] || b == c2v[c]) ec_fail(E, ENC_DEGENERATE, c / 3);
    atomicAdd(&voff[c2v[c] + 1], 1u);
  }
}

// exclusive prefix sum, one wave per mesh: voff[v + 1] held count(v)
__global__ __launch_bounds__(WAVE) void k_enc_table_offsets(uint8_t *arena, EncConn *conns, uint32_t n) {
  const uint32_t mesh = blockIdx.x, lane = threadIdx.x;
  if (mesh >= n) return;
  EncConn *E = &conns[mesh];
  if (E->status != ENC_OK) return;
  const uint32_t V = E->V;
  uint32_t *voff = (uint32_t *)(arena + E->voff), *vcur = (uint32_t *)(arena + E->vcur);
  uint32_t base = 0;
  for (uint32_t v0 = 0; v0 < V; v0 += WAVE) {
    const uint32_t v = v0 + lane;
    uint32_t x = v < V ? voff[v + 1] : 0u, incl = x;
    for (int d = 1; d < WAVE; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d, WAVE); if ((int)lane >= d) incl += y; }
    if (v < V) { voff[v + 1] = base + incl; vcur[v] = base + incl - x; }
    base += (uint32_t)__shfl((int)incl, WAVE - 1, WAVE);
  }
}

__global__ __launch_bounds__(256) void k_enc_table_lists(uint8_t *arena, EncConn *conns, uint32_t n) {
  ENC_TABLE_PROLOGUE
  uint32_t *vcur = (uint32_t *)(arena + E->vcur), *vlist = (uint32_t *)(arena + E->vlist);
  for (uint32_t c = t0; c < NC; c += stride) vlist[atomicAdd(&vcur[c2v[c]], 1u)] = c;
}

// ---- opposites: corner c faces the edge next(c) -> prev(c) = a -> b; its opposite is the corner facing b -> a.  That face has a
// corner k at b whose next is a (opposite: prev(k)) and a corner k' at a whose previous is b (opposite: next(k')): the shorter of
// the two vertex lists is searched, so that a vertex of huge valence (the apex of a cone) costs its neighbours nothing.  The same
// directed edge twice is a non-manifold edge.  (The lists of a vertex come out of the atomic counter in any order; nothing below
// depends on it.)
__global__ __launch_bounds__(256) void k_enc_table_opposites(uint8_t *arena, EncConn *conns, uint32_t n) {
  ENC_TABLE_PROLOGUE
  uint32_t *opp = (uint32_t *)(arena + E->opp);
  const uint32_t *voff = (const uint32_t *)(arena + E->voff), *vlist = (const uint32_t *)(arena + E->vlist);
  for (uint32_t c = t0; c < NC; c += stride) {
    const uint32_t a = c2v[ec_next(c)], b = c2v[ec_prev(c)];
    const uint32_t na = voff[a + 1] - voff[a], nb = voff[b + 1] - voff[b];
    uint32_t found = DSA_INVALID, same = 0;
    if (nb <= na) {
      for (uint32_t i = voff[b]; i < voff[b + 1]; ++i) {
        const uint32_t k = vlist[i];
        if (c2v[ec_next(k)] == a) found = ec_prev(k);          // face (b, a, .): reverse edge
        if (c2v[ec_prev(k)] == a) ++same;                      // face (a, b, .) seen from its corner at b: the edge itself
      }
    } else {
      for (uint32_t i = voff[a]; i < voff[a + 1]; ++i) {
        const uint32_t k = vlist[i];
        if (c2v[ec_prev(k)] == b) found = ec_next(k);          // face (b, a, .) seen from its corner at a
        if (c2v[ec_next(k)] == b) ++same;                      // face (a, b, .): the edge itself
      }
    }
    if (same != 1) ec_fail(E, ENC_NONMANIFOLD_EDGE, c);
    opp[c] = found;
  }
}

// ---- left-most corner per vertex (first corner in index order, moved left to the boundary if there is one) and the manifold
// check: every corner of the vertex is reached by swinging right from there; the count of interior half-edges; and what a step
// of either walk needs of a corner in ONE 16-byte record: the vertex at it, the corners across its right and left edge, and the
// mark of its face (0: not visited; 1: visited; s + 2: visited, and the S with symbol id s was coded at it -- what
// MeshEdgeBreakerEncoder.cs keeps in a face -> split symbol map).  A walk reads the records of the two corners it can move to and
// has the next step's operands AND this step's "is that face done" in the same round trip.
__global__ __launch_bounds__(256) void k_enc_table_corners(uint8_t *arena, EncConn *conns, uint32_t n) {
  ENC_TABLE_PROLOGUE
  const uint32_t *opp = (const uint32_t *)(arena + E->opp), *voff = (const uint32_t *)(arena + E->voff), *vlist = (const uint32_t *)(arena + E->vlist);
  uint32_t *vcorner = (uint32_t *)(arena + E->vcorner);
  uint4 *hrec = (uint4 *)(arena + E->hrec);
  EcTable ct;
  ct.c2v = c2v; ct.opp = opp; ct.vcorner = vcorner;
  uint32_t interior = 0;
  for (uint32_t c = t0; c < NC; c += stride) {
    interior += opp[c] != DSA_INVALID ? 1u : 0u;
    hrec[c] = make_uint4(c2v[c], opp[ec_next(c)], opp[ec_prev(c)], 0u);
  }
  if (interior) atomicAdd(&E->interior_edges, interior);      // half-edges here; the walk kernel halves it
  for (uint32_t v = t0; v < V; v += stride) {
    const uint32_t cnt = voff[v + 1] - voff[v];
    if (cnt == 0) { atomicMin(&E->fail_key, (uint32_t)ENC_ISOLATED); vcorner[v] = DSA_INVALID; continue; }
    uint32_t first = DSA_INVALID;
    for (uint32_t i = voff[v]; i < voff[v + 1]; ++i) first = vlist[i] < first ? vlist[i] : first;
    uint32_t act = ct.swing_left(first), c = first, guard = 0, lm = first;
    while (act != DSA_INVALID && act != first) { c = act; act = ct.swing_left(act); if (++guard >= NC) { atomicMin(&E->fail_key, (uint32_t)ENC_RING); break; } }
    if (act != first) lm = c;
    vcorner[v] = lm;
    uint32_t reach = 0, k = lm;
    do { ++reach; k = ct.swing_right(k); } while (k != DSA_INVALID && k != lm && reach <= cnt);
    if (reach != cnt) atomicMin(&E->fail_key, (uint32_t)ENC_NONMANIFOLD_VERTEX);
  }
}

// ---- the two walks, one wave per mesh
__global__ __launch_bounds__(WAVE) void k_enc_connectivity(uint8_t *arena, EncConn *conns, uint32_t n) {
  const uint32_t mesh = blockIdx.x, lane = threadIdx.x;
  if (mesh >= n) return;
  EncConn *E = &conns[mesh];
  if (E->status != ENC_OK) return;
  if (E->fail_key != 0xFFFFFFFFu) { if (lane == 0) ec_fail(E, E->fail_key, 0); return; }
  const uint32_t F = E->F, V = E->V, NC = 3u * F;
  const uint32_t *c2v = (const uint32_t *)(arena + E->faces);
  uint32_t *opp = (uint32_t *)(arena + E->opp), *vcorner = (uint32_t *)(arena + E->vcorner);
  EcTable ct;
  ct.c2v = c2v; ct.opp = opp; ct.vcorner = vcorner;
#ifdef DSA_ENC_CLOCKS
  uint64_t t_last = realclk();
#endif
  ENC_CLK(2);
  uint8_t *fvis = arena + E->fvis, *vvis = arena + E->vvis, *hole_vis = arena + E->hole_vis;
  int32_t *hole_id = (int32_t *)(arena + E->hole_id);
  uint32_t *stack = (uint32_t *)(arena + E->stack), *processed = (uint32_t *)(arena + E->processed), *init_corners = (uint32_t *)(arena + E->init_corners);
  uint8_t *symbols = arena + E->symbols, *start_bits = arena + E->start_bits;
  uint32_t *splits = (uint32_t *)(arena + E->splits);
  uint32_t *d2c = (uint32_t *)(arena + E->d2c);
  int32_t *v2d = (int32_t *)(arena + E->v2d);
  uint4 *hrec = (uint4 *)(arena + E->hrec);             // k_enc_table_corners

  // The walks are sequential by nature and run on lane 0; what the whole wave does for them is LOOK: "the next corner without an
  // opposite", "the next face not visited" are found 64 candidates at a time, and lane 0 -- which checks again, in order, since
  // its own work may have settled a candidate meanwhile -- only ever sees the hits.
  // Every walk below ends by itself on a mesh that passed the checks above (the host coder, which has no such counter, was fuzzed
  // with 20 000 damaged meshes); the counter only makes sure that a kernel can never spin: a GPU does not take Ctrl-C.
  uint32_t steps = 0;
  bool failed = false;
  const uint32_t step_limit = 64u * NC + 4096u;
  auto runaway = [&]() { if (++steps > step_limit) failed = true; return failed; };
  // ---- hole ids, MeshEdgeBreakerEncoder.cs:331-361
  {
    uint32_t num_holes = 0;
    for (uint32_t base = 0; base < NC; base += WAVE) {
      const uint32_t i = base + lane;
      uint64_t open = __builtin_amdgcn_ballot_w64(i < NC && opp[i] == DSA_INVALID);
      while (open) {
        const uint32_t at = base + (uint32_t)__builtin_ctzll(open);
        open &= open - 1;
        if (lane != 0) continue;
        uint32_t bv = c2v[ec_next(at)];
        if (hole_id[bv] != -1) continue;
        const int32_t id = (int32_t)num_holes++;
        uint32_t c = at;
        while (hole_id[bv] == -1 && !runaway()) {
          hole_id[bv] = id;
          c = ec_next(c);
          while (opp[c] != DSA_INVALID && !runaway()) c = ec_next(opp[c]);
          bv = c2v[ec_next(c)];
        }
      }
    }
  }
  ec_sync();
  // a vertex's marks in one byte: 1 visited, 2 on a boundary (it has a hole id)
  for (uint32_t v = lane; v < V; v += WAVE) vvis[v] = hole_id[v] != -1 ? 2 : 0;
  ec_sync();
  ENC_CLK(3);

  struct Hop { uint32_t v, rc, lc, mark; };                                     // the record of a corner
  auto hop = [&](uint32_t c) {
    const uint4 r = hrec[c == DSA_INVALID ? 0u : c];
    Hop h;
    h.v = r.x; h.rc = r.y; h.lc = r.z; h.mark = r.w;
    return h;
  };
  auto mark_face = [&](uint32_t first_corner, uint32_t mark) {                  // first_corner = 3 * face
    hrec[first_corner].w = mark; hrec[first_corner + 1].w = mark; hrec[first_corner + 2].w = mark;
    fvis[first_corner / 3] = 1;
  };
  // ---- Edgebreaker symbols
  uint32_t nsym = 0, nproc = 0, ninit = 0, nstart = 0, nsplit = 0, nsplit_sym = 0;
  int32_t last_symbol_id = -1;
  auto encode_hole = [&](uint32_t start_corner, bool encode_first) {            // :276-303
    uint32_t c = ec_prev(start_corner);
    while (opp[c] != DSA_INVALID && !runaway()) c = ec_next(opp[c]);
    const uint32_t start_v = c2v[start_corner];
    if (encode_first) vvis[start_v] |= 1;
    if (hole_id[start_v] >= 0) hole_vis[hole_id[start_v]] = 1;
    uint32_t act = c2v[ec_prev(c)];
    while (act != start_v && !runaway()) {
      vvis[act] |= 1;
      c = ec_next(c);
      while (opp[c] != DSA_INVALID && !runaway()) c = ec_next(opp[c]);
      act = c2v[ec_prev(c)];
    }
  };
  auto check_split = [&](int32_t src_symbol, uint32_t edge, uint32_t neighbor_mark) {     // :373-390
    if (neighbor_mark < 2u) return;
    if (nsplit >= E->split_cap) { failed = true; return; }
    splits[3 * nsplit] = (uint32_t)src_symbol; splits[3 * nsplit + 1] = neighbor_mark - 2u; splits[3 * nsplit + 2] = edge;
    ++nsplit;
  };
  // A step reads three things that all hang off what the previous step loaded -- the marks of the vertex, the records of the two
  // corners across -- and issues them together: one round trip per step (a face across an edge is never the face itself on a
  // manifold mesh; the test stays, the mark of the face itself being what this step sets).
  auto encode_from_corner = [&](uint32_t corner0) {                             // :185-274
    uint32_t sp = 0;
    stack[sp++] = corner0;
    while (sp && !failed) {
      uint32_t corner = stack[sp - 1];
      if (corner == DSA_INVALID || fvis[corner / 3]) { --sp; continue; }
      Hop cur = hop(corner);
      for (;;) {
        if (runaway() || nsym >= F || nproc >= F || sp >= F) { failed = true; break; }
        ++last_symbol_id;
        const uint32_t first = corner - corner % 3;
        const uint32_t v = cur.v, rc = cur.rc, lc = cur.lc;
        const uint32_t vm = vvis[v];
        const Hop hr = hop(rc), hl = hop(lc);
        const bool r_self = rc - first < 3u, l_self = lc - first < 3u;            // (an invalid corner is far from any face)
        const bool r_done = rc == DSA_INVALID || r_self || hr.mark != 0, l_done = lc == DSA_INVALID || l_self || hl.mark != 0;
        const bool seen = (vm & 1u) != 0, on_boundary = (vm & 2u) != 0;
        mark_face(first, 1u);
        processed[nproc++] = corner;
        if (!seen) {
          vvis[v] = (uint8_t)(vm | 1u);
          if (!on_boundary) { symbols[nsym++] = 0; corner = rc; cur = hr; continue; }
        }
        if (r_done) {
          if (rc != DSA_INVALID && !r_self) check_split(last_symbol_id, 1, hr.mark);
          if (l_done) {
            if (lc != DSA_INVALID && !l_self) check_split(last_symbol_id, 0, hl.mark);
            symbols[nsym++] = 7;
            --sp;
            break;
          }
          symbols[nsym++] = 5;
          corner = lc; cur = hl;
        } else {
          if (l_done) {
            if (lc != DSA_INVALID && !l_self) check_split(last_symbol_id, 0, hl.mark);
            symbols[nsym++] = 3;
            corner = rc; cur = hr;
          } else {
            symbols[nsym++] = 1;
            ++nsplit_sym;
            if (on_boundary) { const int32_t hid = hole_id[v]; if (!hole_vis[hid]) encode_hole(corner, false); }
            mark_face(first, (uint32_t)last_symbol_id + 2u);
            stack[sp - 1] = lc;
            stack[sp++] = rc;                         // sp <= F: every push marks a face first
            break;
          }
        }
      }
    }
  };
  for (uint32_t base = 0; base < F; base += WAVE) {                               // :38-124
    const uint32_t fl = base + lane;
    uint64_t fresh = __builtin_amdgcn_ballot_w64(fl < F && fvis[fl] == 0);
    while (fresh) {
      const uint32_t face = base + (uint32_t)__builtin_ctzll(fresh);
      fresh &= fresh - 1;
      if (lane != 0) continue;
      // (the reference asks at each of the three corners of a face whether the face is still to do)
      for (int rep = 0; rep < 3 && !failed && !fvis[face]; ++rep) {
        // find_init_face, :158-183
        uint32_t corner = 3 * face, start = DSA_INVALID;
        bool interior_face = true;
        for (int i = 0; i < 3; ++i) {
          if (opp[corner] == DSA_INVALID) { start = corner; interior_face = false; break; }
          if (hole_id[c2v[corner]] != -1) {
            uint32_t rc = corner;
            while (rc != DSA_INVALID && !runaway()) { corner = rc; rc = ct.swing_right(rc); }
            start = ec_prev(corner);
            interior_face = false;
            break;
          }
          corner = ec_next(corner);
        }
        if (interior_face) start = corner;
        start_bits[nstart++] = interior_face ? 1 : 0;
        if (interior_face) {
          vvis[c2v[start]] |= 1; vvis[c2v[ec_next(start)]] |= 1; vvis[c2v[ec_prev(start)]] |= 1;
          mark_face(3 * face, 1u);
          init_corners[ninit++] = ec_next(start);
          const uint32_t o = opp[ec_next(start)];
          if (o != DSA_INVALID && !fvis[o / 3]) encode_from_corner(o);
        } else {
          encode_hole(ec_next(start), true);
          encode_from_corner(start);
        }
      }
    }
  }
  if (lane == 0) {
    ENC_CLK(4);
    if (failed) ec_fail(E, steps > step_limit ? ENC_RING : ENC_SPLITS, nsplit);
    E->num_symbols = nsym; E->num_start_bits = nstart; E->num_splits = nsplit; E->num_split_symbols = nsplit_sym;
    E->num_processed = nproc; E->num_init = ninit; E->interior_edges /= 2;
  }
  ec_sync();
  if (E->status != ENC_OK) return;
  // the traversal marks start over for the attribute order
  for (uint32_t c = lane; c < NC; c += WAVE) hrec[c].w = 0;
  for (uint32_t f = lane; f < F; f += WAVE) fvis[f] = 0;
  for (uint32_t v = lane; v < V; v += WAVE) vvis[v] &= 2;
  ec_sync();
  {
    // ---- depth-first attribute order over the decoder's face order (processed corners last to first, then the init
    // corners), DepthFirstTraverser.cs:9-99.  A vertex is on a boundary -- SwingLeft of its left-most corner is invalid -- exactly
    // when the hole pass gave it an id.
    const uint32_t nproc = E->num_processed, ninit = E->num_init, nstarts = nproc + ninit;
    uint32_t count = 0, dfs_steps = 0;
    bool stuck = false;
    auto visit = [&](uint32_t v, uint32_t vm, uint32_t c) { vvis[v] = (uint8_t)(vm | 1u); v2d[v] = (int32_t)count; if (count < V) d2c[count] = c; ++count; };
    for (uint32_t base = 0; base < nstarts; base += WAVE) {
      const uint32_t i = base + lane;
      const uint32_t mine = i < nstarts ? (i < nproc ? processed[nproc - 1 - i] : init_corners[i - nproc]) : DSA_INVALID;
      uint64_t fresh = __builtin_amdgcn_ballot_w64(mine != DSA_INVALID && fvis[mine / 3] == 0);
      while (fresh) {
        const int from = __builtin_ctzll(fresh);
        fresh &= fresh - 1;
        const uint32_t start = (uint32_t)__shfl((int)mine, from, WAVE);
        if (lane != 0 || stuck) continue;
        if (fvis[start / 3]) continue;
        uint32_t sp = 0;
        stack[sp++] = start;
        const uint32_t nvx = c2v[ec_next(start)], pvx = c2v[ec_prev(start)];
        { const uint32_t m = vvis[nvx]; if (!(m & 1u)) visit(nvx, m, ec_next(start)); }
        { const uint32_t m = vvis[pvx]; if (!(m & 1u)) visit(pvx, m, ec_prev(start)); }
        while (sp && !stuck) {
          uint32_t corner = stack[sp - 1];
          if (corner == DSA_INVALID || fvis[corner / 3]) { --sp; continue; }
          Hop cur = hop(corner);
          for (;;) {
            if (++dfs_steps > 64u * NC + 4096u || sp >= F || count > V) { stuck = true; break; }
            const uint32_t first = corner - corner % 3;
            const uint32_t v = cur.v, rc = cur.rc, lc = cur.lc;
            const uint32_t vm = vvis[v];
            const Hop hr = hop(rc), hl = hop(lc);
            const bool r_done = rc == DSA_INVALID || rc - first < 3u || hr.mark != 0, l_done = lc == DSA_INVALID || lc - first < 3u || hl.mark != 0;
            mark_face(first, 1u);
            if (!(vm & 1u)) {
              visit(v, vm, corner);
              if (!(vm & 2u)) { corner = rc; cur = hr; continue; }
            }
            if (r_done) {
              if (l_done) { --sp; break; }
              corner = lc; cur = hl;
            } else {
              if (l_done) { corner = rc; cur = hr; }
              else { stack[sp - 1] = lc; stack[sp++] = rc; break; }
            }
          }
        }
      }
    }
    if (lane == 0) {
      ENC_CLK(5);
      E->num_entries = count;
      if (stuck) ec_fail(E, ENC_RING, count);
      else if (count != V) ec_fail(E, ENC_UNREACHED, count);
    }
  }
  ec_sync();
  if (E->status != ENC_OK) return;
}

// ---- entry -> vertex and the parallelogram operand entries of every entry (MeshPredictionSchemeParallelogramEncoder.cs:35-56)
__global__ __launch_bounds__(256) void k_enc_operands(uint8_t *arena, EncConn *conns, uint32_t n) {
  ENC_TABLE_PROLOGUE
  const uint32_t *opp = (const uint32_t *)(arena + E->opp), *d2c = (const uint32_t *)(arena + E->d2c);
  const int32_t *v2d = (const int32_t *)(arena + E->v2d);
  uint32_t *e2v = (uint32_t *)(arena + E->e2v);
  int32_t *ops = (int32_t *)(arena + E->ops);
  for (uint32_t p = t0; p < V; p += stride) {
    const uint32_t ci = d2c[p];
    e2v[p] = c2v[ci];
    int32_t on = -1, op = -1, oo = -1;
    if (p > 0) {
      const uint32_t oci = opp[ci];
      if (oci != DSA_INVALID) {
        const int32_t vo = v2d[c2v[oci]], vn = v2d[c2v[ec_next(oci)]], vp = v2d[c2v[ec_prev(oci)]];
        if (vo < (int32_t)p && vn < (int32_t)p && vp < (int32_t)p) { on = vn; op = vp; oo = vo; }
      }
    }
    ops[3 * p] = on; ops[3 * p + 1] = op; ops[3 * p + 2] = oo;
  }
}

}  // namespace dsa
