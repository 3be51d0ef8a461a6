// draco-sharp_amd/csrc/dsa_kernels.h
// HIP kernels of the Draco (bitstream 2.2) mesh-decode path for gfx950 (MI355X).
// One wave64 owns one mesh (or one attribute stream of one mesh); a batch of
// meshes fills the chip.  The serial recurrences the format imposes (rANS
// state, Edgebreaker stack machine, DFS order, parallelogram chain) run with a
// wave-uniform state, the other lanes serving as the table search / window /
// scan engine; everything elementwise is lane-parallel.
//
// Reference restated (paths under /root/reference/src/Draco/IO/):
//   k_locate       DracoDecoder.cs:44-99, Mesh/MeshEdgeBreakerDecoder.cs:25-56,136-193,
//                  Mesh/MeshEdgeBreakerTraversalDecoder.cs:27-61, ConnectivityDecoder.cs:16-44,
//                  Attributes/AttributesDecoder.cs:19-63, SequentialAttributeDecodersController.cs:16-27,
//                  SequentialIntegerAttributeDecoder.cs:23-101, Entropy/SymbolDecoding.cs:7-67,
//                  Entropy/RAnsSymbolDecoder.cs:12-59, AttributeQuantizationTransform.cs:110-121
//   k_connectivity Mesh/MeshEdgeBreakerDecoder.cs:232-471,502-638, MeshEdgeBreakerTraversalDecoder.cs:89-107,
//                  Entropy/AnsDecoder.cs:12-56, BitCoders/RAnsBitDecoder.cs:12-24
//   k_traverse     Mesh/Traverser/DepthFirstTraverser.cs:9-99, MeshAttributeIndicesEncodingObserver.cs:14-21,
//                  MeshTraversalSequencer.cs:13-31, PredictionSchemes/MeshPredictionSchemeParallelogramDecoder.cs:56-89
//   k_symbols      Entropy/RAnsDecoder.cs:20-99, SymbolDecoding.cs:30-67, BitUtilities.cs:72-103
//   k_predict      PredictionSchemes/PredictionSchemeDeltaDecoder.cs:23-37, MeshPredictionSchemeParallelogramDecoder.cs:29-54,
//                  PredictionSchemeWrapDecodingTransform.cs:46-75, PredictionSchemeNormalOctahedron*DecodingTransform.cs
//   k_finalize     AttributeQuantizationTransform.cs:179-199, Core/Dequantizer.cs:15-23,
//                  AttributeOctahedronTransform.cs:82-102, OctahedronToolBox.cs:139-142,220-239,
//                  SequentialIntegerAttributeDecoder.cs:103-160, MeshTraversalSequencer.cs:33-50
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsa_common.h"
#include "dsa_locate.h"
#ifdef DSA_EXPERIMENTS
#include "dsa_lanes.h"       // lane-per-chain kernels: bit-exact, measured slower on every workload tried (profiles/README.md); not in the product library
#else
#define LN_FLAG_SYMBOLS 1u
#define LN_FLAG_PREDICT 2u
#define LN_FLAG_OCT 8u
namespace dsa { namespace lanes {
__device__ __forceinline__ bool ln_sym_eligible(const AttrDesc &, const MeshLayout &, uint32_t, uint32_t) { return false; }
__device__ __forceinline__ bool ln_oct_eligible(const AttrDesc &, uint32_t) { return false; }
} }
#endif

#ifndef DSA_CHAIN_PRIO
#define DSA_CHAIN_PRIO 3      // issue priority of the connectivity / traversal waves
#endif

namespace dsa {

__device__ __forceinline__ uint32_t cnext(uint32_t c) { return (c % 3u == 2u) ? c - 2u : c + 1u; }
__device__ __forceinline__ uint32_t cprev(uint32_t c) { return (c % 3u == 0u) ? c + 2u : c - 1u; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// a wave-uniform pointer the compiler took for a per-lane value (it came out of a vector load): into scalar registers
template <class T>
__device__ __forceinline__ T *uni_ptr(T *p) {
  const uint64_t v = (uint64_t)p;
  return (T *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// The symbol kernels are launched twice when the batch is decoded on four streams: once for the attributes whose prediction
// waits for the traversal (parallelogram: "late") and once, on a stream of higher priority that goes on to predict and
// dequantise them, for those whose prediction does not ("early": difference, octahedral delta, none).
#define SYM_WIDE 0x800u        // k_symbols_wide is part of the launch set (DSA_SYM_WIDE=0: its streams stay with the LDS tiers)
#define SYM_EARLY_ONLY 0x100u
#define SYM_LATE_ONLY 0x200u
#define SYM_CORNER 0x1000u     // the launch for corner attributes (behind k_seam_tables, which counts their entries); every other launch skips them
// late prediction of a batch with corner attributes, in two launches: what only waits for the position traversal / what waits for
// the seam tables and the attribute traversals as well
#define PRED_FRONT 0x10000u
#define PRED_BEHIND 0x20000u
__device__ __forceinline__ bool att_behind_tables(const AttrDesc &a) { return a.corner_data != 0 || a.late_located != 0; }
__device__ __forceinline__ bool att_is_late(const AttrDesc &a) { return (a.have_scheme && a.pred_kind != 0) || att_behind_tables(a); }
#define PW_FLAG 4u             // DSA_LANES bit 2: wrap schemes by k_predict_wrap (default on)
#define LATE_HANDOFF 0x40000u  // late attributes are predicted by the second of their two producers to finish (late_handoff below)
__device__ __forceinline__ void late_handoff(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t flags, uint32_t bit);
__device__ __forceinline__ bool pred_filtered(const AttrDesc &a, uint32_t flags) {
  return ((flags & PRED_FRONT) && att_behind_tables(a)) || ((flags & PRED_BEHIND) && !att_behind_tables(a));
}
__device__ __forceinline__ bool sym_filtered(const AttrDesc &a, uint32_t flags) {
  if (att_behind_tables(a)) return !(flags & SYM_CORNER);
  if (flags & SYM_CORNER) return true;
  const bool late = a.have_scheme && a.pred_kind != 0;      // parallelogram, geometric normal, texture coordinates: after the traversal
  return ((flags & SYM_EARLY_ONLY) && late) || ((flags & SYM_LATE_ONLY) && !late);
}
__device__ __forceinline__ uint64_t clk() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ uint64_t realclk() { return __builtin_amdgcn_s_memrealtime(); }   // constant 100 MHz
// number of leading lanes (from lane 0) whose predicate is set
__device__ __forceinline__ uint32_t leading_lanes(bool pred) { uint64_t m = ~__ballot(pred); return m ? (uint32_t)__builtin_ctzll(m) : 64u; }
// s_waitcnt vmcnt(0) only (expcnt/lgkmcnt left at max).  On gfx950 loads and stores share vmcnt, so a
// load consumed at a loop merge point makes the compiler drain every outstanding store each iteration;
// rare loads are therefore completed inside their own branch with this.
#define WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)

// Wave-wide exclusive prefix sum of one value per lane; total in *total.
// Cross-lane moves on the DPP path (no LDS crossbar round trip as with ds_bpermute / __shfl_up).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false); }
// value of lane-1 (lane 0 receives 0): DPP wave_shr:1
__device__ __forceinline__ uint32_t lane_prev(uint32_t x) { return dpp_mov<0x138>(x); }
// inclusive wave64 scan with an associative op: row_shr 1,2,4,8 inside the 16-lane rows, then row_bcast:15 / :31
template <class Op>
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, Op op) {
  const uint32_t lane = lane_id(), rl = lane & 15u;
  uint32_t t;
  t = dpp_mov<0x111>(x); if (rl >= 1) x = op(t, x);
  t = dpp_mov<0x112>(x); if (rl >= 2) x = op(t, x);
  t = dpp_mov<0x114>(x); if (rl >= 4) x = op(t, x);
  t = dpp_mov<0x118>(x); if (rl >= 8) x = op(t, x);
  t = dpp_mov<0x142>(x); if ((lane & 31u) >= 16) x = op(t, x);
  t = dpp_mov<0x143>(x); if (lane >= 32) x = op(t, x);
  return x;
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t *total) {
  const uint32_t x = wave_incl_scan(v, [](uint32_t a, uint32_t b) { return a + b; });
  *total = rdlane(x, 63);
  return x - v;
}

// =========================================================================
// Face records of the fast path: per face the three vertices and the three opposite corners (corner ids are quad coded, 4 * face + k).
// Two formats, chosen per mesh by the host from the header's counts (MeshLayout::rec_compact):
//   wide     32 B  {v0, v1, v2, 0, o0, o1, o2, 0}                                      any size
//   compact  16 B  {vv, oo}: two 64-bit words of three 21-bit fields, sign-extended on read so that "no opposite" is all ones in
//                  either format; corners and vertices below 2^20 (F <= 262 144).  Half the bytes per hop of the traversal's
//                  gathers, per staged store of the connectivity and per face of k_faces.
// An opposite slot is only ever set once, from "none" (MeshEdgeBreakerDecoder.cs:254,272,314,392 reject a second link; here
// the census of k_faces / k_seal does): in the compact format that is an atomic AND on the 64-bit word -- no read, no wait.
// =========================================================================
template <bool CP> struct Rec;
struct FaceIds { uint32_t v0, v1, v2, o0, o1, o2; };
template <> struct Rec<false> {
  static constexpr uint32_t WORDS = 8;          // dwords per record
  struct Raw { uint4 v, o; };
  static __device__ __forceinline__ Raw load(const uint32_t *frec, uint32_t f) { Raw r; r.v = ((const uint4 *)frec)[(size_t)f * 2]; r.o = ((const uint4 *)frec)[(size_t)f * 2 + 1]; return r; }
  static __device__ __forceinline__ Raw none() { Raw r; r.v = make_uint4(0, 0, 0, 0); r.o = make_uint4(DSA_INVALID, DSA_INVALID, DSA_INVALID, 0); return r; }
  // (operands by value: a conditional expression over members of a referenced struct selects an ADDRESS, which keeps the struct in scratch)
  static __device__ __forceinline__ uint32_t sel3(uint32_t k, uint32_t x, uint32_t y, uint32_t z) { return k == 0 ? x : (k == 1 ? y : z); }
  static __device__ __forceinline__ uint32_t vertex(const Raw &r, uint32_t k) { return sel3(k, r.v.x, r.v.y, r.v.z); }
  static __device__ __forceinline__ uint32_t opp(const Raw &r, uint32_t k) { return sel3(k, r.o.x, r.o.y, r.o.z); }
  // single fields straight from memory (rare paths and the element-parallel passes)
  static __device__ __forceinline__ uint32_t get_v(const uint32_t *frec, uint32_t c) { return frec[fv_idx(c)]; }
  static __device__ __forceinline__ uint32_t get_o(const uint32_t *frec, uint32_t c) { return frec[fo_idx(c)]; }
  static __device__ __forceinline__ uint32_t get_o_plain(const uint32_t *frec, uint32_t c) { return frec[fo_idx(c)]; }
  static __device__ __forceinline__ void set_v(uint32_t *frec, uint32_t c, uint32_t val) { frec[fv_idx(c)] = val; }
  static __device__ __forceinline__ void link(uint32_t *frec, uint32_t c, uint32_t val) { frec[fo_idx(c)] = val; }           // opposite slot: none -> val
  static __device__ __forceinline__ void link_lds(uint32_t *stage, uint32_t c_rel, uint32_t val) { stage[fo_idx(c_rel)] = val; }
  static __device__ __forceinline__ void store(uint32_t *frec, uint32_t f, const FaceIds &x) {
    ((uint4 *)frec)[(size_t)f * 2] = make_uint4(x.v0, x.v1, x.v2, 0u); ((uint4 *)frec)[(size_t)f * 2 + 1] = make_uint4(x.o0, x.o1, x.o2, 0u);
  }
  static __device__ __forceinline__ void store_lds(uint32_t *stage, uint32_t slot, const FaceIds &x) {
    *(uint4 *)&stage[slot * 8] = make_uint4(x.v0, x.v1, x.v2, 0u); *(uint4 *)&stage[slot * 8 + 4] = make_uint4(x.o0, x.o1, x.o2, 0u);
  }
  static __device__ __forceinline__ uint4 vertices_of(const uint32_t *frec, uint32_t f) { return ((const uint4 *)frec)[(size_t)f * 2]; }   // .x .y .z
};
template <> struct Rec<true> {
  static constexpr uint32_t WORDS = 4;
  static constexpr uint32_t M = 0x1FFFFFu;
  struct Raw { uint64_t v, o; };
  static __device__ __forceinline__ uint64_t pack(uint32_t a, uint32_t b, uint32_t c) { return (uint64_t)(a & M) | ((uint64_t)(b & M) << 21) | ((uint64_t)(c & M) << 42); }
  static __device__ __forceinline__ uint32_t field(uint64_t w, uint32_t k) { return (uint32_t)((int32_t)((uint32_t)(w >> (21u * k)) << 11) >> 11); }   // sign-extended
  static __device__ __forceinline__ Raw load(const uint32_t *frec, uint32_t f) { const uint4 q = ((const uint4 *)frec)[f]; Raw r; r.v = (uint64_t)q.x | ((uint64_t)q.y << 32); r.o = (uint64_t)q.z | ((uint64_t)q.w << 32); return r; }
  static __device__ __forceinline__ Raw none() { Raw r; r.v = 0; r.o = ~0ull; return r; }
  static __device__ __forceinline__ uint32_t vertex(const Raw &r, uint32_t k) { return field(r.v, k); }
  static __device__ __forceinline__ uint32_t opp(const Raw &r, uint32_t k) { return field(r.o, k); }
  static __device__ __forceinline__ uint32_t get_v(const uint32_t *frec, uint32_t c) { return field(((const uint64_t *)frec)[(size_t)(c >> 2) * 2], c & 3u); }
  // opposite words are changed by atomics (performed in L2, this CU's L1 keeps what it has): read them past L1
  static __device__ __forceinline__ uint32_t get_o(const uint32_t *frec, uint32_t c) {
    return field(__hip_atomic_load((const unsigned long long *)frec + (size_t)(c >> 2) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c & 3u);
  }
  // once nothing links any more (behind the acquire fence of k_chain / the kernel boundary): an ordinary load
  static __device__ __forceinline__ uint32_t get_o_plain(const uint32_t *frec, uint32_t c) { return field(((const uint64_t *)frec)[(size_t)(c >> 2) * 2 + 1], c & 3u); }
  static __device__ __forceinline__ void set_v(uint32_t *frec, uint32_t c, uint32_t val) {      // plain read-modify-write: lane 0 on synchronised memory only
    uint64_t *w = (uint64_t *)frec + (size_t)(c >> 2) * 2;
    const uint32_t sh = 21u * (c & 3u);
    *w = (*w & ~((uint64_t)M << sh)) | ((uint64_t)(val & M) << sh);
  }
  static __device__ __forceinline__ void link(uint32_t *frec, uint32_t c, uint32_t val) {
    const uint32_t sh = 21u * (c & 3u);
    atomicAnd((unsigned long long *)frec + (size_t)(c >> 2) * 2 + 1, ((unsigned long long)(val & M) << sh) | ~((unsigned long long)M << sh));
  }
  static __device__ __forceinline__ void link_lds(uint32_t *stage, uint32_t c_rel, uint32_t val) {
    const uint32_t sh = 21u * (c_rel & 3u);
    atomicAnd((unsigned long long *)stage + (size_t)(c_rel >> 2) * 2 + 1, ((unsigned long long)(val & M) << sh) | ~((unsigned long long)M << sh));
  }
  static __device__ __forceinline__ uint4 quad(const FaceIds &x) { const uint64_t v = pack(x.v0, x.v1, x.v2), o = pack(x.o0, x.o1, x.o2); return make_uint4((uint32_t)v, (uint32_t)(v >> 32), (uint32_t)o, (uint32_t)(o >> 32)); }
  static __device__ __forceinline__ void store(uint32_t *frec, uint32_t f, const FaceIds &x) { ((uint4 *)frec)[f] = quad(x); }
  static __device__ __forceinline__ void store_lds(uint32_t *stage, uint32_t slot, const FaceIds &x) { *(uint4 *)&stage[slot * 4] = quad(x); }
  static __device__ __forceinline__ uint4 vertices_of(const uint32_t *frec, uint32_t f) {
    const uint64_t v = ((const uint64_t *)frec)[(size_t)f * 2];
    return make_uint4(field(v, 0), field(v, 1), field(v, 2), 0u);
  }
};
__device__ __forceinline__ uint32_t k_next(uint32_t k) { return k == 2u ? 0u : k + 1u; }
__device__ __forceinline__ uint32_t k_prev(uint32_t k) { return k == 0u ? 2u : k - 1u; }

// The per-attribute-data block of the fast seam path (SeamLayout) and the parallelogram operands an attribute's prediction reads:
// the position table's, or those of the attribute's own corner table (k_traverse_att).
__device__ __forceinline__ uint8_t *seam_block(uint8_t *arena, const MeshLayout &L, const SeamLayout &g, uint32_t d) {
  return arena + L.seam + g.data + (uint64_t)d * g.data_stride;
}
__device__ __forceinline__ const uint32_t *att_para(uint8_t *arena, const MeshLayout &L, const MeshDesc *D, const AttrDesc &a) {
  if (a.corner_data == 0) return (const uint32_t *)(arena + L.para);
  const SeamLayout g = seam_layout(L.cap_faces, L.cap_vertices, D->num_att_data, L.rec_compact != 0);
  return (const uint32_t *)(seam_block(arena, L, g, (uint32_t)a.corner_data - 1u) + g.para);
}


// A whole rABS block (RAnsBitDecoder.cs:12-24, AnsDecoder.cs:42-56) decoded by ONE LANE into bit words -- the seam bits of an
// attribute data, the flip bits of a GeometricNormal attribute, the orientation bits of a TexCoordsPortable one: serial streams
// that 64 lanes of a wave decode side by side, each its own.  Two things keep the shared instruction stream short: every lane is
// at the same bit index (one loop, one store per 32 bits, no group-of-zeros shortcut that would send the lanes down different
// paths), and the stream bytes come from two aligned words held in registers, the lower one requested a word ahead of its use
// (a load inside the renormalisation branch would make the whole wave wait for memory at nearly every bit: some lane always
// renormalises).  TOGGLE: a decoded bit says "same as the one before" (MeshPredictionSchemeTexCoordsPortableDecoder.cs:66-85).
// Returns the index of the first set bit (DSA_INVALID: none).
template <bool TOGGLE>
__device__ __forceinline__ uint32_t rabs_block_to_words(const Rabs &rb, uint32_t count, uint32_t *words) {
  uint32_t state = rb.state, left = rb.off, first = DSA_INVALID;
  const uint32_t p = rb.p;
  // the next byte to take is buf[left - 1]
  const uintptr_t a0 = (uintptr_t)rb.buf + (left ? left - 1u : 0u);
  const uint32_t *wp = (const uint32_t *)(a0 & ~(uintptr_t)3);
  uint32_t sh = (uint32_t)(a0 & 3u) * 8u;
  uint32_t cur = wp[0], nxt = wp[-1];                      // (reads below the block stay inside the arena: layouts and streams precede it)
  uint32_t word = 0, last = 1;
  for (uint32_t i = 0; i < count; ++i) {
    if (state < 4096u && left > 0u) {
      state = state * 256u + ((cur >> sh) & 255u);
      --left;
      if (sh == 0u) { sh = 24u; cur = nxt; --wp; nxt = wp[-1]; } else sh -= 8u;
    }
    const uint32_t quot = state >> 8, rem = state & 255u, xn = quot * p;
    const bool val = rem < p;
    state = val ? xn + rem : state - xn - p;
    if (val && first == DSA_INVALID) first = i;
    if (TOGGLE) { last ^= val ? 0u : 1u; word |= last << (i & 31u); }
    else word |= (val ? 1u : 0u) << (i & 31u);
    if ((i & 31u) == 31u) { words[i >> 5] = word; word = 0; }
  }
  if (count & 31u) words[count >> 5] = word;
  return first;
}

// =========================================================================
// k_locate, k_locate_resume: one wave per mesh, lane 0 walks the stream (dsa_locate.h).  k_locate goes to the end of the stream
// or to the first tagged symbol stream; k_tags (below, with the symbol kernels) decodes that tag stream on the whole wave, and
// k_locate_resume takes the walk up behind it, to the next tag stream or the end -- the host queues as many rounds of the two as
// a mesh of the batch has attributes; without tagged streams they find nothing to do.  (A tag stream k_tags is not made for -- a
// mesh too small for its tables, a one-tag alphabet -- is decoded by the resuming lane itself, as every tag stream used to be:
// 23 ms in front of everything for the bench batch with tagged symbols.)
// =========================================================================
__global__ __launch_bounds__(WAVE, 4) void k_locate(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, BatchGlobals *G) {
  uint32_t mesh = blockIdx.x;
  if (mesh >= n || threadIdx.x != 0) return;
  MeshDesc *D = &descs[mesh];
  locate_mesh(arena, layouts[mesh], D);
  if (D->status != ST_OK || D->general) return;
  locate_attribute_headers(arena, layouts[mesh], D);
  if (D->status != ST_OK) return;
  locate_attribute_values(arena, layouts[mesh], D, G, nullptr, nullptr, LOC_UNTIL_TAGS);
}
__global__ __launch_bounds__(WAVE, 4) void k_locate_resume(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, BatchGlobals *G) {
  __shared__ uint32_t s_cum[LOC_MAX_TAGS + 1];
  __shared__ __attribute__((aligned(16))) uint32_t s_lut[LOC_LDS_WORDS];     // a tag stream decoded by the walk itself: slot table, byte ring, tags of a block
  uint32_t mesh = blockIdx.x;
  if (mesh >= n || threadIdx.x != 0) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || !D->values_pending) return;
  if (D->values_pending == 2 && !D->seam_tables_done) return;      // waits for the entry count of a corner attribute (k_seam_tables)
  locate_attribute_values(arena, layouts[mesh], D, G, s_cum, s_lut, LOC_RESUME);
}

// =========================================================================
// k_connectivity: standard Edgebreaker stack machine, one wave per mesh.
// =========================================================================
// Internal corner ids are "quad coded": corner k of face f is 4*f + k, so face = c >> 2, k = c & 3 with no
// division, and a face record is 32 bytes: {v0, v1, v2, flags, o0, o1, o2, 0} (vertices, opposite corners).
// The host converts to the reference's 3*f + k numbering where a corner id leaves the library.
// Runs of the strip pattern (C R)^k are retired up to 64 pairs per step (see cr_run below); everything
// else goes through the scalar machine.
// k_connectivity runs the Edgebreaker stack machine on wave-uniform state: every lane executes the same
// scalar program (values broadcast with readfirstlane live in SGPRs, branches are scalar, no exec-mask
// juggling), because with one wave per mesh the chip is instruction-issue bound and the per-symbol path
// has to stay at a few dozen instructions.  Single words are written by all lanes to the same address
// (one LDS / memory transaction); the 64 lanes become useful for
//   * flushing 64 staged face records (2 KB of LDS) to global memory with full-width stores,
//   * filling / writing back a block of the per-vertex record cache (64 x {left-most corner, vertex
//     before it}, write-back, direct mapped) with one coalesced access,
//   * refilling the 256-byte window of symbol bits.
// What keeps the common symbols short: the active corner is always corner 0 of the previous face, whose
// vertices stay in registers; C reads one per-vertex record from the LDS cache; R/L/E read nothing.
// "corner already has an opposite" is checked after the loop by a lane-parallel symmetry pass.
// Rare events (S, topology splits, start faces, vertex compaction) sync everything to global memory and
// run there on lane 0.
#define CN_REC_BLOCKS 8      // 8 blocks x 64 records x 8 B = 4 KB of LDS, write-back (16 blocks measured no faster; the LDS is worth more to the symbol tiers beside)
#define CN_STAGE 32          // faces per staging block (LDS per wave: 1 KB stage + 4 KB records + 256 B window = 5.4 KB)
#define CN_WIN 64            // dwords of symbol bits per window

#define CN_CTX_WORDS (6 * 64)     // valence traversal: a 64-symbol window of each of the six context lists
#define CN_LDS_WORDS (CN_STAGE * 8 + CN_REC_BLOCKS * 64 * 2 + CN_WIN + CN_CTX_WORDS)
// The body of k_connectivity for one mesh on one wave; LDS: sh_stage[CN_STAGE * 8] and sh_rec[CN_REC_BLOCKS * 128] 16-byte
// aligned, sh_win[CN_WIN].  Also the first half of k_chain.
// VAL: valence traversal (MeshEdgeBreakerTraversalValenceDecoder.cs:22-154).  The symbols come from six lists, chosen by the valence
// of the vertex at corner 1 of the newest face; the machine keeps the valences of the two gate vertices (and their records) in
// scalar registers and every other vertex's valence in the upper 11 bits of its record's first word (the 16-byte face records
// this mode requires keep corners below 2^20; a valence saturates at 2047, far above the 7 the contexts distinguish).  The lists
// are rANS streams over five symbols: the wave decodes them first, into the faces output region (written by k_faces much later).
// Strip runs are found in this mode too: the lanes work out the lists their symbols come from and look the symbols up.
#define VAL_LM(x_) ((x_) == DSA_INVALID ? DSA_INVALID : ((x_) & 0x1FFFFFu))
#define VAL_OF(x_) ((x_) >> 21)
__device__ __forceinline__ uint32_t val_pack(uint32_t lm, uint32_t val) { return (lm & 0x1FFFFFu) | ((val < 2047u ? val : 2047u) << 21); }
// One context list: raw rANS stream over at most 64 symbols (Entropy/RAnsSymbolDecoder.cs:12-59, RAnsDecoder.cs:20-99), decoded by the
// whole wave: cumulative frequencies across the lanes, one ballot per symbol.  sh_tmp: 64 words of LDS.
__device__ __forceinline__ bool valence_decode_list(MeshDesc *D, const uint8_t *stream, uint32_t stream_len, uint32_t off_table, uint32_t nsym, uint32_t P,
                                                    uint32_t off_rans, uint32_t size_rans, uint32_t count, uint32_t *out, uint32_t *sh_tmp) {
  const uint32_t lane = lane_id();
  const uint32_t precision = 1u << P, l_base = precision * 4;
  __syncthreads();
  if (lane == 0) { Rd r(stream, stream_len, off_table); if (!read_prob_table(r, nsym, sh_tmp)) fail(D, ST_INVALID, 400); }
  __syncthreads();
  if (status_of(D) != ST_OK) return false;
  const uint32_t pr = lane < nsym ? sh_tmp[lane] : 0u;
  uint32_t tot;
  const uint32_t ex = wave_excl_scan(pr, &tot);
  if (tot != precision) { if (lane == 0) fail(D, ST_INVALID, 401); return false; }
  const uint32_t cum = lane < nsym ? ex : precision;            // beyond the alphabet: never <= rem
  const uint8_t *buf = stream + off_rans;
  uint32_t x, off;
  { uint32_t st = 0, o = 0; if (!rans_init(buf, size_rans, l_base, &st, &o)) { if (lane == 0) fail(D, ST_INVALID, 402); return false; } x = uni(st); off = uni(o); }
  const uintptr_t base_addr = (uintptr_t)buf;
  const uint32_t mis = (uint32_t)(base_addr & 3u);
  const uint32_t *abuf = (const uint32_t *)(base_addr - mis);
  uint32_t chunk = 0xFFFFFFFFu, W = 0, mine = 0;
  const uint32_t mask = precision - 1;
  for (uint32_t i = 0; i < count; ++i) {
    while (x < l_base && off > 0) {
      --off;
      const uint32_t q = off + mis, ch = q >> 8;
      if (ch != chunk) { chunk = ch; W = abuf[(size_t)ch * 64 + lane]; }      // arena padding makes the over-read safe
      x = (x << 8) | ((rdlane(W, (q & 255u) >> 2) >> ((q & 3u) * 8)) & 0xFFu);
    }
    const uint32_t rem = x & mask;
    const uint32_t j = (uint32_t)__popcll(__ballot(cum <= rem)) - 1u;         // cum of lane 0 is 0: at least one
    const uint32_t cs = rdlane(cum, j), nx = j < 63 ? rdlane(cum, j + 1) : precision;
    x = (nx - cs) * (x >> P) + rem - cs;
    if ((i & 63u) == lane) mine = j;
    if ((i & 63u) == 63u) out[i - 63u + lane] = mine;
  }
  const uint32_t tail = count & 63u;
  if (tail && lane < tail) out[count - tail + lane] = mine;
  return true;
}

template <bool CP, bool VAL>
__device__ __forceinline__ void connectivity_wave(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t *sh_stage, uint32_t *sh_rec, uint32_t *sh_win, uint32_t *sh_ctx) {
  typedef Rec<CP> R;
  static_assert(!VAL || CP, "valence traversal on the fast kernels needs the 16-byte face records");
  if (D->status != ST_OK || D->encoder_type == 0 || D->general) return;   // point clouds have no connectivity; general meshes: k_general
  const uint8_t *s = arena + L.stream;
  uint32_t *frec = (uint32_t *)(arena + L.frec);
  uint2 *vrec = (uint2 *)(arena + L.vrec);
  uint32_t *stack_mem = (uint32_t *)(arena + L.para);     // active corners below the top (<= #E <= cap_vertices/3); para[] is written later
  uint32_t *invalid_list = (uint32_t *)(arena + L.vrank); // merged-away vertices (<= num_split_symbols); consumed before vrank[] is written
  uint32_t *events = (uint32_t *)(arena + L.splits);      // (source, split | edge<<31) per topology split event
  uint32_t *split_map = (uint32_t *)(arena + L.fstamp);   // topologySplitActiveCorners: decoder symbol id -> corner (k_init: INVALID)
  const uint32_t F = uni(D->num_faces), VMAX = uni(L.cap_vertices);
  const uint32_t num_symbols = uni(D->num_symbols);
  const bool remove_invalid = uni(D->num_att_data) == 0;
  const uint32_t lane = lane_id();
  const uint32_t nsplits = uni(D->num_splits);

  // ---- topology split events, MeshEdgeBreakerDecoder.cs:164-193 (lane 0, rare)
  if (lane == 0 && nsplits) {
    Rd r(s, L.stream_len, D->off_splits);
    uint32_t last = 0;
    for (uint32_t i = 0; i < nsplits; ++i) {
      uint32_t source = (uint32_t)r.varint() + last;
      uint32_t delta = (uint32_t)r.varint();
      if (!r.ok || delta > source) { fail(D, ST_INVALID, 200); break; }
      uint32_t edge = read_bits(s + D->off_split_bits, L.stream_len - D->off_split_bits, i, 1);
      events[2 * i] = source;
      events[2 * i + 1] = ((source - delta) & 0x7FFFFFFFu) | (edge << 31);
      last = source;
    }
  }
  uint32_t tagv = DSA_INVALID;          // lane i < CN_REC_BLOCKS: block resident in record-cache slot i
#define TAG(slot_) rdlane(tagv, (slot_))
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  const uint64_t t_start = clk(), r_start = realclk();

  // symbol section as 4-byte aligned words (the arena pads streams, so whole-window reads stay in bounds)
  const uint32_t sym_off = uni(D->off_symbols);
  const uint32_t sym_mis = (uint32_t)((L.stream + sym_off) & 3u);
  const uint32_t *sym_words = (const uint32_t *)(arena + (L.stream + sym_off - sym_mis));
  const uint64_t bit_end = ((uint64_t)sym_mis + uni(D->size_symbols)) * 8;

  // ---- wave-uniform machine state
  uint32_t sid = 0, stage_base = 0;
  uint32_t T1 = 0, T2 = 0;              // vertices at corners 1,2 of the top face (the top is corner 0 of face sid-1)
  bool have_top = false;
  uint32_t sp = 0;                      // entries of stack_mem below the top
  uint32_t num_verts = 0, num_invalid = 0;
  uint32_t splits_left = nsplits;
  uint32_t next_src = nsplits ? uni(events[2 * (nsplits - 1)]) : DSA_INVALID;
  uint32_t dirty = 0;                   // bit per record-cache slot
  uint64_t bb = 0;                      // bit buffer (LSB first)
  uint32_t bcnt = 0, widx = 0, wbase = 0x7FFFFF00u;   // next dword to take, first dword of the LDS window
  uint32_t drop_bits = sym_mis * 8;     // bits of the first word that precede the section
  uint64_t bits_used = (uint64_t)sym_mis * 8;
  bool failed = false;
  uint32_t n_links = 0;                 // opposite links made (each sets two corners)

  // ---- valence traversal: the six lists, and what the machine knows of the two gate vertices
  uint32_t valT1 = 0, valT2 = 0, lmT1 = 0, nvT1 = 0, lmT2 = 0, nvT2 = 0;
  uint32_t active_ctx = 6;              // 6: none yet (the first symbol is E by definition, :77-98)
  uint32_t val_backoff = 0;             // symbols to take one by one before the next attempt at a strip run
  uint32_t cntv = 0, offv = 0, wbv = 0x7FFFFFC0u;   // lane c < 6: symbols left in list c, its offset in ctx_syms, first index of its LDS window
  const uint32_t *ctx_syms = (const uint32_t *)(arena + L.faces);
  if (VAL) {
    uint32_t off_c = 0;
    for (uint32_t c = 0; c < 6; ++c) {
      const uint32_t num = uni(D->val_count[c]);
      if (lane == c) { cntv = num; offv = off_c; }
      if (num && !((uni(D->val_lists_done) >> c) & 1u)) {           // (k_valence_lists has decoded it, as a rule)
        if (!valence_decode_list(D, s, L.stream_len, uni(D->val_off_table[c]), uni(D->val_nsym[c]), uni((uint32_t)D->val_prec[c]), uni(D->val_off_rans[c]),
                                 uni(D->val_size_rans[c]), num, (uint32_t *)(arena + L.faces) + off_c, sh_win)) return;
      }
      off_c += num;
    }
    WAIT_VM0();
    __syncthreads();
  }
  // write back one dirty block of the record cache (all lanes)
  auto rec_writeback = [&](uint32_t slot) {
    uint32_t v = TAG(slot) * 64 + lane;
    if (v < VMAX) vrec[v] = make_uint2(sh_rec[(slot * 64 + lane) * 2], sh_rec[(slot * 64 + lane) * 2 + 1]);
  };
  // make `blk` resident in its slot; fill=false claims the block for freshly created vertices
  auto rec_make_resident = [&](uint32_t blk, bool fill) {
    const uint32_t slot = blk & (CN_REC_BLOCKS - 1);
    if ((dirty >> slot) & 1u) { rec_writeback(slot); dirty &= ~(1u << slot); }
    if (fill) {
      WAIT_VM0();                       // this wave's own record stores must have landed
      uint32_t v = blk * 64 + lane;
      uint2 rr = v < VMAX ? vrec[v] : make_uint2(DSA_INVALID, DSA_INVALID);
      sh_rec[(slot * 64 + lane) * 2] = rr.x; sh_rec[(slot * 64 + lane) * 2 + 1] = rr.y;
      WAIT_VM0();
    }
    if (lane == slot) tagv = blk;
    __syncthreads();
  };
  // Single-word stores are issued by lane 0 only (one `if (lane == 0)` region per symbol): 64 lanes
  // writing one LDS address serialise in the LDS pipeline.  Whether a record / opposite slot lives in
  // LDS or in global memory is decided on uniform values, so the branches inside stay scalar.
  // record store: LDS if the block is resident, else straight to global memory
#define REC_HIT(v_) (TAG(((v_) >> 6) & (CN_REC_BLOCKS - 1)) == ((v_) >> 6))
#define REC_STORE(v_, hit_, lm_, nv_, val_)                                                                   \
  { const uint32_t x_ = VAL ? val_pack((lm_), (val_)) : (lm_);                                                \
    if (hit_) *(uint2 *)&sh_rec[((((v_) >> 6) & (CN_REC_BLOCKS - 1)) * 64 + ((v_) & 63u)) * 2] = make_uint2(x_, (nv_)); \
    vrec[(v_)] = make_uint2(x_, (nv_)); }   /* write-through: global memory is always current */
#define REC_DIRTY(v_, hit_) {}
  // opposite slot of corner c: staged or already in global memory
#define SET_OPP(c_, val_)                                                                \
  { if ((c_) >= 4 * stage_base) R::link_lds(sh_stage, (c_) - 4 * stage_base, (val_));    \
    else R::link(frec, (c_), (val_)); }
  // staged faces [stage_base, upto) -> global memory, 16 bytes per lane and store
  auto flush_stage = [&](uint32_t upto) {
    const uint32_t quads = (upto - stage_base) * (R::WORDS / 4);
    __syncthreads();
    for (uint32_t i = lane; i < quads; i += WAVE) ((uint4 *)frec)[(size_t)stage_base * (R::WORDS / 4) + i] = ((const uint4 *)sh_stage)[i];
    stage_base = upto;
  };
  // everything to global memory, record cache emptied (before lane 0 works on global memory directly)
  auto sync_all = [&]() {
    flush_stage(sid);
    for (uint32_t slot = 0; slot < CN_REC_BLOCKS; ++slot) if ((dirty >> slot) & 1u) rec_writeback(slot);
    dirty = 0;
    tagv = DSA_INVALID;
    WAIT_VM0();
    __syncthreads();
  };
#define CN_FAIL(site) { if (lane == 0) fail(D, ST_INVALID, (site)); failed = true; break; }

#ifdef DSA_LOOP_PROFILE
  uint64_t acc_c = 0, acc_rl = 0, acc_fetch = 0, tp = clk();
  uint32_t prof_run_syms = 0, prof_runs = 0;
  uint32_t n_c = 0, n_rl = 0;
#define PROF(acc, cnt) { uint64_t t_ = clk(); acc += t_ - tp; tp = t_; ++cnt; }
#else
#define PROF(acc, cnt)
#endif
  while (sid < num_symbols) {
    if (sid - stage_base == CN_STAGE) flush_stage(sid);
#ifdef DSA_LOOP_PROFILE
    { uint32_t dummy = 0; PROF(acc_fetch, dummy); }
#endif
#ifndef DSA_NO_CR_RUNS
    // ---------------------------------------------------------------- (C R)^k run, up to 64 pairs per step
    // Symbol bits of "C R" are 0,1,0,1 (LSB first): a run is a string of 0xA nibbles.  Pair j closes old
    // boundary vertex vx_j and creates vertex nv0+j.  vx_{j+1} is the vertex stored behind vx_j's left-most
    // corner; along a regular strip these ids are consecutive, which each lane verifies on its own record.
    // Then every face record, opposite link and vertex record of the run is a closed form of j.
    bool try_run = false;
    uint32_t cand = 0;
    if (!VAL) {
      if (have_top && bcnt >= 16 && ((uint32_t)bb & 0xFFFFu) == 0xAAAAu) {
        // candidate pairs: nibble `lane` lies 4 * lane bits ahead -- in the bit buffer, across its end, or in the LDS window (whose
        // dword widx, bit 0, follows the buffer's last bit), at whatever alignment the symbols before left
        uint32_t nib;
        {
          const uint32_t q = 4 * lane;
          if (q + 4 <= bcnt) nib = (uint32_t)(bb >> q) & 0xFu;
          else {
            const uint32_t have = q < bcnt ? bcnt - q : 0u;              // bits of the nibble still in the buffer (0 .. 3)
            const uint32_t rel = q + have - bcnt;                        // window bit the rest starts at
            const uint32_t wi = widx - wbase + (rel >> 5), sh = rel & 31u;
            const bool in_win = drop_bits == 0 && wi < CN_WIN && (sh + (4u - have) <= 32u || wi + 1 < CN_WIN);
            uint64_t w = 0;
            if (in_win) { w = sh_win[wi]; if (wi + 1 < CN_WIN) w |= (uint64_t)sh_win[wi + 1] << 32; }
            const uint32_t from_win = (uint32_t)(w >> sh);
            nib = in_win ? (((have ? (uint32_t)(bb >> q) : 0u) | (from_win << have)) & 0xFu) : 0u;
          }
        }
        cand = leading_lanes(nib == 0xAu);       // leading 0xA nibbles
        try_run = true;
      }
    } else if (have_top && active_ctx < 6 && val_backoff == 0) {
      // valence mode: which symbols come next is only known once the valences along the strip are (below); the attempt is made
      // when the next symbol is a C (a look at the list's window, nothing consumed)
      const uint32_t left = rdlane(cntv, active_ctx);
      if (left != 0 && ((left - 1u) & ~63u) == rdlane(wbv, active_ctx) && uni(sh_ctx[active_ctx * 64 + ((left - 1u) & 63u)]) == 0u) { cand = WAVE - 1; try_run = true; }
    } else if (VAL && val_backoff) --val_backoff;
    if (try_run) {
      if (2 * cand > num_symbols - sid) cand = (num_symbols - sid) / 2;
      if (cand > VMAX - num_verts) cand = VMAX - num_verts;
      if (splits_left > 0) {               // stop before the symbol that carries the next topology-split event
        const uint32_t sid_evt = num_symbols - 1 - next_src;
        if (sid_evt < sid + 2 * cand) cand = sid_evt > sid ? (sid_evt - sid) / 2 : 0;
      }
      if (cand >= 4) {
        flush_stage(sid);                  // the run writes face records straight to global memory
        const uint32_t f0 = sid, nv0 = num_verts;
        const uint32_t vx0 = T1, va0 = T2;
        // vertex records of the candidate pairs in both directions at once (lane 0 holds vx0's own record in
        // either); the direction is the step to the vertex behind vx0's left-most corner.  (Valence mode loads one record more
        // than it can retire pairs: pair j needs the valence of the vertex it moves the gate to, which is lane j + 1's record.)
        const uint32_t lanes_in = VAL ? cand + 1 : cand;
        const int64_t idp = (int64_t)vx0 + (int64_t)lane, idm = (int64_t)vx0 - (int64_t)lane;
        const bool okp = lane < lanes_in && idp < (int64_t)nv0, okm = lane < lanes_in && idm >= 0 && idm < (int64_t)nv0;
        uint2 rp = make_uint2(DSA_INVALID, DSA_INVALID), rm = rp;
        if (okp) rp = vrec[(uint32_t)idp];
        if (okm) rm = vrec[(uint32_t)idm];
        const int32_t delta = (int32_t)rdlane(rp.y, 0) - (int32_t)vx0;
        uint32_t k = 0;
        uint2 rj = make_uint2(DSA_INVALID, DSA_INVALID);
        uint32_t vxj = 0, lmj = DSA_INVALID;
        if (delta == 1 || delta == -1) {
          const int64_t id = delta == 1 ? idp : idm;
          const bool idok = (delta == 1 ? okp : okm) && (uint32_t)id != va0;
          vxj = (uint32_t)id;
          rj = delta == 1 ? rp : rm;
          lmj = VAL ? VAL_LM(rj.x) : rj.x;
          const uint32_t prev_nv = lane_prev(rj.y);
          // own record must be sane, and the previous pair must hand over exactly this vertex
          bool ok = idok && lmj < 4 * f0 && (lmj & 3u) != 3u && rj.y < nv0 && rj.y != vxj && (lane == 0 || prev_nv == vxj);
          k = leading_lanes(ok);
        }
        // ---- valence mode: are the next 2k symbols really (C R)^k?  Pair j's C comes from the list the valence of vx_j selects
        // (after R_(j-1): its stored valence + 2; pair 0: the list in effect now), its R from the list of vb_j after C_j (stored
        // valence + 1, lane j + 1's record).  How far into its list a symbol lies = how many earlier symbols of the run use that list.
        uint32_t ctxC = 0, ctxR = 0;
        if (VAL && k >= 3) {
          const uint32_t kmax = k - 1;                     // lane k - 1's record only serves as pair k - 2's gate vertex
          const uint32_t val_own = VAL_OF(rj.x), val_next = VAL_OF(dpp_mov<0x130>(rj.x));     // wave_shl:1 -- lane j + 1's word
          { const uint32_t v = val_own + 2u; ctxC = lane == 0 ? active_ctx : (v < 2u ? 2u : (v > 7u ? 7u : v)) - 2u; }
          { const uint32_t v = val_next + 1u; ctxR = (v < 2u ? 2u : (v > 7u ? 7u : v)) - 2u; }
          const uint64_t lt = (1ull << lane) - 1ull, le = lt | (1ull << lane), in = (1ull << kmax) - 1ull;
          uint32_t idxC = DSA_INVALID, idxR = DSA_INVALID;
          for (uint32_t c = 0; c < 6; ++c) {
            const uint64_t mC = __ballot(ctxC == c) & in, mR = __ballot(ctxR == c) & in;
            if (!(mC | mR)) continue;
            const uint32_t left = rdlane(cntv, c), o = rdlane(offv, c);
            const uint32_t nC = (uint32_t)__popcll(mC & lt) + (uint32_t)__popcll(mR & lt), nR = (uint32_t)__popcll(mC & le) + (uint32_t)__popcll(mR & lt);
            if (ctxC == c && nC < left) idxC = o + left - 1u - nC;
            if (ctxR == c && nR < left) idxR = o + left - 1u - nR;
          }
          uint32_t idC = 9, idR = 9;
          if (lane < kmax && idxC != DSA_INVALID) idC = ctx_syms[idxC];
          if (lane < kmax && idxR != DSA_INVALID) idR = ctx_syms[idxR];
          const uint32_t ks = leading_lanes(lane < kmax && idC == 0u && idR == 3u);
          k = ks;
        } else if (VAL) k = 0;
        if (k >= 2) {
          if (lane < k) {
            const uint32_t j = lane;
            const uint32_t fc = f0 + 2 * j, fr = fc + 1;                 // faces of C_j and R_j
            const uint32_t cc = 4 * fc, cr = 4 * fr;                     // their corner 0
            const uint32_t vb = rj.y, cb = qnext(lmj);
            const uint32_t va = j == 0 ? va0 : nv0 + j - 1, nvj = nv0 + j;
            // C_j: face (vx, vb, va), opposites (R_j corner 2, previous face corner 0, cb)
            { FaceIds x = {vxj, vb, va, cr + 2, cc - 4, cb}; R::store(frec, fc, x); }
            // R_j: face (va, vb, nv), opposites (next C corner 1 | open, open, C_j corner 0)
            { FaceIds x = {va, vb, nvj, j + 1 < k ? cr + 4 + 1 : DSA_INVALID, DSA_INVALID, cc}; R::store(frec, fr, x); }
            R::link(frec, cb, cc + 2);                                   // old boundary edge now faces C_j corner 2
            if (j == 0) R::link(frec, cc - 4, cc + 1);                   // the previous top faces C_0 corner 1
            // vertex records after the run (the last SetLeftMostCorner of each vertex wins).  Valence mode: va gained an edge from
            // C_j and one from R_j on top of the 2 it was created with (pair 0: on top of what the gate vertex had)
            vrec[va] = make_uint2(VAL ? val_pack(cr, j == 0 ? valT2 + 2u : 4u) : cr, nvj);   // R_j: left-most corner = R_j corner 0
            if (j + 1 == k) vrec[nvj] = make_uint2(VAL ? val_pack(cr + 2, 2u) : cr + 2, vb);          // later pairs overwrite this for j < k-1
          }
          // keep the LDS copy of touched record blocks coherent: drop them (write-through makes this safe)
          {
            const uint32_t b_lo = (va0 < nv0 ? va0 : nv0) >> 6, b_hi = (nv0 + k) >> 6;
            if (lane < CN_REC_BLOCKS) { const uint32_t t = tagv; if (t != DSA_INVALID && ((t >= (nv0 >> 6) && t <= b_hi) || t == (va0 >> 6))) tagv = DSA_INVALID; }
            (void)b_lo;
          }
          // advance the machine past 2k symbols
          const uint32_t vb_last = rdlane(rj.y, k - 1);
          if (VAL) {
            // the lists give up what the run used; the gate is (vb_last, the newest vertex): vb_last's record is lane k's
            for (uint32_t c = 0; c < 6; ++c) {
              const uint64_t in = (1ull << k) - 1ull;
              const uint32_t used = (uint32_t)__popcll(__ballot(ctxC == c) & in) + (uint32_t)__popcll(__ballot(ctxR == c) & in);
              if (lane == c) cntv -= used;
            }
            const uint32_t gx = rdlane(rj.x, k), gy = rdlane(rj.y, k);
            valT1 = VAL_OF(gx) + 2u; lmT1 = VAL_LM(gx); nvT1 = gy;
            valT2 = 2u; lmT2 = 4 * (f0 + 2 * (k - 1) + 1) + 2; nvT2 = vb_last;
            active_ctx = (valT1 < 2u ? 2u : (valT1 > 7u ? 7u : valT1)) - 2u;
          }
          T1 = vb_last; T2 = nv0 + k - 1;
          num_verts = nv0 + k;
          sid += 2 * k;
          n_links += 3 * k;
#ifdef DSA_LOOP_PROFILE
          prof_run_syms += 2 * k; ++prof_runs;
#endif
          stage_base = sid;
          if (!VAL) {
            bits_used += 4ull * k;
            {                                 // consume 4k bits: first from the buffer, the rest from the window
              uint32_t need = 4 * k;
              if (need <= bcnt) { bb = need >= 64 ? 0 : bb >> need; bcnt -= need; }
              else { need -= bcnt; bb = 0; bcnt = 0; widx += need >> 5; const uint32_t rem = need & 31u; if (rem) { const uint32_t wv = uni(sh_win[widx - wbase]); bb = (uint64_t)(wv >> rem); bcnt = 32 - rem; ++widx; } }
            }
          }
          __syncthreads();                   // the wave's own stores are seen by its later loads: no drain needed
          continue;
        }
        if (VAL) val_backoff = 6;            // not a strip here: the next few symbols go one by one
      }
    }
#endif
    uint32_t val_b3 = 7u;
    if (VAL) {
      if (active_ctx < 6) {                // the next symbol of the list the gate vertex's valence selects, taken from its end (:77-98)
        const uint32_t left = rdlane(cntv, active_ctx);
        if (left == 0) CN_FAIL(642);
        const uint32_t i = left - 1u;
        if ((i & ~63u) != rdlane(wbv, active_ctx)) {          // this list's window: 64 symbols around i
          const uint32_t o = rdlane(offv, active_ctx), at = (i & ~63u) + lane;
          __syncthreads();
          sh_ctx[active_ctx * 64 + lane] = at <= i ? ctx_syms[o + at] : 0u;
          if (lane == active_ctx) wbv = i & ~63u;
          WAIT_VM0();
          __syncthreads();
        }
        const uint32_t id = uni(sh_ctx[active_ctx * 64 + (i & 63u)]);
        if (lane == active_ctx) cntv = i;
        if (id > 4) CN_FAIL(643);
        val_b3 = id == 0 ? 0u : id == 1 ? 1u : id == 2 ? 3u : id == 3 ? 5u : 7u;      // C, S, L, R, E
      }
    } else
    if (bcnt < 3) {                       // refill the bit buffer from the LDS window
      if (widx - wbase >= CN_WIN) {
        __syncthreads();
        sh_win[lane] = sym_words[(size_t)widx + lane];
        wbase = widx;
        WAIT_VM0();
        __syncthreads();
      }
      uint32_t w = uni(sh_win[widx - wbase]);
      ++widx;
      if (drop_bits) { w >>= drop_bits; bb |= (uint64_t)w << bcnt; bcnt += 32 - drop_bits; drop_bits = 0; }
      else { bb |= (uint64_t)w << bcnt; bcnt += 32; }
      if (bcnt < 3) continue;
    }
    // MeshEdgeBreakerTraversalDecoder.cs:89-99: 1 bit, then 2 more unless C
    const uint32_t b3 = VAL ? val_b3 : ((uint32_t)bb & 7u);
    const uint32_t face = sid, corner = 4 * face, ca = corner - 4;
    const uint32_t so = face - stage_base;       // slot of the new face in the staging block
    if ((b3 & 1u) == 0) {                 // C, :247-267
      if (!have_top) CN_FAIL(210);
      const uint32_t vx = T1, va_prev = T2;
      const uint32_t blk = vx >> 6, slot = blk & (CN_REC_BLOCKS - 1);
      if (TAG(slot) != blk) rec_make_resident(blk, true);
      const uint2 rr = *(const uint2 *)&sh_rec[(slot * 64 + (vx & 63u)) * 2];
      const uint32_t lm = VAL ? VAL_LM(uni(rr.x)) : uni(rr.x), vb_next = uni(rr.y);
      const uint32_t cb = qnext(lm);
      if (lm >= corner || (lm & 3u) == 3u || ca == cb || vx == va_prev || vx == vb_next || vb_next >= num_verts) CN_FAIL(213);
      if (!VAL) { bb >>= 1; bcnt -= 1; bits_used += 1; }
      uint32_t nxt_x = 0, nxt_y = 0;              // valence mode: the record of the vertex the gate moves to
      if (VAL) {
        // a face with a vertex twice: the reference's single valence array and the two registers here would part ways
        if (vb_next == va_prev) { if (lane == 0) fail(D, ST_NOTIMPL, DSA_SITE_RETRY_GENERAL); failed = true; break; }
        const uint32_t blk2 = vb_next >> 6, slot2 = blk2 & (CN_REC_BLOCKS - 1);
        if (TAG(slot2) != blk2) rec_make_resident(blk2, true);
        const uint2 r2 = *(const uint2 *)&sh_rec[(slot2 * 64 + (vb_next & 63u)) * 2];
        nxt_x = uni(r2.x); nxt_y = uni(r2.y);
      }
      const bool hit_a = REC_HIT(va_prev);
      if (lane == 0) {
        { FaceIds x = {vx, vb_next, va_prev, DSA_INVALID, ca, cb}; R::store_lds(sh_stage, so, x); }
        SET_OPP(ca, corner + 1);
        SET_OPP(cb, corner + 2);
        REC_STORE(va_prev, hit_a, corner + 2, vb_next, valT2 + 1u);   // SetLeftMostCorner(va_prev, corner + 2) + the vertex before that corner
      }
      REC_DIRTY(va_prev, hit_a);
      T1 = vb_next;                              // face (vx, vb_next, va_prev)
      if (VAL) {                                 // NewActiveCornerReached after C (:100-149): corners 1 and 2 gain an edge
        valT1 = VAL_OF(nxt_x) + 1u; lmT1 = VAL_LM(nxt_x); nvT1 = nxt_y;
        valT2 += 1u; lmT2 = corner + 2; nvT2 = vb_next;
        active_ctx = (valT1 < 2u ? 2u : (valT1 > 7u ? 7u : valT1)) - 2u;
      }
      n_links += 2;
      ++sid;
      PROF(acc_c, n_c);
      continue;
    }
    if (b3 == 1u) {                       // S, :300-343 -- rare: on global memory, lane 0
      if (!have_top) CN_FAIL(230);
      if (!VAL) { bb >>= 3; bcnt -= 3; bits_used += 3; }
      if (VAL && lane == 0) {               // the gate vertices' valences join the others in memory
        const bool h1 = REC_HIT(T1), h2 = REC_HIT(T2);
        REC_STORE(T1, h1, lmT1, nvT1, valT1);
        REC_STORE(T2, h2, lmT2, nvT2, valT2);
      }
      sync_all();
      uint32_t ok = 0, r_sp = sp, r_inv = num_invalid, rT1 = 0, rT2 = 0;
      uint32_t rv1 = 0, rv2 = 0, rl1 = 0, rn1 = 0, rl2 = 0, rn2 = 0;
      if (lane == 0) {
        do {
          uint32_t cb = ca, ca2 = DSA_INVALID, sp2 = sp;
          bool pushed = false;
          if (nsplits) { ca2 = split_map[sid]; pushed = ca2 != DSA_INVALID; }     // topologySplitActiveCorners lookup (:305)
          if (!pushed) { if (sp2 == 0) { fail(D, ST_INVALID, 232); break; } ca2 = stack_mem[--sp2]; }
          if (ca2 >= corner || (ca2 & 3u) == 3u) { fail(D, ST_INVALID, 237); break; }
          if (ca2 == cb || R::get_o(frec, ca2) != DSA_INVALID || R::get_o(frec, cb) != DSA_INVALID) { fail(D, ST_INVALID, 233); break; }
          uint32_t vp = R::get_v(frec, qprev(ca2)), vq = R::get_v(frec, qnext(ca2)), vb_prev = R::get_v(frec, qprev(cb));
          uint32_t cn = qnext(cb);
          uint32_t vn = R::get_v(frec, cn);
          if (vn >= num_verts || vp >= num_verts || vq >= num_verts || vb_prev >= num_verts) { fail(D, ST_INVALID, 234); break; }
          R::link(frec, ca2, corner + 2); R::link(frec, cb, corner + 1);
          { FaceIds x = {vp, vq, vb_prev, DSA_INVALID, cb, ca2}; R::store(frec, face, x); }
          const uint32_t val_n = VAL ? VAL_OF(vrec[vn].x) : 0u, val_b = VAL ? VAL_OF(vrec[vb_prev].x) : 0u;
          vrec[vb_prev] = make_uint2(VAL ? val_pack(corner + 2, val_b) : corner + 2, vq);
          uint32_t lm_n = VAL ? VAL_LM(vrec[vn].x) : vrec[vn].x;
          uint32_t first = cn, guard = 0;
          bool bad = false;
          while (cn != DSA_INVALID) {
            R::set_v(frec, cn, vp);
            // the record of the vertex whose left-most corner follows cn caches the vertex at cn
            uint32_t w = R::get_v(frec, qnext(cn));
            if (w < num_verts && (VAL ? VAL_LM(vrec[w].x) : vrec[w].x) == qnext(cn)) vrec[w].y = vp;
            uint32_t o = R::get_o(frec, qnext(cn));          // SwingLeft
            cn = o == DSA_INVALID ? DSA_INVALID : qnext(o);
            if (cn == first || ++guard > 3 * F) { bad = true; break; }
          }
          if (bad) { fail(D, ST_INVALID, 235); break; }
          if (lm_n >= corner + 4 || (lm_n & 3u) == 3u) { fail(D, ST_INVALID, 238); break; }
          // (valence mode: MergeVertices :151-154, the merged-away vertex's edges become vp's)
          vrec[vp] = make_uint2(VAL ? val_pack(lm_n, VAL_OF(vrec[vp].x) + val_n) : lm_n, R::get_v(frec, qprev(lm_n)));
          vrec[vn] = make_uint2(DSA_INVALID, DSA_INVALID);
          if (remove_invalid) { if (r_inv >= VMAX) { fail(D, ST_INVALID, 236); break; } invalid_list[r_inv++] = vn; }
          r_sp = sp2;                     // the new top replaces the pushed or the exposed entry
          rT1 = vq; rT2 = vb_prev;
          if (VAL) {                      // the new gate: (vq, vb_prev), each one edge richer (NewActiveCornerReached after S)
            const uint2 q1 = vrec[vq], q2 = vrec[vb_prev];
            if (q1.x == DSA_INVALID || q2.x == DSA_INVALID || vq == vb_prev) { fail(D, ST_NOTIMPL, DSA_SITE_RETRY_GENERAL); break; }
            rv1 = VAL_OF(q1.x) + 1u; rl1 = VAL_LM(q1.x); rn1 = q1.y;
            rv2 = VAL_OF(q2.x) + 1u; rl2 = VAL_LM(q2.x); rn2 = q2.y;
          }
          ok = 1;
        } while (0);
        WAIT_VM0();
      }
      if (!uni(ok)) { failed = true; break; }
      sp = uni(r_sp); num_invalid = uni(r_inv); T1 = uni(rT1); T2 = uni(rT2);
      if (VAL) {
        valT1 = uni(rv1); lmT1 = uni(rl1); nvT1 = uni(rn1); valT2 = uni(rv2); lmT2 = uni(rl2); nvT2 = uni(rn2);
        active_ctx = (valT1 < 2u ? 2u : (valT1 > 7u ? 7u : valT1)) - 2u;
      }
      n_links += 2;
      ++sid;
      stage_base = sid;                   // the new face went straight to global memory
      __syncthreads();
      continue;
    }
    if (b3 != 7u) {                       // R (5) / L (3), :268-299
      if (!have_top || num_verts >= VMAX) CN_FAIL(220);
      const uint32_t nv = num_verts;
      if ((nv & 63u) == 0 && !REC_HIT(nv)) rec_make_resident(nv >> 6, false);
      if (!VAL) { bb >>= 3; bcnt -= 3; bits_used += 3; }
      ++num_verts;
      const bool hit_n = REC_HIT(nv), hit_2 = REC_HIT(T2);
      n_links += 1;
      if (b3 == 5u) {
        if (lane == 0) {
          { FaceIds x = {T2, T1, nv, DSA_INVALID, DSA_INVALID, ca}; R::store_lds(sh_stage, so, x); }
          SET_OPP(ca, corner + 2);
          REC_STORE(nv, hit_n, corner + 2, T1, 2u);
          REC_STORE(T2, hit_2, corner, nv, valT2 + 1u);
        }
        REC_DIRTY(nv, hit_n); REC_DIRTY(T2, hit_2);
        if (VAL) { lmT2 = corner + 2; nvT2 = T1; valT2 = 2u; valT1 += 1u; }      // R: corner 0 (the old T2, left behind) +1, corner 1 +1, the new vertex +2
        T2 = nv;                          // face (T2, T1, nv): corner 1 keeps T1
      } else {
        if (lane == 0) {
          { FaceIds x = {T1, nv, T2, DSA_INVALID, ca, DSA_INVALID}; R::store_lds(sh_stage, so, x); }
          SET_OPP(ca, corner + 1);
          REC_STORE(nv, hit_n, corner + 1, T1, 2u);
          REC_STORE(T2, hit_2, corner + 2, nv, valT2 + 1u);
          if (VAL) { const bool h1 = REC_HIT(T1); REC_STORE(T1, h1, lmT1, nvT1, valT1 + 1u); }   // the old T1 leaves the gate: corner 0 +1
        }
        REC_DIRTY(nv, hit_n); REC_DIRTY(T2, hit_2);
        if (VAL) { lmT2 = corner + 2; nvT2 = nv; valT2 += 1u; lmT1 = corner + 1; nvT1 = T1; valT1 = 2u; }      // L: the new vertex (corner 1) +2, corner 2 +1
        T1 = nv;                          // face (T1, nv, T2)
      }
    } else {                              // E, :344-357
      if (num_verts + 3 > VMAX || sp >= VMAX) CN_FAIL(240);
      const uint32_t v0 = num_verts;
      if (!VAL) { bb >>= 3; bcnt -= 3; bits_used += 3; }
      num_verts += 3;
      if (VAL && have_top && lane == 0) {       // the gate that goes onto the stack: its vertices' valences to memory
        const bool g1 = REC_HIT(T1), g2 = REC_HIT(T2);
        REC_STORE(T1, g1, lmT1, nvT1, valT1);
        REC_STORE(T2, g2, lmT2, nvT2, valT2);
      }
      const bool h0 = REC_HIT(v0), h1 = REC_HIT(v0 + 1), h2 = REC_HIT(v0 + 2);
      if (lane == 0) {
        { FaceIds x = {v0, v0 + 1, v0 + 2, DSA_INVALID, DSA_INVALID, DSA_INVALID}; R::store_lds(sh_stage, so, x); }
        REC_STORE(v0, h0, corner, v0 + 2, 2u);
        REC_STORE(v0 + 1, h1, corner + 1, v0, 2u);
        REC_STORE(v0 + 2, h2, corner + 2, v0 + 1, 2u);
        if (have_top) stack_mem[sp] = ca;
      }
      if (VAL) { lmT1 = corner + 1; nvT1 = v0; valT1 = 2u; lmT2 = corner + 2; nvT2 = v0 + 1; valT2 = 2u; }      // E: every corner +2
      REC_DIRTY(v0, h0); REC_DIRTY(v0 + 1, h1); REC_DIRTY(v0 + 2, h2);
      if (have_top) ++sp;
      have_top = true;
      T1 = v0 + 1; T2 = v0 + 2;
    }
    ++sid;
    if (VAL) active_ctx = (valT1 < 2u ? 2u : (valT1 > 7u ? 7u : valT1)) - 2u;
    PROF(acc_rl, n_rl);
    if (splits_left > 0) {                // :363-375 (sid already advanced: the symbol just decoded is sid-1)
      const uint32_t enc_id = num_symbols - sid;
      if (next_src > enc_id) CN_FAIL(243);   // encoderSplitSymbolId < 0 in the reference
      if (next_src == enc_id) {
        uint32_t r_left = splits_left, r_next = DSA_INVALID, ok = 1;
        if (lane == 0) {
          while (r_left > 0) {
            uint32_t source = events[2 * (r_left - 1)];
            uint32_t packed = events[2 * (r_left - 1) + 1];
            if (source != enc_id) break;
            --r_left;
            uint32_t edge = packed >> 31, enc_split = packed & 0x7FFFFFFFu;
            if (enc_split >= num_symbols) { ok = 0; break; }
            uint32_t nc = edge == 1 ? corner + 1 : corner + 2;   // Next / Previous of the new top (1 = RightFaceEdge)
            uint32_t key = num_symbols - enc_split - 1;
            split_map[key] = nc;                                  // dictionary semantics: overwrite
          }
          r_next = r_left ? events[2 * (r_left - 1)] : DSA_INVALID;
          WAIT_VM0();
        }
        if (!uni(ok)) CN_FAIL(244);
        splits_left = uni(r_left); next_src = uni(r_next);
      }
    }
  }
  if (failed) return;
  if (!VAL && bits_used > bit_end) { if (lane == 0) fail(D, ST_INVALID, 246); return; }   // symbols ran past their section
  sync_all();
  const uint64_t t_loop = clk();

  // ---- start faces (:378-415) and isolated-vertex compaction (:417-441): lane 0 on global memory
  uint32_t final_nv = 0;
  if (lane == 0) {
    do {
      Rabs start_faces;
      uint32_t endp;
      start_faces.start(s, L.stream_len, D->off_start_faces, &endp);
      if (!start_faces.ok) { fail(D, ST_INVALID, 201); break; }
      uint32_t num_faces = num_symbols;
      bool top_pending = have_top, bad = false;
      uint32_t spx = sp;
      while (top_pending || spx > 0) {
        uint32_t corner;
        if (top_pending) { corner = 4 * (num_symbols - 1); top_pending = false; }
        else corner = stack_mem[--spx];
        bool interior = start_faces.next() != 0;
        if (!interior) continue;
        if (num_faces >= F || corner >= 4 * num_faces) { fail(D, ST_INVALID, 251); bad = true; break; }
        uint32_t ca = corner;
        uint32_t vn = R::get_v(frec, qnext(ca));
        const uint32_t lm_vn = VAL ? VAL_LM(vrec[vn < num_verts ? vn : 0].x) : vrec[vn < num_verts ? vn : 0].x;
        if (vn >= num_verts || lm_vn >= 4 * num_faces) { fail(D, ST_INVALID, 252); bad = true; break; }
        uint32_t cb = qnext(lm_vn);
        uint32_t vx = R::get_v(frec, qnext(cb));
        const uint32_t lm_vx = VAL ? VAL_LM(vrec[vx < num_verts ? vx : 0].x) : vrec[vx < num_verts ? vx : 0].x;
        if (vx >= num_verts || lm_vx >= 4 * num_faces) { fail(D, ST_INVALID, 253); bad = true; break; }
        uint32_t cc = qnext(lm_vx);
        if (ca == cb || ca == cc || cb == cc) { fail(D, ST_INVALID, 254); bad = true; break; }
        if (R::get_o(frec, ca) != DSA_INVALID || R::get_o(frec, cb) != DSA_INVALID || R::get_o(frec, cc) != DSA_INVALID) { fail(D, ST_INVALID, 255); bad = true; break; }
        uint32_t vp = R::get_v(frec, qnext(cc));
        if (vp >= num_verts) { fail(D, ST_INVALID, 262); bad = true; break; }
        uint32_t face = num_faces++;
        uint32_t nc = 4 * face;
        R::link(frec, ca, nc); R::link(frec, cb, nc + 1); R::link(frec, cc, nc + 2);
        { FaceIds x = {vx, vp, vn, ca, cb, cc}; R::store(frec, face, x); }
      }
      if (bad) break;
      if (num_faces != F) { fail(D, ST_INVALID, 256); break; }
      D->interior_corners = 2 * (n_links + 3 * (num_faces - num_symbols));
      uint32_t nvert = num_verts;
      for (uint32_t k = 0; k < num_invalid && !bad; ++k) {
        uint32_t inv = invalid_list[k];
        if (nvert == 0) { fail(D, ST_INVALID, 257); bad = true; break; }
        uint32_t src = nvert - 1;
        while (vrec[src].x == DSA_INVALID) { if (nvert <= 1) { bad = true; break; } src = --nvert - 1; }
        if (bad) { fail(D, ST_INVALID, 258); break; }
        if (src < inv) continue;
        uint32_t start = VAL ? VAL_LM(vrec[src].x) : vrec[src].x, c = start, guard = 0;
        bool left = true;
        while (c != DSA_INVALID) {   // VertexCornersIterator (D-10: starts at the left-most corner itself)
          if (c >= 4 * F || (c & 3u) == 3u || R::get_v(frec, c) != src || ++guard > 3 * F) { fail(D, ST_INVALID, 259); bad = true; break; }
          R::set_v(frec, c, inv);
          if (left) {
            uint32_t o = R::get_o(frec, qnext(c));
            c = o == DSA_INVALID ? DSA_INVALID : qnext(o);
            if (c == DSA_INVALID) { uint32_t o2 = R::get_o(frec, qprev(start)); c = o2 == DSA_INVALID ? DSA_INVALID : qprev(o2); left = false; }
            else if (c == start) c = DSA_INVALID;
          } else {
            uint32_t o = R::get_o(frec, qprev(c));
            c = o == DSA_INVALID ? DSA_INVALID : qprev(o);
          }
        }
        vrec[inv] = vrec[src];
        vrec[src] = make_uint2(DSA_INVALID, DSA_INVALID);
        nvert--;
      }
      if (bad) break;
      final_nv = remove_invalid ? nvert : num_verts;
      D->num_vertices = final_nv;
    } while (0);
    WAIT_VM0();
    __threadfence_block();
  }
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  const uint32_t NV = uni(final_nv);
  const uint32_t NVALL = num_verts;
  const uint64_t t_tail = clk();

  const uint64_t t_sym = clk();
  const uint32_t nad = D->num_att_data;
  const uint64_t t_seam = clk();
  // ---- vertex -> point id (AssignPointsToCorners, :537-638, seam-free case):
  // single connectivity: point == vertex; per-attribute connectivity: rank among vertices that own a corner.
  uint32_t *vrank = (uint32_t *)(arena + L.vrank);
  if (nad == 0) {
    for (uint32_t v = lane; v < NV; v += WAVE) vrank[v] = v;
    if (lane == 0) D->num_points = NV;
  } else {
    uint32_t base = 0;
    for (uint32_t v0 = 0; v0 < NVALL; v0 += WAVE) {
      uint32_t v = v0 + lane;
      uint32_t has = (v < NVALL && vrec[v].x != DSA_INVALID) ? 1u : 0u;
      uint64_t m = __ballot(has);
      uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (v < NVALL) vrank[v] = base + before;
      base += (uint32_t)__popcll(m);
    }
    if (lane == 0) D->num_points = base;
  }
  // ---- IsOnBoundary per vertex (CornerTable.cs:174-178) for the traversal: bit1 of the vertex flag.  A record load, then a
  // gather that depends on it: four vertices per lane in flight.
  {
    uint8_t *vflag = arena + L.vvis;
    for (uint32_t v0 = 0; v0 < NVALL; v0 += 4 * WAVE) {
      uint32_t lm[4], oo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const uint32_t v = v0 + u * WAVE + lane; lm[u] = v < NVALL ? vrec[v].x : DSA_INVALID; if (VAL) lm[u] = VAL_LM(lm[u]); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = lm[u] != DSA_INVALID && lm[u] < 4 * F && (lm[u] & 3u) != 3u;
        oo[u] = ok ? R::get_o(frec, qnext(lm[u])) : 0u;          // (not a corner: not on the boundary)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const uint32_t v = v0 + u * WAVE + lane; if (v < NVALL) vflag[v] = oo[u] == DSA_INVALID ? 2 : 0; }
    }
  }
  if (lane == 0) {
    D->dbg[0] = (uint32_t)(t_loop - t_start); D->dbg[1] = (uint32_t)(t_tail - t_loop); D->dbg[2] = (uint32_t)(t_sym - t_tail);
    D->dbg[3] = (uint32_t)(t_seam - t_sym); D->dbg[4] = (uint32_t)(clk() - t_seam);
    D->dbg[13] = (uint32_t)(clk() - t_start); D->dbg[14] = (uint32_t)r_start; D->dbg[15] = (uint32_t)(realclk() - r_start);
    D->num_all_vertices = NVALL;
#ifdef DSA_LOOP_PROFILE
    // (slots the traversal half of k_chain does not overwrite)
    D->dbg[10] = prof_run_syms | (prof_runs << 20); D->dbg[11] = n_c | (n_rl << 16);
    D->dbg[12] = (uint32_t)(acc_c / (n_c ? n_c : 1)); D->dbg[18] = (uint32_t)(acc_rl / (n_rl ? n_rl : 1));
    D->dbg[19] = (uint32_t)(acc_fetch / (n_c + n_rl + 1));
#endif
  }
#undef CN_FAIL
#undef TAG
#undef REC_HIT
#undef REC_STORE
#undef REC_DIRTY
#undef SET_OPP
}
#undef VAL_LM
#undef VAL_OF

__global__ __launch_bounds__(WAVE) void k_connectivity(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  __shared__ __attribute__((aligned(16))) uint32_t sh[CN_LDS_WORDS];
  __builtin_amdgcn_s_setprio(DSA_CHAIN_PRIO);   // critical path: issue ahead of the entropy-decode waves sharing the CU
  const uint32_t mesh = blockIdx.x;
  if (mesh >= n) return;
  uint32_t *sh_rec = sh + CN_STAGE * 8, *sh_win = sh_rec + CN_REC_BLOCKS * 64 * 2, *sh_ctx = sh_win + CN_WIN;
  if (!layouts[mesh].rec_compact) connectivity_wave<false, false>(arena, layouts[mesh], &descs[mesh], sh, sh_rec, sh_win, sh_ctx);
  else if (descs[mesh].traversal_type == 2) connectivity_wave<true, true>(arena, layouts[mesh], &descs[mesh], sh, sh_rec, sh_win, sh_ctx);
  else connectivity_wave<true, false>(arena, layouts[mesh], &descs[mesh], sh, sh_rec, sh_win, sh_ctx);
}

// =========================================================================
// k_conn_checks: attribute seams (MeshEdgeBreakerDecoder.cs:502-535), on the third stream.  The fast kernels decode seam-free
// attribute connectivity only: the position of the first set seam bit goes to k_seal, which sends such a mesh to the general path.
// =========================================================================
__global__ __launch_bounds__(WAVE) void k_conn_checks(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t lanes_per_mesh) {
  // one lane per (mesh, attribute data): the bit-serial rABS decode has no cross-lane traffic, so a wave
  // carries 64 / lanes_per_mesh meshes and the instruction stream is shared between them
  const uint32_t lane = lane_id();
  const uint32_t mesh = blockIdx.x * (WAVE / lanes_per_mesh) + lane / lanes_per_mesh, d = lane % lanes_per_mesh;
  if (mesh >= n) return;
  const MeshLayout &L = layouts[mesh];
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->encoder_type == 0 || D->general || d >= D->num_att_data) return;
  // the seam bits of a mesh with corner attributes gate its seam tables: a hundred thousand dependent steps on a handful of waves,
  // which must not wait twenty cycles for every issue slot behind the entropy decoders
  if (L.seam_bytes) __builtin_amdgcn_s_setprio(DSA_CHAIN_PRIO);
  const uint8_t *s = arena + L.stream;
  // One seam bit per interior edge and attribute data.  How many edges are interior is the connectivity's result, which this
  // kernel does not wait for: it looks for the first set bit among as many bits as a mesh of F faces can have (3F / 2) and
  // k_seal compares its position with the number of edges (decoding past the coded bits yields arbitrary bits: harmless).
  const uint32_t edges = (uint32_t)(((uint64_t)D->num_faces * 3) / 2);
  Rabs rb;
  uint32_t endp;
  rb.start(s, L.stream_len, D->off_seams[d], &endp);
  if (!rb.ok) { fail(D, ST_INVALID, 260); return; }
  // AnsDecoder.cs:42-56, restructured so that the common case (no renormalisation) is a load-free loop
  uint32_t state = rb.state, off = rb.off, first = DSA_INVALID;
  const uint32_t p = rb.p;
  uint32_t i = 0;
  // A mesh with corner-attribute decoders (seam scratch): every bit is wanted -- set bits are OR-ed into the attribute data's bit
  // array (zeroed by k_init; this lane is its only writer), k_seam_tables hands them to the edges once the connectivity is there.
  uint32_t *store = nullptr;
  if (L.seam_bytes) {
    const SeamLayout g = seam_layout(L.cap_faces, L.cap_vertices, D->num_att_data, L.rec_compact != 0);
    store = (uint32_t *)(seam_block(arena, L, g, d) + g.bits);
  }
  if (store) {       // (every lane of the wave at the same bit: see rabs_block_to_words)
    D->seam_first[d] = rabs_block_to_words<false>(rb, edges, store);
    return;
  }
  while (i < edges) {
    if (state < 4096 && off > 0) state = state * 256 + rb.buf[--off];
    if (p <= 16 && state >= 8192 && i + 8 <= edges) {
      // eight zero bits shrink the state by at most (15/16)^8 > 1/2, so none of them renormalises; only the zero-bit
      // successor is computed, and a group with a set bit is decoded again bit by bit
      const uint32_t state0 = state;
      uint32_t any = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t quot = state >> 8, rem = state & 255u;
        any |= rem < p ? 1u : 0u;
        state = state - quot * p - p;
      }
      if (!any) { i += 8; continue; }
      state = state0;
    }
    const uint32_t quot = state >> 8, rem = state & 255u, xn = quot * p;
    const bool val = rem < p;
    if (val) { first = i; break; }
    state = state - xn - p;
    ++i;
  }
  D->seam_first[d] = first;
}

// =========================================================================
// k_traverse: depth-first attribute sequencing on the position corner table
// (DepthFirstTraverser.cs:9-99 + MeshAttributeIndicesEncodingObserver.cs:14-21 +
// MeshTraversalSequencer.cs:13-31).  One wave per mesh, wave-uniform DFS state.
//
// The DFS mostly marches along triangle strips whose faces were created consecutively by the
// connectivity decoder, so a step first *speculates* that the next faces are g0, g0+d, g0+2d, ...
// (d = +-1): lane i loads face record g0+i*d, derives the corner it would be entered through, and
// evaluates the reference's decision at its own element (tip new & interior -> right; tip visited ->
// the single open side) against the faces/vertices visited before the step or earlier in the run.
// The longest prefix of elements whose decision leads exactly to the next guessed face is retired at
// once (face marks, new-vertex numbering by ballot prefix, coalesced d2c stores); everything else --
// pushes, pops, boundary tips, irregular turns -- takes the scalar step, which is the reference's
// loop body verbatim.  Either way each step is the sequential algorithm's result.
// =========================================================================
// Scratch initialisation for the traversal (no dependence on the stream contents: runs first):
// face-visited marks, vertex_to_data = -1.
__global__ __launch_bounds__(256) void k_init(uint8_t *arena, const MeshLayout *layouts, uint32_t n) {
  uint32_t mesh = blockIdx.y;
  if (mesh >= n) return;
  const MeshLayout &L = layouts[mesh];
  int32_t *v2d = (int32_t *)(arena + L.v2d);
  uint32_t *fvis4 = (uint32_t *)(arena + L.fvis);
  const uint32_t F = L.cap_faces, V = L.cap_vertices;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (uint32_t w4 = tid; w4 < (F + 3) / 4; w4 += stride) fvis4[w4] = 0;     // regions are 256-byte padded
  for (uint32_t v = tid; v < V; v += stride) v2d[v] = -1;
  if (L.cap_splits) {                    // topologySplitActiveCorners as a direct map: decoder symbol id -> corner
    uint32_t *split_map = (uint32_t *)(arena + L.fstamp);
    for (uint32_t f = tid; f < F; f += stride) split_map[f] = DSA_INVALID;
  }
  if (L.seam_bytes) {                    // fast seam path: per attribute data the seam bits (set bits are OR-ed in), the face marks and
                                         // vertex -> entry of the attribute's traversal; per mesh the "touches a seam" flags
    const uint32_t nad = (uint32_t)((L.seam_bytes - seam_layout(F, V, 0, L.rec_compact != 0).total) / seam_layout(F, V, 1, L.rec_compact != 0).data_stride);
    const SeamLayout g = seam_layout(F, V, nad, L.rec_compact != 0);
    uint32_t *vseam4 = (uint32_t *)(arena + L.seam + g.vseam), *eseam4 = (uint32_t *)(arena + L.seam + g.eseam);
    for (uint32_t w4 = tid; w4 < (V + 3) / 4; w4 += stride) vseam4[w4] = 0;
    for (uint32_t f = tid; f < F; f += stride) eseam4[f] = 0;           // seam masks: k_seam_tables writes the non-zero bytes only
    for (uint32_t d = 0; d < nad; ++d) {
      uint8_t *blk = seam_block(arena, L, g, d);
      uint32_t *bits = (uint32_t *)(blk + g.bits), *fv4 = (uint32_t *)(blk + g.fvis);
      int32_t *av2d = (int32_t *)(blk + g.v2d);
      for (uint32_t w = tid; w < (3 * F / 2 + 31) / 32 + 4; w += stride) bits[w] = 0;
      for (uint32_t w4 = tid; w4 < (F + 3) / 4; w4 += stride) fv4[w4] = 0;
      for (uint32_t v = tid; v < 3 * F; v += stride) av2d[v] = -1;
    }
  }
}

template <bool CP>
__device__ __forceinline__ void para_operands_flat(uint32_t p, const uint32_t *frec, const uint32_t *d2c, const int32_t *v2d, uint32_t F, uint32_t NV,
                                                   uint32_t &en, uint32_t &ep, uint32_t &eo);
__global__ __launch_bounds__(256) void k_para_operands(uint8_t *arena, const MeshLayout *layouts, const MeshDesc *descs, uint32_t n) {
  uint32_t mesh = blockIdx.y;
  if (mesh >= n) return;
  const MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->encoder_type == 0 || D->general) return;   // point clouds have no connectivity; general meshes: k_general
  const MeshLayout &L = layouts[mesh];
  const uint32_t *frec = (const uint32_t *)(arena + L.frec);
  const uint32_t *d2c = (const uint32_t *)(arena + L.d2c);
  const int32_t *v2d = (const int32_t *)(arena + L.v2d);
  uint32_t *para = (uint32_t *)(arena + L.para);
  const uint32_t entries = D->num_entries, F = D->num_faces, NV = D->num_vertices;
  const bool compact = L.rec_compact != 0;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < entries; p += gridDim.x * blockDim.x) {
    uint32_t en, ep, eo;
    if (compact) para_operands_flat<true>(p, frec, d2c, v2d, F, NV, en, ep, eo);
    else para_operands_flat<false>(p, frec, d2c, v2d, F, NV, en, ep, eo);
    para[3 * p] = en; para[3 * p + 1] = ep; para[3 * p + 2] = eo;
  }
}

// k_traverse speculation: from a corner whose tip is new and interior the DFS usually repeats the pair
// "tip new & interior -> right (N), then tip visited & right side done -> left (L)", i.e. it continues at
//   succ(a) = Opposite(Previous(Opposite(Next(a)))).
// The candidate path a_0, a_1 = succ(a_0), ... is extrapolated (the corner ids along a regular strip or
// across the rings of a spiral follow a constant or linearly changing step; the step parameters carry
// over from the previous run, otherwise three exact hops seed them) and each lane verifies its own link
// succ(a_(i-1)) == a_i from the face records it loads anyway; then lane i evaluates the
// reference's two decisions of pair i against the faces / vertices visited before the step or earlier in
// the run (exact membership through atomicMin stamps), and the leading pairs that all check out are
// retired at once.  Everything else is the scalar step = the reference's loop body.
#define TR_PAIRS 64

#define TR_SLOT_BITS 8
#define TR_SLOTS (1u << TR_SLOT_BITS)
struct Triple { uint32_t a, b, c; };
// para_operands_of (dsa_common.h) without branches: every load happens, at a clamped index when its guard is false.
template <bool CP>
__device__ __forceinline__ void para_operands_flat(uint32_t p, const uint32_t *frec, const uint32_t *d2c, const int32_t *v2d, uint32_t F, uint32_t NV,
                                                   uint32_t &en, uint32_t &ep, uint32_t &eo) {
  typedef Rec<CP> R;
  const uint32_t c0 = d2c[p];
  const bool ok0 = p > 0 && c0 < 4 * F && (c0 & 3u) != 3u;
  const uint32_t oci = R::get_o_plain(frec, ok0 ? c0 : 0u);
  const bool ok1 = ok0 && oci != DSA_INVALID && oci < 4 * F && (oci & 3u) != 3u;
  const uint4 fr = R::vertices_of(frec, ok1 ? (oci >> 2) : 0u);
  const uint32_t k = oci & 3u;
  const uint32_t a = k == 0 ? fr.x : (k == 1 ? fr.y : fr.z), b = k == 0 ? fr.y : (k == 1 ? fr.z : fr.x), c = k == 0 ? fr.z : (k == 1 ? fr.x : fr.y);
  const bool ok2 = ok1 && a < NV && b < NV && c < NV;
  const int32_t vo = v2d[ok2 ? a : 0u], vn = v2d[ok2 ? b : 0u], vp = v2d[ok2 ? c : 0u];
  const bool ok3 = ok2 && vo >= 0 && vn >= 0 && vp >= 0 && (uint32_t)vo < p && (uint32_t)vn < p && (uint32_t)vp < p;
  en = ok3 ? (uint32_t)vn : DSA_INVALID; ep = ok3 ? (uint32_t)vp : 0u; eo = ok3 ? (uint32_t)vo : 0u;
}

// The body of k_traverse for one mesh on one wave; LDS: sh_tf[TR_SLOTS], sh_tv[TR_SLOTS] (in-run membership of faces / tips) and
// sh_hist[TR_HIST_WORDS] (step history), zeroed by the caller.  Also the second half of k_chain.
//
// A step of the speculative traversal is bound by dependent memory round trips (0.5 - 1 us each on the busy chip) and by the
// LDS membership tables, so the loop is built to need few of either:
//   * "fast" attempt (along a strip): everything a pair needs -- the record at a, the record at b = Opposite(Next(a)), the visited
//     marks of both faces, of the faces behind their other edges and of both tips -- is loaded in ONE round trip at extrapolated
//     ids (six progressions: a, b, the two tips, the corner left of a, the corner right of b); each lane then checks every
//     extrapolated id against what the records it loaded say, and only lanes whose ids all agree take part.  The progressions
//     continue the previous run, or, at the start of a side, combine the exact ids of the first pair with the steps the same
//     direction had the last time (step history keyed by the step of a: a spiral has four directions).
//   * when all six progressions are linear, "which earlier pair of this run holds face / tip X" is arithmetic
//     ((X - first) / step is an integer below the run length) instead of a search in the LDS tables.
//   * dependent attempt (no history for the direction): exact hops seed the path, then records of a, records of b and the marks
//     in three round trips, as the data dependences dictate; membership through the tables.
//   * the pair a run ends on was loaded and judged like every other: its two elements go to the scalar step with their record
//     and their marks (as they are after the retired pairs), so a turn of the spiral costs no load until the next side's seed.
#define TR_HIST_WORDS 32
#define TR_NONE 0xFFFFFFFFu
// The tables a traversal runs on: the position corner table of a mesh (trav_position) or the corner table of one of its attributes,
// cut along the attribute's seams (trav_attribute: the "virtual mesh" k_seam_tables built -- records whose vertices are the
// attribute's vertices and whose opposites end at the seams, so that DepthFirstTraverser.cs:9-99 over MeshAttributeCornerTable.cs is
// the same wave program as over CornerTable.cs).
struct TravIO {
  const uint32_t *frec; uint32_t *d2c; int32_t *v2d; uint8_t *fvis, *vflag; uint32_t *stack, *para;
  uint32_t F, NV, cap_entries, expect;     // expect: the entries a valid stream yields (one per encoded vertex / per attribute vertex)
  int att_data;                            // -1: the position table
};
__device__ __forceinline__ TravIO trav_position(uint8_t *arena, const MeshLayout &L, const MeshDesc *D) {
  TravIO io;
  io.frec = (const uint32_t *)(arena + L.frec); io.d2c = (uint32_t *)(arena + L.d2c); io.v2d = (int32_t *)(arena + L.v2d);
  io.fvis = arena + L.fvis; io.vflag = arena + L.vvis; io.stack = (uint32_t *)(arena + L.fstamp); io.para = (uint32_t *)(arena + L.para);
  io.F = D->num_faces; io.NV = D->num_vertices; io.cap_entries = L.cap_vertices; io.expect = D->num_enc_vertices; io.att_data = -1;
  return io;
}
__device__ __forceinline__ TravIO trav_attribute(uint8_t *arena, const MeshLayout &L, const MeshDesc *D, uint32_t d) {
  const SeamLayout g = seam_layout(L.cap_faces, L.cap_vertices, D->num_att_data, L.rec_compact != 0);
  uint8_t *blk = seam_block(arena, L, g, d);
  TravIO io;
  io.frec = (const uint32_t *)(blk + g.rec); io.d2c = (uint32_t *)(blk + g.d2c); io.v2d = (int32_t *)(blk + g.v2d);
  io.fvis = blk + g.fvis; io.vflag = blk + g.vflag; io.stack = (uint32_t *)(blk + g.stack); io.para = (uint32_t *)(blk + g.para);
  io.F = D->num_faces; io.NV = D->seam_nv[d]; io.cap_entries = 3u * L.cap_faces; io.expect = D->seam_nv[d]; io.att_data = (int)d;
  return io;
}
template <bool CP>
__device__ __forceinline__ void traverse_wave(uint8_t *arena, const MeshLayout &L, MeshDesc *D, const TravIO &io, uint32_t fuse_operands, unsigned long long *sh_tf,
                                              unsigned long long *sh_tv, uint32_t *sh_hist) {
  typedef Rec<CP> R;
  typedef typename R::Raw Raw;
  if (status_of(D) != ST_OK || D->encoder_type == 0 || D->general) return;   // point clouds have no connectivity; general meshes: k_general
  const uint32_t *frec = io.frec;
  uint32_t *d2c = io.d2c;
  int32_t *v2d = io.v2d;
  uint8_t *fvis = io.fvis;
  uint8_t *vflag = io.vflag;            // bit0 visited, bit1 on boundary
  const uint32_t F = uni(io.F), NV = uni(io.NV), cap_entries = uni(io.cap_entries), expect = uni(io.expect);
  const bool position = io.att_data < 0;
  // DFS stack: the topology-split map of k_connectivity is dead by now (the faces output is being written by k_faces
  // meanwhile).  Only a face with two open sides pushes, and the last face cannot, so F entries always suffice.
  uint32_t *stack = io.stack;
  const uint32_t stack_cap = F;
  const uint32_t lane = lane_id();
  const uint64_t t_start = clk(), r_start = realclk();

  uint32_t count = 0, sp = 0, f_scan = 0;
  uint32_t run_id = 0;        // stamps of newer runs compare smaller, so atomicMin always replaces older ones
  // Ids the next fast attempt loads at, per lane (32-bit wrap-around arithmetic: a wrong value is merely a wrong guess, every id
  // is checked against the records): a of this pair and of the next, b, the tips, the corner left of a, the corner right of b
  bool have_prog = false, prog_lin = false;
  uint32_t p_a = 0, p_an = 0, p_b = 0, p_ta = 0, p_tb = 0, p_la = 0, p_rb = 0;
  // Step history: one entry per direction of a spiral (the turns counted modulo 4), holding the step of a it was learned with and
  // the steps of the other five ids; an entry whose steps stop a run at its second pair twice in a row is forgotten.
  uint32_t dir = 0, hist_strikes = 0;
  uint32_t backoff = 0;       // scalar steps to take before speculating again
  uint32_t fail_streak = 0;   // attempts in a row that retired nothing
  // Pairs the next attempt loads and checks (its memory traffic is proportional to it).  A run that ended on a turn
  // of the spiral predicts the following sides: they grow by one pair per ring, and sides cut by the boundary
  // alternate, so the window is the longer of the last two such runs plus a margin; anything else opens it fully.
  uint32_t window = WAVE, side1 = WAVE, side2 = WAVE;
  uint32_t n_run = 0, n_run_faces = 0, n_scalar = 0, n_fail = 0, n_fast = 0;
  bool failed = false;
  // What is already known of the element the DFS stands on (handed over by the attempt that ended on it): its record, its marks
  // as they are now; and the same for the element behind its right edge, should the step go there.  bits: 1 tip new, 2 tip on
  // the boundary, 4 right side done, 8 left side done
  bool have_rec = false, have_state = false, must_scalar = false, have2 = false;
  bool no_hist = false;       // the step history was tried from this element and was wrong: exact hops
  uint32_t c_v = 0, c_rc = 0, c_lc = 0, c_bits = 0;
  uint32_t c2_v = 0, c2_rc = 0, c2_lc = 0, c2_bits = 0;
#ifdef DSA_TRAV_PROFILE
  // shader clocks by phase: [0] fast loads (issue -> ids checked), [1] element inputs + seed, [2] dependent hops + loads, [3] membership + verdict,
  // [4] retirement + progressions, [5] scalar step
  uint64_t tp_acc[6] = {0, 0, 0, 0, 0, 0}, tp_last = clk();
  uint32_t np_fast_hit = 0, np_dep = 0, np_head = 0, np_hist = 0, np_lin = 0, np_hand = 0;
#define TPROF(slot_) { const uint64_t t_ = clk(); tp_acc[slot_] += t_ - tp_last; tp_last = t_; }
#define TCOUNT(x_) ++x_
#else
#define TPROF(slot_)
#define TCOUNT(x_)
#endif
#define TR_FAIL(site) { if (lane == 0) fail(D, ST_INVALID, (site)); failed = true; break; }
#define VISIT_SCALAR(v_, c_, fl_) { if (lane == 0) { vflag[v_] = (uint8_t)((fl_) | 1u); d2c[count] = (c_); v2d[v_] = (int32_t)count; } ++count; }
  auto corner_ok = [&](uint32_t c) -> bool { return c < 4 * F && (c & 3u) != 3u; };

  for (;;) {
    if (sp == 0) {
      // next traversal start: first unvisited face at or after f_scan (MeshTraversalSequencer.cs:22-29)
      uint32_t found = DSA_INVALID;
      while (f_scan < F) {
        uint32_t f = f_scan + lane;
        uint64_t m = __ballot(f < F && fvis[f] == 0);
        if (m) { found = f_scan + (uint32_t)__builtin_ctzll(m); break; }
        f_scan += WAVE;
      }
      if (found == DSA_INVALID) break;
      f_scan = found;
      const uint32_t corner = 4 * found;
      if (lane == 0) stack[0] = corner;
      sp = 1;
      // DepthFirstTraverser.cs:17-30: the two vertices of the start edge
      const uint4 vv = R::vertices_of(frec, found);
      const uint32_t nv = uni(vv.y), pv = uni(vv.z);       // Next(corner 0) = corner 1, Previous = corner 2
      if (nv >= NV || pv >= NV) TR_FAIL(300);
      { uint32_t uni_flag = uni((uint32_t)vflag[nv]); if (!(uni_flag & 1u)) { if (count >= cap_entries) TR_FAIL(302); VISIT_SCALAR(nv, corner + 1, uni_flag); } }
      WAIT_VM0();
      { uint32_t uni_flag = uni((uint32_t)vflag[pv]); if (!(uni_flag & 1u)) { if (count >= cap_entries) TR_FAIL(302); VISIT_SCALAR(pv, corner + 2, uni_flag); } }
      WAIT_VM0();
    }
    uint32_t corner = uni(stack[sp - 1]);
    have_prog = false; have_rec = false; have_state = false; must_scalar = false; have2 = false;
    if (corner == DSA_INVALID || corner >= 4 * F || uni((uint32_t)fvis[corner >> 2])) { --sp; continue; }

    for (;;) {   // DepthFirstTraverser.cs:39-97 inner loop
      uint32_t face = corner >> 2;
      // inputs of the scalar step
      uint32_t v = 0, rc = 0, lc = 0, bits = 0;
      // (tip, corner right, corner left) of the corners behind this element's two other edges, when their records were loaded with its marks
      bool have_kids = false;
      uint32_t kr_v = 0, kr_rc = 0, kr_lc = 0, kl_v = 0, kl_rc = 0, kl_lc = 0;
      // per-lane state of an attempt (pair `lane`: N element at a, face A; L element at b = Opposite(Next(a)), face B)
      uint32_t kind = 0;                 // 0 none, 1 fast attempt at the carried progressions, 2 fast attempt from the step history, 3 dependent attempt
      bool lin = false;      // (read by the DSA_TRAV_TRACE build only)
      (void)lin;
      uint32_t a = 0, b = 0, tipA = 0, lcA = DSA_INVALID, tipB = 0, rcB = DSA_INVALID, lcB = DSA_INVALID;
      // marks before the step, packed: bit 0 face A visited, 1 face B, 2 the face right of b, 3 the face left of b, 4 the face left of a;
      // bits 8-9 flags of a's tip, 16-17 flags of b's tip
      uint32_t marks = 0x0000011Eu;
#define fA_before (marks & 1u)
#define fB_before ((marks >> 1) & 1u)
#define fR_before ((marks >> 2) & 1u)
#define fL_before ((marks >> 3) & 1u)
#define fLA_before ((marks >> 4) & 1u)
#define flA ((marks >> 8) & 3u)
#define flB ((marks >> 16) & 3u)
#define TR_MARKS(fa_, fb_, fr_, fl_, fla_, ta_, tb_) (((fa_) ? 1u : 0u) | ((fb_) ? 2u : 0u) | ((fr_) ? 4u : 0u) | ((fl_) ? 8u : 0u) | ((fla_) ? 16u : 0u) | (((ta_) & 3u) << 8) | (((tb_) & 3u) << 16))
      bool pair_ok = false;
      uint32_t len = 0;

      if (have_prog && backoff == 0) { kind = 1; lin = prog_lin; }
      else {
        // ---------------------------------------------------------------- this element: record, marks; the record behind its right edge
        Raw rb0 = R::none();
        bool have_seed = false;
        if (have_rec) { v = c_v; rc = c_rc; lc = c_lc; }
        else {
          const Raw r0 = R::load(frec, face);
          const uint32_t kc0 = corner & 3u;
          v = uni(R::vertex(r0, kc0)); rc = uni(R::opp(r0, k_next(kc0))); lc = uni(R::opp(r0, k_prev(kc0)));
        }
        if (v >= NV || (rc != DSA_INVALID && !corner_ok(rc)) || (lc != DSA_INVALID && !corner_ok(lc))) TR_FAIL(301);
        if (have_state) bits = c_bits;
        else {
          // tip flag and the state of both sides, and the record a run from here would need next, issued together
          const uint32_t tip_flag = vflag[v];
          const uint32_t side_r = rc != DSA_INVALID ? (uint32_t)fvis[rc >> 2] : 1u, side_l = lc != DSA_INVALID ? (uint32_t)fvis[lc >> 2] : 1u;
          // (the records behind both edges ride along: whichever way the step goes, the next element's record is here)
          Raw rl0 = R::none();
          if (rc != DSA_INVALID) { rb0 = R::load(frec, rc >> 2); have_seed = true; }
          if (lc != DSA_INVALID) rl0 = R::load(frec, lc >> 2);
          have_kids = true;
          { const uint32_t kr = rc & 3u, kl = lc & 3u;
            kr_v = uni(R::vertex(rb0, kr)); kr_rc = uni(R::opp(rb0, k_next(kr))); kr_lc = uni(R::opp(rb0, k_prev(kr)));
            kl_v = uni(R::vertex(rl0, kl)); kl_rc = uni(R::opp(rl0, k_next(kl))); kl_lc = uni(R::opp(rl0, k_prev(kl))); }
          const uint32_t uf = uni(tip_flag);
          const bool rdone = rc == DSA_INVALID || (rc >> 2) == face || uni(side_r) != 0;
          const bool ldone = lc == DSA_INVALID || (lc >> 2) == face || uni(side_l) != 0;
          bits = ((uf & 1u) ? 0u : 1u) | (uf & 2u) | (rdone ? 4u : 0u) | (ldone ? 8u : 0u);
          TCOUNT(np_head);
        }
        have_rec = false; have_state = false;
        // this element moves right: tip new & interior, or only the right side is open
        const bool moves_right = (bits & 3u) == 1u || ((bits & 12u) == 8u);
        if (moves_right && backoff == 0 && !must_scalar && rc != DSA_INVALID) {
          if (!have_seed) rb0 = R::load(frec, rc >> 2);
          // the first pair exactly: a_0 = corner, b_0 = rc, and from b_0's record its tip, the corner right of it and a_1
          const uint32_t kb0 = rc & 3u;
          const uint32_t tB0 = uni(R::vertex(rb0, kb0)), rB0 = uni(R::opp(rb0, k_next(kb0))), a1 = uni(R::opp(rb0, k_prev(kb0)));
          const uint32_t d1 = a1 - corner;
          // steps this direction had the last time (keyed by the step of a)
          const uint32_t e = dir & 3u;
          const bool hm = !no_hist && d1 != 0u && corner_ok(a1) && uni(sh_hist[8 * e]) == d1;
          if (hm) {
            const uint32_t s_b = uni(sh_hist[8 * e + 1]), s_ta = uni(sh_hist[8 * e + 2]), s_tb = uni(sh_hist[8 * e + 3]), s_la = uni(sh_hist[8 * e + 4]), s_rb = uni(sh_hist[8 * e + 5]);
            p_a = corner + lane * d1; p_an = p_a + d1; p_b = rc + lane * s_b; p_ta = v + lane * s_ta; p_tb = tB0 + lane * s_tb;
            p_la = lc == DSA_INVALID ? DSA_INVALID : lc + lane * s_la; p_rb = rB0 == DSA_INVALID ? DSA_INVALID : rB0 + lane * s_rb;
            kind = 2; lin = true;
            TCOUNT(np_hist);
          } else {
            // ------------------------------------------------------------ dependent attempt: exact hops seed the candidate path
            // a_0, a_1, a_2 = succ(a_1) with succ(a) = Opposite(Previous(Opposite(Next(a)))); lanes 3.. extrapolate with constant
            // second difference; every link is verified from the records the lanes load anyway
            kind = 3;
            auto opp_prev_of = [&](uint32_t b1) -> uint32_t { return corner_ok(b1) ? uni(R::get_o_plain(frec, qprev(b1))) : DSA_INVALID; };
            auto opp_next_of = [&](uint32_t c) -> uint32_t { return corner_ok(c) ? uni(R::get_o_plain(frec, qnext(c))) : DSA_INVALID; };
            const uint32_t a0 = corner;
            const uint32_t a2 = opp_prev_of(opp_next_of(a1));
            const uint32_t d2 = a2 - a1, ddh = d2 - d1;
            const uint32_t exact = corner_ok(a2) ? 3u : (corner_ok(a1) ? 2u : 1u);
            // (the triangular number of the lane is made here from an opaque copy: as a loop invariant the compiler kept it in a
            // register it then had to spill -- a scratch reload and a full vmcnt wait in this path)
            uint32_t lq = lane;
            asm volatile("" : "+v"(lq));
            a = lane == 0 ? a0 : lane == 1 ? a1 : lane == 2 ? a2 : a2 + (lane - 2u) * (d2 + ddh) + ddh * ((lq - 2u) * (lq - 3u) / 2u);
            bool a_ok = lane < window && corner_ok(a) && (lane < exact || exact == 3);
            Raw ra = R::none(), rb = R::none();
            if (a_ok) ra = R::load(frec, a >> 2);
            const uint32_t ka = a & 3u;
            tipA = R::vertex(ra, ka); b = R::opp(ra, k_next(ka)); lcA = R::opp(ra, k_prev(ka));
            const bool b_ok = a_ok && corner_ok(b) && tipA < NV && (lcA == DSA_INVALID || corner_ok(lcA));
            if (b_ok) rb = R::load(frec, b >> 2);
            const uint32_t kb = b & 3u;
            tipB = R::vertex(rb, kb); rcB = R::opp(rb, k_next(kb)); lcB = R::opp(rb, k_prev(kb));
            const uint32_t next_a = b_ok ? lcB : DSA_INVALID;           // succ(a)
            {
              const uint32_t prev_next = lane_prev(next_a);
              a_ok = a_ok && (lane == 0 || prev_next == a);
            }
            len = leading_lanes(a_ok);                 // verified chain a_0 .. a_(len-1)
            pair_ok = lane < len && b_ok && tipB < NV && corner_ok(lcB) && (rcB == DSA_INVALID || corner_ok(rcB));
            if (pair_ok) {                             // state before the step
              const uint32_t m0 = lane == 0 ? 0u : fvis[a >> 2], m4 = lcA != DSA_INVALID ? fvis[lcA >> 2] : 1u, m1 = fvis[b >> 2];
              const uint32_t m2 = rcB != DSA_INVALID ? fvis[rcB >> 2] : 1u, m3 = fvis[lcB >> 2], ta = vflag[tipA], tb = vflag[tipB];
              marks = TR_MARKS(m0, m1, m2, m3, m4, ta, tb);
            }
            TCOUNT(np_dep);
          }
        } else if (backoff) --backoff;
        must_scalar = false;
        TPROF(1);
      }

      if (kind == 1 || kind == 2) {
        // ---------------------------------------------------------------- fast attempt: one round trip at the extrapolated ids
        ++n_fast;
        a = p_a;
        const bool a_ok = lane < window && corner_ok(a);
        const bool bp_ok = a_ok && corner_ok(p_b);
        Raw ra = R::none(), rb = R::none();
        if (a_ok) ra = R::load(frec, a >> 2);
        if (bp_ok) rb = R::load(frec, p_b >> 2);
        if (bp_ok) {
          const uint32_t m0 = lane == 0 ? 0u : fvis[a >> 2], m1 = fvis[p_b >> 2], m4 = corner_ok(p_la) ? fvis[p_la >> 2] : 1u;
          const uint32_t m2 = corner_ok(p_rb) ? fvis[p_rb >> 2] : 1u, m3 = corner_ok(p_an) ? fvis[p_an >> 2] : 1u;
          const uint32_t ta = p_ta < NV ? vflag[p_ta] : 1u, tb = p_tb < NV ? vflag[p_tb] : 1u;
          marks = TR_MARKS(m0, m1, m2, m3, m4, ta, tb);
        }
        const uint32_t ka = a & 3u, kb = p_b & 3u;
        tipA = R::vertex(ra, ka); b = R::opp(ra, k_next(ka)); lcA = R::opp(ra, k_prev(ka));
        tipB = R::vertex(rb, kb); rcB = R::opp(rb, k_next(kb)); lcB = R::opp(rb, k_prev(kb));
        // every extrapolated id against the records; then the ranges the dependent attempt checks
        const bool match = bp_ok && b == p_b && tipA == p_ta && lcA == p_la && tipB == p_tb && rcB == p_rb && lcB == p_an &&
                           tipA < NV && tipB < NV && corner_ok(lcB) && (lcA == DSA_INVALID || corner_ok(lcA)) && (rcB == DSA_INVALID || corner_ok(rcB));
        len = leading_lanes(match);
        pair_ok = lane < len;
        if (kind == 1) {       // the scalar step, should it come to that, starts from lane 0's record
          v = rdlane(tipA, 0); rc = rdlane(b, 0); lc = rdlane(lcA, 0);
          if (v >= NV || (rc != DSA_INVALID && !corner_ok(rc)) || (lc != DSA_INVALID && !corner_ok(lc))) TR_FAIL(301);
        }
        TPROF(0);
      } else if (kind == 3) { TPROF(2); }

      if (kind) {
        const uint32_t fa = a >> 2, fb = b >> 2;
        const uint32_t keyN = 2 * lane, keyL = 2 * lane + 1;
        // First position (N element of pair j: 2j, L element: 2j + 1) of a face / a tip among the candidates, TR_NONE if it is none
        // of them.  A tip that was visited before the run can never count as new, so only unvisited tips are looked up; the
        // neighbour faces only where their state before the run leaves the question open.
        uint32_t sfa = TR_NONE, sfb = TR_NONE, sta = TR_NONE, stb = TR_NONE, srf = TR_NONE, slf = TR_NONE, sla = TR_NONE;
        // Are the four id sequences (faces of a, faces of b, tips of a, tips of b) closed forms over the verified lanes?  A fast
        // attempt checked every lane against its progression; a dependent one built a by formula, the others are checked here against
        // the fit through lanes 0 .. 2.  Strictly monotonic sequences with a constant second difference: "which pair holds X" is a
        // division (linear) or a six-step bisection on the closed form -- no LDS tables.
        int32_t fa0, fas, fad, fb0, fbs, fbd, ta0, tas, tad, tb0, tbs, tbd;
        bool arith;
        {
          const uint32_t tri = lane * (lane - 1u) / 2u;
          const bool two = len >= 2, three = len >= 3;
          auto fit = [&](uint32_t x, int32_t &q0, int32_t &qs, int32_t &qd) {
            const uint32_t x0 = rdlane(x, 0), x1 = rdlane(x, 1), x2 = rdlane(x, 2);
            q0 = (int32_t)x0; qs = two ? (int32_t)(x1 - x0) : 1; qd = three ? (int32_t)((x2 - x1) - (x1 - x0)) : 0;
          };
          auto mono = [&](int32_t s_, int32_t d_) -> bool {
            const int32_t last = s_ + d_ * (int32_t)(two ? len - 2u : 0u);
            return s_ != 0 && last != 0 && ((s_ ^ last) >= 0) && d_ > -(1 << 24) && d_ < (1 << 24);
          };
          int32_t a0, as_, ad, b0, bs_, bd;
          fit(a, a0, as_, ad); fit(b, b0, bs_, bd); fit(tipA, ta0, tas, tad); fit(tipB, tb0, tbs, tbd);
          arith = ((as_ | ad | bs_ | bd) & 3) == 0 && mono(as_, ad) && mono(bs_, bd) && mono(tas, tad) && mono(tbs, tbd);
          if (arith && kind == 3)
            arith = __ballot(lane < len && (a != (uint32_t)a0 + lane * (uint32_t)as_ + (uint32_t)ad * tri || b != (uint32_t)b0 + lane * (uint32_t)bs_ + (uint32_t)bd * tri ||
                                            tipA != (uint32_t)ta0 + lane * (uint32_t)tas + (uint32_t)tad * tri || tipB != (uint32_t)tb0 + lane * (uint32_t)tbs + (uint32_t)tbd * tri)) == 0;
          fa0 = a0 >> 2; fas = as_ >> 2; fad = ad >> 2; fb0 = b0 >> 2; fbs = bs_ >> 2; fbd = bd >> 2;
        }
        if (arith) {
          TCOUNT(np_lin);
          const float iA = 1.0f / (float)fas, iB = 1.0f / (float)fbs, iTA = 1.0f / (float)tas, iTB = 1.0f / (float)tbs;
          auto pos_in = [&](uint32_t X, int32_t first, int32_t step, int32_t dd, float inv, uint32_t odd) -> uint32_t {
            if (dd == 0) {                                    // X = first + j step
              const int32_t d = (int32_t)X - first;
              const int32_t j = (int32_t)__builtin_rintf((float)d * inv);
              return (j >= 0 && (uint32_t)j < len && j * step == d) ? 2u * (uint32_t)j + odd : TR_NONE;
            }
            int32_t j = 0;                                    // the last j whose element is not beyond X
#pragma unroll
            for (int32_t bit = 32; bit >= 1; bit >>= 1) {
              const int32_t t = j + bit;
              const int32_t qt = first + t * step + dd * (t * (t - 1) / 2);
              if ((uint32_t)t < len && (step > 0 ? qt <= (int32_t)X : qt >= (int32_t)X)) j = t;
            }
            return first + j * step + dd * (j * (j - 1) / 2) == (int32_t)X ? 2u * (uint32_t)j + odd : TR_NONE;
          };
          auto face_pos = [&](uint32_t X) -> uint32_t { const uint32_t x = pos_in(X, fa0, fas, fad, iA, 0u), y = pos_in(X, fb0, fbs, fbd, iB, 1u); return x < y ? x : y; };
          auto tip_pos = [&](uint32_t X) -> uint32_t { const uint32_t x = pos_in(X, ta0, tas, tad, iTA, 0u), y = pos_in(X, tb0, tbs, tbd, iTB, 1u); return x < y ? x : y; };
          if (pair_ok) {
            sfa = face_pos(fa); sfb = face_pos(fb);
            if (!(flA & 1u)) sta = tip_pos(tipA);
            if (!(flB & 1u)) stb = tip_pos(tipB);
            if (rcB != DSA_INVALID && fR_before == 0) srf = face_pos(rcB >> 2);
            if (fL_before == 0) slf = face_pos(lcB >> 2);
            if (lcA != DSA_INVALID && fLA_before == 0 && flA != 0) sla = face_pos(lcA >> 2);
          }
        } else {
          // two small open-addressing tables in LDS, slot = run tag (24) | id (32) | position (8); a slot of an older run counts
          // as empty, so nothing is cleared between runs.  Exact: the smallest position per id wins (ds_min_u64).
          if (++run_id >= 0x00FFFFF0u) {        // 24-bit run tags: start over with empty tables (meshes with > 16 M runs)
            __syncthreads();
            for (uint32_t i = lane; i < TR_SLOTS; i += WAVE) { sh_tf[i] = 0; sh_tv[i] = 0; }
            __syncthreads();
            run_id = 1;
          }
          const uint32_t run_tag = 0x00FFFFFFu - run_id;
          auto tbl_insert = [&](unsigned long long *t, uint32_t id, uint32_t pos) {
            const unsigned long long want = ((unsigned long long)run_tag << 40) | ((unsigned long long)id << 8) | pos;
            uint32_t sl = (id * 2654435761u) >> (32 - TR_SLOT_BITS);
            for (;;) {
              const unsigned long long cur = __hip_atomic_load(&t[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if ((uint32_t)(cur >> 40) != run_tag) { if (atomicCAS(&t[sl], cur, want) == cur) break; continue; }
              if ((uint32_t)(cur >> 8) == id) { atomicMin(&t[sl], want); break; }
              sl = (sl + 1) & (TR_SLOTS - 1);
            }
          };
          auto tbl_lookup = [&](unsigned long long *t, uint32_t id) -> uint32_t {
            uint32_t sl = (id * 2654435761u) >> (32 - TR_SLOT_BITS);
            for (uint32_t probes = 0; probes < TR_SLOTS; ++probes) {
              const unsigned long long cur = __hip_atomic_load(&t[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if ((uint32_t)(cur >> 40) != run_tag) return TR_NONE;
              if ((uint32_t)(cur >> 8) == id) return (uint32_t)(cur & 0xFFu);
              sl = (sl + 1) & (TR_SLOTS - 1);
            }
            return TR_NONE;
          };
          if (pair_ok) {
            tbl_insert(sh_tf, fa, 2 * lane); tbl_insert(sh_tf, fb, 2 * lane + 1);
            if (!(flA & 1u)) tbl_insert(sh_tv, tipA, 2 * lane);
            if (!(flB & 1u)) tbl_insert(sh_tv, tipB, 2 * lane + 1);
          }
          __syncthreads();
          if (pair_ok) {
            sfa = tbl_lookup(sh_tf, fa); sfb = tbl_lookup(sh_tf, fb);
            if (!(flA & 1u)) sta = tbl_lookup(sh_tv, tipA);
            if (!(flB & 1u)) stb = tbl_lookup(sh_tv, tipB);
            if (rcB != DSA_INVALID && fR_before == 0) srf = tbl_lookup(sh_tf, rcB >> 2);
            if (fL_before == 0) slf = tbl_lookup(sh_tf, lcB >> 2);
            if (lcA != DSA_INVALID && fLA_before == 0 && flA != 0) sla = tbl_lookup(sh_tf, lcA >> 2);
          }
        }
        bool good = false, newA = false, newB = false;
        uint32_t hand = 0;                  // what the scalar step needs of this pair's two elements, should the run end on it
        if (pair_ok) {
          // first element (at a, face A): the face is first seen here and the DFS moves right -- because the
          // tip is new and interior (DepthFirstTraverser.cs:53-64), or because the left side is done
          // (:66-87; that the right side is open is the second element's "face B first seen")
          newA = !(flA & 1u) && sta == keyN;
          const bool la_done = lcA == DSA_INVALID || (lcA >> 2) == fa || fLA_before != 0 || sla < keyN;
          const bool x_ok = fA_before == 0 && sfa == keyN && ((newA && !(flA & 2u)) || la_done);
          // second element (at b, face B): face first seen here, the tip does not send the DFS right (new and
          // interior), right side done, left side open -> left
          newB = !(flB & 1u) && stb == keyL;
          const bool r_done = rcB == DSA_INVALID || (rcB >> 2) == fb || fR_before != 0 || srf < keyL;
          const bool l_open = (lcB >> 2) != fb && fL_before == 0 && !(slf < keyL);
          const bool y_ok = fB_before == 0 && sfb == keyL && !(newB && !(flB & 2u)) && r_done && l_open;
          good = x_ok && y_ok;
          // the element at a as the scalar step sees it once the pairs before this one are retired, and the element at b once the
          // step has gone right from a (its face and tip are then visited: positions keyN < keyL)
          const bool ra_done = fb == fa || fB_before != 0 || sfb < keyN;
          hand = 1u | (newA ? 2u : 0u) | ((flA & 2u) ? 4u : 0u) | (ra_done ? 8u : 0u) | (la_done ? 16u : 0u) |
                 (newB ? 32u : 0u) | ((flB & 2u) ? 64u : 0u) | (r_done ? 128u : 0u) | (l_open ? 0u : 256u) |
                 ((fA_before == 0 && sfa == keyN) ? 512u : 0u) | ((fB_before == 0 && sfb == keyL) ? 1024u : 0u);
        }
        const uint32_t K = leading_lanes(good);
#ifdef DSA_TRAV_TRACE
        // one word per attempt in the vertex-stamp scratch: kind | K << 4 | len << 12 | window << 20 (+ per-lane detail of attempts 2000 .. 2007)
        { uint32_t *tr = (uint32_t *)(arena + L.vstamp); const uint32_t at = n_run + n_fail;
          if (at < 8000 && lane == 0) { tr[0] = at + 1; tr[1 + at] = kind | (K << 4) | (len << 12) | (window << 20) | ((lin ? 1u : 0u) << 28); }
          if (at >= 1000 && at < 1008 && L.cap_vertices > 20000) { uint32_t *dt = tr + 8192 + (at - 1000) * 64 * 16 + lane * 16; dt[0] = a; dt[1] = b; dt[2] = tipA; dt[3] = tipB; dt[4] = lcA; dt[5] = rcB; dt[6] = lcB; dt[7] = marks | (good ? 1u << 31 : 0u) | (pair_ok ? 1u << 30 : 0u) | (hand << 20);
            dt[8] = p_a; dt[9] = p_b; dt[10] = p_ta; dt[11] = p_tb; dt[12] = p_la; dt[13] = p_rb; dt[14] = p_an; dt[15] = kind; } }
#endif
        TPROF(3);
        // entries made by the retired pairs: a new tip is numbered when its element is reached (:53-58)
        const uint64_t kmask = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
        const uint64_t mA = __ballot(newA) & kmask, mB = __ballot(newB) & kmask;
        const uint32_t made = (uint32_t)__popcll(mA) + (uint32_t)__popcll(mB);
        const bool retire = K >= 1 && count + made <= cap_entries;
        const uint32_t win_used = window;          // lanes that took part in this attempt
        have_prog = false;
        if (kind == 2) {
          // steps that let the first pair through and fail the second, twice in a row, are stale (a side along the boundary has
          // no face right of b, the next ring has): forget them, the exact hops of the dependent attempt learn them again
          if (K == len && len <= 2 && len < window) { if (++hist_strikes >= 2) { if (lane == 0) sh_hist[8 * (dir & 3u)] = 0; hist_strikes = 0; } }
          else if (K >= 3) hist_strikes = 0;
        }
        if (retire) {
          if (lane < K) {
            const uint64_t lt = (1ull << lane) - 1ull;
            const uint32_t posA = count + (uint32_t)__popcll(mA & lt) + (uint32_t)__popcll(mB & lt);
            fvis[fa] = 1; fvis[fb] = 1;
            if (newA) { vflag[tipA] = (uint8_t)(flA | 1u); d2c[posA] = a; v2d[tipA] = (int32_t)posA; }
            if (newB) { const uint32_t posB = posA + (newA ? 1u : 0u); vflag[tipB] = (uint8_t)(flB | 1u); d2c[posB] = b; v2d[tipB] = (int32_t)posB; }
          }
          const uint32_t nxt = rdlane(lcB, K - 1);             // Opposite(Previous(b_(K-1))): where the DFS continues
          count += made;
          if (K >= 3) {
            // the verified progressions: continued by the next attempt when the run filled its window, remembered as the steps
            // of this direction when they are linear
            const uint32_t aK1 = rdlane(a, K - 1), aK2 = rdlane(a, K - 2), aK3 = rdlane(a, K - 3);
            const uint32_t s1 = nxt - aK1, s0 = aK1 - aK2, sm = aK2 - aK3;
            const uint32_t a_d = (s1 - s0 == s0 - sm) ? s1 - s0 : 0u, a_s = s1 + a_d;
            uint32_t any_d = a_d | (s1 ^ s0);
            const bool cont = K == window;           // the run filled its window: nothing stopped it
            const uint32_t l1 = lane + 1u, tri1 = l1 * (l1 + 1u) / 2u;
            if (cont) {
              uint32_t lq = lane;                    // (see the dependent attempt: not a loop invariant to keep and spill)
              asm volatile("" : "+v"(lq));
              p_a = nxt + lane * a_s + a_d * (lq * (lq - 1u) / 2u); p_an = nxt + l1 * a_s + a_d * ((lq + 1u) * lq / 2u);
            }
            uint32_t st_b, st_ta, st_tb, st_la, st_rb;
#define TR_PROG(q_, p_, st_) { const uint32_t x1 = rdlane(q_, K - 1), x2 = rdlane(q_, K - 2), x3 = rdlane(q_, K - 3); \
                               const uint32_t t0 = x1 - x2, qd = t0 - (x2 - x3); any_d |= qd; st_ = t0; if (cont) p_ = x1 + l1 * t0 + qd * tri1; }
            TR_PROG(b, p_b, st_b); TR_PROG(tipA, p_ta, st_ta); TR_PROG(tipB, p_tb, st_tb); TR_PROG(lcA, p_la, st_la); TR_PROG(rcB, p_rb, st_rb);
#undef TR_PROG
            // linear, faces and tips all distinct along the run: membership by arithmetic
            const bool linear = any_d == 0 && (s0 & 3u) == 0 && (st_b & 3u) == 0 && s0 != 0 && st_b != 0 && st_ta != 0 && st_tb != 0;
            have_prog = cont; prog_lin = linear;
            if (linear && kind == 3) {       // remember the steps of this direction (a fast run only confirms what it was given)
              const uint32_t e = dir & 3u;
              if (lane == 0) { sh_hist[8 * e] = s0; sh_hist[8 * e + 1] = st_b; sh_hist[8 * e + 2] = st_ta; sh_hist[8 * e + 3] = st_tb; sh_hist[8 * e + 4] = st_la; sh_hist[8 * e + 5] = st_rb; }
            }
          }
          corner = nxt;
          face = corner >> 2;
          have_kids = false;                 // (they were the neighbours of the corner the run started from)
          n_run += 1; n_run_faces += 2 * K;
          fail_streak = 0; no_hist = false;
          if ((fuse_operands & 2u) && K < window && len > K) { side2 = side1; side1 = K; const uint32_t m = (side1 > side2 ? side1 : side2) + 4; window = m < WAVE ? m : WAVE; }
          else window = WAVE;
          if (kind != 3) TCOUNT(np_fast_hit);
          TPROF(4);
          if (K >= win_used) continue;
          if (K < len) ++dir;                    // stopped by a verdict, not by a wrong guess: a turn
          // Pair K was loaded too.  If it lies on the verified path it is not an (N L) pair (a turn of the spiral, a boundary vertex,
          // a split): re-attempting from it would reach the same verdict, so its first element takes the scalar step -- from what
          // lane K holds.  If the path ended on it (an extrapolated id was wrong), its record at least is the right one.
        } else {
          ++n_fail;
          if (kind == 3) { backoff = fail_streak < 3 ? fail_streak : 3; ++fail_streak; }
          if (kind == 2) no_hist = true;
        }
        // ---- the pair the DFS stands on now (K if pairs were retired, else 0) was loaded by its lane: hand its elements over
        {
          const uint32_t h_lane = retire ? K : 0u;
          const uint32_t h = (h_lane < win_used && h_lane < len) ? rdlane(hand, h_lane) : 0u;
          if (h & 1u) {
            v = rdlane(tipA, h_lane); rc = rdlane(b, h_lane); lc = rdlane(lcA, h_lane);
            // (fA_before | first occurrence) failing means the face was visited meanwhile: the stack logic below handles a visited face
            // only at a pop, so such a pair goes back to the loads
            if (h & 512u) {
              bits = ((h & 2u) ? 1u : 0u) | ((h & 4u) ? 2u : 0u) | ((h & 8u) ? 4u : 0u) | ((h & 16u) ? 8u : 0u);
              have2 = (h & 1024u) != 0;
              c2_v = rdlane(tipB, h_lane); c2_rc = rdlane(rcB, h_lane); c2_lc = rdlane(lcB, h_lane);
              c2_bits = ((h & 32u) ? 1u : 0u) | ((h & 64u) ? 2u : 0u) | ((h & 128u) ? 4u : 0u) | ((h & 256u) ? 8u : 0u);
              TCOUNT(np_hand);
            } else { have_rec = true; c_v = v; c_rc = rc; c_lc = lc; must_scalar = true; continue; }
          } else if (kind != 3 || retire) {
            // no marks for this element (its ids were guessed wrong, or the chain ended here): its record if lane h_lane loaded the
            // right face (a_(h_lane) is exact: lane 0 is the corner itself, lane K is where lane K - 1's record points), then the loads
            const bool rec_ok = h_lane < WAVE && h_lane < win_used && (kind != 3 || h_lane < len);
            if (rec_ok && rdlane(a, h_lane) == corner) { have_rec = true; c_v = rdlane(tipA, h_lane); c_rc = rdlane(b, h_lane); c_lc = rdlane(lcA, h_lane); }
            must_scalar = retire && h_lane < len;
            continue;
          }
          // (a dependent attempt that retired nothing: the element's inputs are the ones loaded above)
        }
      }
      // ------------------------------------------------------------------ scalar step (reference loop body)
      ++n_scalar;
      no_hist = false;
      if (lane == 0) fvis[face] = 1;
      bool went_right = false;
      if (bits & 1u) {
        if (count >= cap_entries) TR_FAIL(302);
        VISIT_SCALAR(v, corner, bits & 2u);
        if (!(bits & 2u)) {
          if (rc == DSA_INVALID) TR_FAIL(303);
          corner = rc;
          went_right = true;
        }
      }
      bool went_left = false;
      if (!went_right) {
        if (bits & 4u) {
          if (bits & 8u) { --sp; TPROF(5); break; }
          corner = lc;
          went_left = true;
        } else {
          if (bits & 8u) { corner = rc; went_right = true; }
          else {
            if (sp >= stack_cap) TR_FAIL(304);
            if (lane == 0) { stack[sp - 1] = lc; stack[sp] = rc; }
            ++sp;
            TPROF(5);
            break;
          }
        }
      }
      // the element behind the right edge is known too when the attempt judged it
      if (went_right && have2) { have_rec = true; have_state = true; c_v = c2_v; c_rc = c2_rc; c_lc = c2_lc; c_bits = c2_bits; }
      else if (have_kids && went_right) { have_rec = true; c_v = kr_v; c_rc = kr_rc; c_lc = kr_lc; }
      else if (have_kids && went_left) { have_rec = true; c_v = kl_v; c_rc = kl_rc; c_lc = kl_lc; }
      have2 = false;
      TPROF(5);
    }
    if (failed) break;
  }
  if (failed) return;
  if (lane == 0 && !position) { if (count != expect) fail(D, ST_INVALID, 305); }
  if (lane == 0 && position) {
    D->num_entries = count;
    D->dbg[5] = n_fail; D->dbg[6] = (uint32_t)(clk() - t_start);
    D->dbg[16] = (uint32_t)r_start; D->dbg[17] = (uint32_t)(realclk() - r_start);
    D->dbg[7] = n_run; D->dbg[8] = n_run_faces; D->dbg[9] = n_scalar; D->dbg[3] = n_fast;
#ifdef DSA_TRAV_PROFILE
    for (int i = 0; i < 5; ++i) D->dbg[10 + (i < 3 ? i : i + 5)] = (uint32_t)(tp_acc[i] >> 4);   // [10] [11] [12] [18] [19], in units of 16 clocks
    D->dbg[0] = (uint32_t)(tp_acc[5] >> 4); D->dbg[1] = np_fast_hit; D->dbg[2] = np_dep; D->dbg[4] = np_head; D->dbg[13] = np_hist; D->dbg[14] = np_lin; D->dbg[15] = np_hand;
#endif
    // a valid stream carries exactly one entry per encoded vertex (k_locate sized the symbol streams on that)
    if (count != expect) fail(D, ST_INVALID, 305);
  }
  // Large batches: the wave that just produced the order also derives the parallelogram operands of its mesh, while
  // other meshes are still being traversed (small batches use the element-parallel k_para_operands instead).
  // Both passes below are chains of dependent gathers with independent iterations: written without branches (clamped
  // indices, selects) and four iterations at a time, so that four chains are in flight per lane.
  if ((fuse_operands & 1u) && count == expect) {
    WAIT_VM0();
    __threadfence_block();
    uint32_t *para = io.para;
    for (uint32_t p0 = 0; p0 < count; p0 += 4 * WAVE) {
      uint32_t en[4], ep[4], eo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t p = p0 + u * WAVE + lane;
        para_operands_flat<CP>(p < count ? p : 0u, frec, d2c, v2d, F, NV, en[u], ep[u], eo[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t p = p0 + u * WAVE + lane;
        if (p < count) { Triple t; t.a = en[u]; t.b = ep[u]; t.c = eo[u]; *(Triple *)(para + 3 * (size_t)p) = t; }
      }
    }
  }
  // ---- point -> entry map of every attribute (MeshTraversalSequencer.cs:33-50), from the order just produced
  // (a mesh with corner attributes gets every map from k_seam_maps: its points are not its vertices)
  if (position && !D->seam_fast && count == expect && NV) {
    WAIT_VM0();
    __threadfence_block();
    const uint32_t *vrank = (const uint32_t *)(arena + L.vrank);
    const uint32_t na = uni(D->num_attributes), npts = uni(D->num_points);
    for (uint32_t v0 = 0; v0 < NV; v0 += 4 * WAVE) {
      int32_t e[4];
      uint32_t point[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t v = v0 + u * WAVE + lane, vc = v < NV ? v : NV - 1;
        e[u] = v < NV ? v2d[vc] : -1;          // no corner: the map keeps its initial value
        point[u] = vrank[vc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e[u] >= 0 && point[u] < npts)
          for (uint32_t ai = 0; ai < na; ++ai) ((uint32_t *)(arena + L.map[ai]))[point[u]] = (uint32_t)e[u];
    }
  }
#undef TR_FAIL
#undef VISIT_SCALAR
#undef TPROF
#undef TCOUNT
#undef fA_before
#undef fB_before
#undef fR_before
#undef fL_before
#undef fLA_before
#undef flA
#undef flB
#undef TR_MARKS
}

__global__ __launch_bounds__(WAVE) void k_traverse(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t fuse_operands) {
  __shared__ unsigned long long sh_tf[TR_SLOTS], sh_tv[TR_SLOTS];
  __shared__ uint32_t sh_hist[TR_HIST_WORDS];
  for (uint32_t i = threadIdx.x; i < TR_SLOTS; i += WAVE) { sh_tf[i] = 0; sh_tv[i] = 0; }
  if (threadIdx.x < TR_HIST_WORDS) sh_hist[threadIdx.x] = 0;
  __syncthreads();
  __builtin_amdgcn_s_setprio(DSA_CHAIN_PRIO);   // critical path: issue ahead of the entropy-decode waves sharing the CU
  const uint32_t mesh = blockIdx.x;
  if (mesh >= n) return;
  if (status_of(&descs[mesh]) != ST_OK) return;
  const TravIO io = trav_position(arena, layouts[mesh], &descs[mesh]);
  if (layouts[mesh].rec_compact) traverse_wave<true>(arena, layouts[mesh], &descs[mesh], io, fuse_operands, sh_tf, sh_tv, sh_hist);
  else traverse_wave<false>(arena, layouts[mesh], &descs[mesh], io, fuse_operands, sh_tf, sh_tv, sh_hist);
}

// k_chain: connectivity and traversal of a mesh by the same wave, back to back.  As two kernels the traversal's waves
// find their places taken: the connectivity waves of a large batch retire over 3 ms, every freed slot goes to a waiting
// entropy-decode wave (that kernel is mid-dispatch, k_traverse is not launched yet), and most traversal waves then start up
// to 11 ms late, when those decoders finish (tools/wave_times.py).  A wave that keeps its slot has no such gap.
// LDS (CN_LDS_WORDS * 4 bytes) is passed at launch: with the size hidden from the compiler, the launch bound alone sets the
// register budget (64 VGPRs: four of these waves and three entropy-decode waves of 80 share a SIMD's 512).
extern __shared__ __attribute__((aligned(16))) uint32_t sh_chain[];
__global__ __launch_bounds__(WAVE, 8) void k_chain(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t fuse_operands) {
  uint32_t *sh = sh_chain;
  static_assert(CN_LDS_WORDS * 4 >= 2 * TR_SLOTS * 8 + TR_HIST_WORDS * 4, "the traversal's tables reuse the connectivity's LDS");
  __builtin_amdgcn_s_setprio(DSA_CHAIN_PRIO);
  const uint32_t mesh = blockIdx.x;
  if (mesh >= n) return;
  const MeshLayout &L = layouts[mesh];
  MeshDesc *D = &descs[mesh];
  const bool compact = uni(L.rec_compact) != 0;
  uint32_t *sh_rec = sh + CN_STAGE * 8, *sh_win = sh_rec + CN_REC_BLOCKS * 64 * 2, *sh_ctx = sh_win + CN_WIN;
  if (!compact) connectivity_wave<false, false>(arena, L, D, sh, sh_rec, sh_win, sh_ctx);
  else if (uni((uint32_t)D->traversal_type) == 2u) connectivity_wave<true, true>(arena, L, D, sh, sh_rec, sh_win, sh_ctx);
  else connectivity_wave<true, false>(arena, L, D, sh, sh_rec, sh_win, sh_ctx);
  // the traversal reads what this wave (all lanes) just wrote: records, ranks, flags
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  __syncthreads();
  unsigned long long *sh_tf = (unsigned long long *)sh, *sh_tv = sh_tf + TR_SLOTS;
  uint32_t *sh_hist = sh + 2 * TR_SLOTS * 2;
  for (uint32_t i = threadIdx.x; i < TR_SLOTS; i += WAVE) { sh_tf[i] = 0; sh_tv[i] = 0; }
  if (threadIdx.x < TR_HIST_WORDS) sh_hist[threadIdx.x] = 0;
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  const TravIO io = trav_position(arena, L, D);
  if (compact) traverse_wave<true>(arena, L, D, io, fuse_operands, sh_tf, sh_tv, sh_hist);
  else traverse_wave<false>(arena, L, D, io, fuse_operands, sh_tf, sh_tv, sh_hist);
  if ((fuse_operands & 1u) && (fuse_operands & 8u) && status_of(D) == ST_OK) {      // bit 3: late_handoff (with the operands written by this wave)
    const uint32_t na = uni(D->num_attributes);
    for (uint32_t ai = 0; ai < na && ai < DSA_MAX_ATT; ++ai) late_handoff(arena, L, D, ai, PW_FLAG | LATE_HANDOFF, 2u);
  }
}


// =========================================================================
// k_symbols: one wave per (mesh, attribute) value stream -> int32 corrections.
// =========================================================================

// Wave-uniform rANS symbol decode (Entropy/RAnsDecoder.cs:56-99).  The state and the
// stream offset are wave-uniform; the cumulative-frequency table is spread over the
// lanes (first 64 boundaries in a register, the rest in LDS) and a symbol is found with
// one or two ballot+popcount steps instead of the reference's 2^precision-entry LUT.
// symtab != nullptr: the alphabet is sparse (more symbol ids than the LDS search holds, few of them used): the table is
// compacted to the symbols with a non-zero frequency, the chain decodes compact indices and a lane-parallel pass
// maps them back through symtab[] (global scratch).
__device__ __forceinline__ void rans_decode_wave(MeshDesc *D, const uint8_t *stream, uint32_t stream_len, const AttrDesc &a, uint32_t num_values,
                                 uint32_t *out, uint32_t *lds_cum, uint32_t *symtab) {
  const uint32_t lane = lane_id();
  const uint32_t nsym = symtab ? a.num_distinct : a.num_symbols;
  const uint32_t P = a.precision_bits, precision = 1u << P, l_base = precision * 4;
  // 1. probability table -> LDS (lane 0), then cumulative in place
  if (lane == 0) {
    Rd r(stream, stream_len, a.off_table);
    if (symtab) {                       // RAnsSymbolDecoder.cs:21-48, keeping the non-zero entries only
      uint32_t k = 0;
      bool ok = true;
      for (uint32_t i = 0; i < a.num_symbols && ok; ++i) {
        const uint32_t pd = r.u8(), token = pd & 3u;
        if (token == 3u) { const uint32_t offset = pd >> 2; if (i + offset >= a.num_symbols) ok = false; i += offset; }
        else {
          uint32_t pr = pd >> 2;
          for (uint32_t j = 0; j < token; ++j) pr |= r.u8() << (8 * (j + 1) - 2);
          if (pr) { if (k >= nsym) ok = false; else { lds_cum[k] = pr; symtab[k] = i; ++k; } }
        }
      }
      if (!ok || !r.ok || k != nsym) fail(D, ST_INVALID, 400);
    } else if (!read_prob_table(r, nsym, lds_cum)) fail(D, ST_INVALID, 400);
  }
  WAIT_VM0();
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nsym; b0 += WAVE) {
    uint32_t i = b0 + lane;
    uint32_t pr = i < nsym ? lds_cum[i] : 0u;
    uint32_t tot;
    uint32_t ex = wave_excl_scan(pr, &tot);
    if (i < nsym) lds_cum[i] = carry + ex;
    carry += tot;
  }
  if (carry != precision) { if (lane == 0) fail(D, ST_INVALID, 401); return; }
  const uint32_t nblocks = (nsym + WAVE - 1) / WAVE;
  for (uint32_t i = nsym + lane; i < nblocks * WAVE + 1; i += WAVE) lds_cum[i] = precision;   // padding compares false, last next == precision
  __syncthreads();
  // coarse boundaries: lane l holds cum[64*l]
  uint32_t coarse = (lane < nblocks) ? lds_cum[lane * WAVE] : precision;
  uint32_t fine0 = lds_cum[lane];   // block 0, used when the alphabet fits one block
  // 2. initial state from the stream tail
  const uint8_t *buf = stream + a.off_rans;
  uint32_t x, off;
  {
    uint32_t st = 0, o = 0;
    bool ok = rans_init(buf, a.size_rans, l_base, &st, &o);
    if (!ok) { if (lane == 0) fail(D, ST_INVALID, 402); return; }
    x = uni(st); off = uni(o);
  }
  // 3. byte window: lane l holds the dword at aligned byte (chunk*256 + 4l) of the arena-relative stream
  const uintptr_t base_addr = (uintptr_t)buf;
  const uint32_t mis = (uint32_t)(base_addr & 3u);            // buf = aligned + mis
  const uint32_t *abuf = (const uint32_t *)(base_addr - mis);
  uint32_t chunk = 0xFFFFFFFFu, W = 0;
  const uint32_t mask = precision - 1;
  uint32_t mine = 0;
  for (uint32_t i = 0; i < num_values; ++i) {
    while (x < l_base && off > 0) {
      --off;
      uint32_t q = off + mis;                                  // byte index from abuf
      uint32_t ch = q >> 8;
      if (ch != chunk) { chunk = ch; W = abuf[(size_t)ch * 64 + lane]; }   // arena padding makes the over-read safe
      uint32_t byte = (rdlane(W, (q & 255u) >> 2) >> ((q & 3u) * 8)) & 0xFFu;
      x = (x << 8) | byte;
    }
    uint32_t rem = x & mask;
    uint32_t s, cs, nx;
    if (nblocks == 1) {
      uint32_t j = (uint32_t)__popcll(__ballot(fine0 <= rem)) - 1u;
      cs = rdlane(fine0, j);
      nx = (j < 63) ? rdlane(fine0, j + 1) : precision;
      s = j;
    } else {
      uint32_t b = (uint32_t)__popcll(__ballot(coarse <= rem)) - 1u;
      uint32_t v = lds_cum[b * WAVE + lane];
      uint32_t j = (uint32_t)__popcll(__ballot(v <= rem)) - 1u;
      cs = rdlane(v, j);
      nx = (j < 63) ? rdlane(v, j + 1) : rdlane(coarse, b + 1);
      s = b * WAVE + j;
    }
    x = (nx - cs) * (x >> P) + rem - cs;
    if ((i & 63u) == lane) mine = s;
    if ((i & 63u) == 63u) out[i - 63u + lane] = mine;
  }
  uint32_t tail = num_values & 63u;
  if (tail && lane < tail) out[num_values - tail + lane] = mine;
  if (symtab) {                         // compact index -> symbol id
    WAIT_VM0();
    __syncthreads();
    for (uint32_t i = lane; i < num_values; i += WAVE) out[i] = symtab[out[i]];
  }
}

// Generic (large alphabet) fallback: cumulative table in global scratch, lane 0, binary search.
__device__ __forceinline__ void rans_decode_serial(MeshDesc *D, const uint8_t *stream, uint32_t stream_len, const AttrDesc &a, uint32_t num_values,
                                   uint32_t *out, uint32_t *cum /* nsym+1 entries of global scratch */) {
  if (threadIdx.x != 0) return;
  const uint32_t nsym = a.num_symbols, P = a.precision_bits, precision = 1u << P, l_base = precision * 4;
  Rd r(stream, stream_len, a.off_table);
  if (!read_prob_table(r, nsym, cum)) { fail(D, ST_INVALID, 410); return; }
  uint32_t c = 0;
  for (uint32_t i = 0; i < nsym; ++i) { uint32_t pr = cum[i]; cum[i] = c; c += pr; if (c > precision) break; }
  if (c != precision) { fail(D, ST_INVALID, 411); return; }
  cum[nsym] = precision;
  const uint8_t *buf = stream + a.off_rans;
  uint32_t x, off;
  if (!rans_init(buf, a.size_rans, l_base, &x, &off)) { fail(D, ST_INVALID, 412); return; }
  for (uint32_t i = 0; i < num_values; ++i) {
    while (x < l_base && off > 0) x = (x << 8) | buf[--off];
    uint32_t rem = x & (precision - 1);
    uint32_t lo = 0, hi = nsym;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= rem) lo = mid; else hi = mid; }
    // zero-probability symbols share cum with their successor: the search lands on the last index with cum <= rem,
    // which is the (unique) symbol whose range contains rem.
    uint32_t cs = cum[lo];
    x = (cum[lo + 1] - cs) * (x >> P) + rem - cs;
    out[i] = lo;
  }
}

// ---------------------------------------------------------------------------------------------
// Register-file rANS decode for 12-bit precision (what 8..11-bit quantised attributes produce).
// The serial state update  x' = freq[s]*(x >> 12) + (rem - cum[s])  (RAnsDecoder.cs:56-67) only needs, per
// slot rem, the pair {freq, rem - cum}: the reference's 4096-entry slot table (RAnsDecoder.cs:69-88) is kept
// as one packed word per slot in 64 VGPRs (slot r: register r>>6, lane r&63) and read with an M0-indexed
// register move + v_readlane, so the chain never touches LDS or memory.  The chain records the *slot* of
// every position; which symbol a slot belongs to is looked up afterwards by all 64 lanes in parallel
// (slot -> symbol table in the attribute's scratch), fused with the zig-zag step.
typedef uint32_t v32u __attribute__((ext_vector_type(32)));   // largest vector the backend indexes through M0
#define REG_MAX_SYMS 4096     // a 12-bit-precision table has 4096 slots: no more symbols than that can have a frequency
#define WIDE_MAX_SYMS 2048    // k_symbols_wide: 32 registers of 64 {cumulative, frequency} words

// SYM_EARLY_FUSE (option, off by default): the wave that decoded the symbols of an "early" attribute (its prediction needs no
// traversal data) goes on to predict and dequantise it -- what k_predict / k_predict_wrap / k_finalize of phase 0 do behind the
// whole symbol launch, which ends 28 ms into a 4096-mesh decode and leaves the octahedral prediction (11 ms) in the tail.  The
// wave keeps its 80-register slot for that, and measured, the late symbols then end 4.7 ms later for a tail 2.9 ms shorter.
#define SYM_EARLY_FUSE 0x400u
__device__ __forceinline__ void predict_wave(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t phase, uint32_t flags);
__device__ __forceinline__ void predict_wrap_attribute(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t phase, uint32_t flags);
__device__ __forceinline__ void finalize_attribute(uint8_t *arena, const MeshLayout &L, const MeshDesc *D, uint32_t ai, uint32_t phase, uint32_t flags, uint32_t tid, uint32_t stride);
__device__ __forceinline__ void early_tail(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t flags) {
  if (!(flags & SYM_EARLY_FUSE) || D->general) return;
  const AttrDesc &a = D->att[ai];
  if (att_is_late(a)) return;                                    // late: waits for the traversal
  WAIT_VM0();
  __threadfence_block();
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  predict_wrap_attribute(arena, L, D, ai, 0u, flags);
  predict_wave(arena, L, D, ai, 0u, flags);
  WAIT_VM0();
  __threadfence_block();
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  finalize_attribute(arena, L, D, ai, 0u, flags, lane_id(), WAVE);
  if (lane_id() == 0) D->att[ai].early_done = 1;
}

// Late attributes of a crowded batch (parallelogram on the position connectivity) are predicted by whichever of their two producers
// finishes second -- the wave that decoded the corrections or the wave that traversed the mesh -- instead of by a kernel behind
// both launches: the prediction of a mesh then starts when ITS inputs are there (the position streams are decoded ten milliseconds
// before the traversals end; the traversals end three before the last texture-coordinate streams), in the shadow of the other
// meshes' work, and the k_predict_wrap launch behind everything finds little left.  `bit`: 1 = corrections, 2 = order and operands.
// Release / acquire at agent scope around the flag: the two waves may sit on different XCDs (an L2 each).
__device__ __forceinline__ bool wrap_fast_ok(const AttrDesc &a, uint32_t flags);
__device__ __forceinline__ void late_handoff(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t flags, uint32_t bit) {
  if (!(flags & LATE_HANDOFF) || D->general) return;
  AttrDesc &a = D->att[ai];
  if (!wrap_fast_ok(a, flags) || !att_is_late(a) || att_behind_tables(a) || a.pred_kind != 1) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  uint32_t old = 0;
  if (lane_id() == 0) old = __hip_atomic_fetch_or(&a.late_ready, bit, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  old = uni(old);
  if (!(old & (3u ^ bit))) return;                   // the other wave will find this one's bit
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (status_of(D) != ST_OK) return;
  predict_wrap_attribute(arena, L, D, ai, 1u, flags);
  WAIT_VM0();
  __syncthreads();
  if (lane_id() == 0) a.late_ready = 7u;
}

// The decoder serves three kinds of 12-bit-precision streams (MODE):
//   REG_ATTR   the raw symbol stream of attribute ai
//   REG_TAGS   the tag stream of a tagged attribute (SymbolDecoding.cs:30-50: one 5-bit tag per entry) -- the attribute the walk of the
//              mesh stopped at (k_locate / k_locate_resume); the tags go to the attribute's output region as bytes, where k_symbols<0>
//              expects them, and their bit total to the descriptor, where the walk needs it to find what follows.  The tables then live
//              behind the slots in the work region (the output region holds the tags)
//   REG_VLIST  context list ai (0..5) of valence-coded connectivity (MeshEdgeBreakerTraversalValenceDecoder.cs:22-69), into the
//              face-output region, where the connectivity wave reads the lists; tables in the face-record region, which that
//              wave fills later.  The connectivity wave decodes a list itself when this kernel could not (val_lists_done)
#define REG_ATTR 0
#define REG_TAGS 1
#define REG_VLIST 2
#define REG_SCRATCH_BYTES (4096u * 6u + REG_MAX_SYMS * 4u)
template <int MODE>
__device__ __forceinline__ void reg_decode_stream(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t flags) {
  constexpr bool TAGS = MODE == REG_TAGS, VLIST = MODE == REG_VLIST;
  AttrDesc &a = D->att[VLIST ? 0u : ai];
  uint8_t *scratch;
  uint32_t st_off_table = a.off_table, st_off_rans = a.off_rans, st_size_rans = a.size_rans, st_nsym = a.num_symbols;
  uint32_t st_values = a.num_entries * (TAGS ? 1u : (uint32_t)a.nc_portable);
  uint32_t *out = (uint32_t *)(arena + L.work[VLIST ? 0u : ai]);
  if (VLIST) {
    st_values = D->val_count[ai];
    if (st_values == 0 || D->val_prec[ai] != 12 || D->val_nsym[ai] > 64) return;
    if ((uint64_t)L.rec_compact * 16ull * D->num_faces < 6ull * REG_SCRATCH_BYTES) return;     // (small meshes: the connectivity wave decodes its lists)
    st_off_table = D->val_off_table[ai]; st_off_rans = D->val_off_rans[ai]; st_size_rans = D->val_size_rans[ai]; st_nsym = D->val_nsym[ai];
    uint32_t before = 0;
    for (uint32_t k = 0; k < ai; ++k) before += D->val_count[k];
    if ((uint64_t)before + st_values > 3ull * D->num_faces) return;                                // (the lists of a sound stream hold one symbol per face)
    out = (uint32_t *)(arena + L.faces) + before;
    scratch = arena + L.frec + (size_t)ai * REG_SCRATCH_BYTES;
  } else if (TAGS) {
    if (a.source != SRC_TAGGED || a.tags_done || a.num_symbols > LOC_MAX_TAGS || a.num_distinct <= 1 || a.num_entries == 0) return;
    const uint64_t slots = ((uint64_t)a.num_entries * 4 + 15) & ~15ull;
    if ((uint64_t)L.work_cap[ai] * 4 < slots + REG_SCRATCH_BYTES) return;      // (small meshes: the walk decodes the tags itself)
    scratch = arena + L.work[ai] + slots;
  } else {
    if (lanes::ln_sym_eligible(a, L, ai, flags)) return;             // k_symbols_lanes
    if (sym_filtered(a, flags)) return;                              // the other launch of the early / late pair
    // one non-zero symbol = a frequency of 4096, which the packed {freq, rem - cum} word cannot hold: k_symbols<T> takes it
    if (a.source != SRC_RAW || a.precision_bits != 12 || a.num_symbols > REG_MAX_SYMS || a.num_distinct <= 1) return;
    if (L.out_cap[ai] < 4096 * 6 + REG_MAX_SYMS * 4) return;      // scratch for the tables (k_symbols<T> takes the stream instead)
    scratch = arena + L.out[ai];
  }
  const uint8_t *stream = arena + L.stream;
  uint32_t *slot_tab = (uint32_t *)scratch;                        // {freq | (rem - cum) << 16} per slot
  uint16_t *slot_sym = (uint16_t *)(scratch + 4096 * 4);           // symbol per slot
  // probabilities, then packed {freq << 12 | cum} per symbol: global scratch rather than LDS, which the
  // connectivity and traversal waves sharing the CU need for their caches and tables
  uint32_t *lds = (uint32_t *)(scratch + 4096 * 6);
  const uint32_t lane = lane_id();
  const uint32_t nsym = uni(st_nsym);
  const uint32_t num_values = uni(st_values);
  if (num_values == 0) return;
  // 1. probability table -> LDS (lane 0), RAnsSymbolDecoder.cs:21-48
  if (lane == 0) {
    Rd r(stream, L.stream_len, st_off_table);
    if (!read_prob_table(r, nsym, lds)) fail(D, ST_INVALID, 400);
  }
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  // 2. cumulative frequencies; a frequency of 4096 (one-symbol alphabet) does not fit the packing
  uint32_t carry = 0, one_at = DSA_INVALID;
  bool over = false;
  for (uint32_t b0 = 0; b0 < REG_MAX_SYMS; b0 += WAVE) {
    uint32_t i = b0 + lane;
    uint32_t pr = i < nsym ? lds[i] : 0u;
    if (pr > 4095) { over = true; one_at = i; }
    uint32_t tot;
    uint32_t ex = wave_excl_scan(pr, &tot);
    lds[i] = (pr << 12) | ((carry + ex) & 4095u);
    carry += tot;
    if (carry > 4096) over = true;
  }
  if (VLIST && carry == 4096 && __ballot(over)) {
    // A list of one symbol (frequency 4096, which the packed word cannot hold) -- the rule for the busiest context of a regular
    // mesh, where the valence of the gate vertex settles the symbol: the state of the coder never moves (x' = 4096 (x >> 12) +
    // (x & 4095)) and no byte is read, so the list is that symbol `count` times, once the stream's tail has passed rans_init.
    uint32_t st = 0, o = 0;
    if (!rans_init(stream + st_off_rans, st_size_rans, 16384, &st, &o)) { if (lane == 0) fail(D, ST_INVALID, 402); return; }
    const uint32_t one = (uint32_t)__builtin_ctzll(__ballot(one_at != DSA_INVALID));
    const uint32_t sym = rdlane(one_at, one);
    for (uint32_t i = lane; i < num_values; i += WAVE) out[i] = sym;
    WAIT_VM0();
    __syncthreads();
    if (lane == 0) atomicOr(&D->val_lists_done, 1u << ai);
    return;
  }
  if (__ballot(over) || carry != 4096) { if (lane == 0) fail(D, carry != 4096 ? ST_INVALID : ST_NOTIMPL, 401); return; }
  __syncthreads();
  // 3. slot tables through the attribute's scratch: every symbol writes its own slots
  for (uint32_t sy = lane; sy < nsym; sy += WAVE) {
    const uint32_t e = lds[sy], f = e >> 12, c = e & 4095u;
    for (uint32_t j = 0; j < f; ++j) { slot_tab[c + j] = f | (j << 16); slot_sym[c + j] = (uint16_t)sy; }   // {freq, rem - cum} as two 16-bit words (SDWA operands of the chain)
  }
  WAIT_VM0();
  __syncthreads();
  v32u tab_lo, tab_hi;
#pragma unroll
  for (int k = 0; k < 32; ++k) { tab_lo[k] = slot_tab[k * WAVE + lane]; tab_hi[k] = slot_tab[(k + 32) * WAVE + lane]; }
  // 4. initial state from the stream tail, RAnsDecoder.cs:20-54
  const uint8_t *buf = stream + st_off_rans;
  uint32_t x, off;
  {
    uint32_t st = 0, o = 0;
    bool ok = rans_init(buf, st_size_rans, 16384, &st, &o);
    if (!ok) { if (lane == 0) fail(D, ST_INVALID, 402); return; }
    x = uni(st); off = uni(o);
  }
  // byte window: lane l holds the dword at aligned byte (chunk*256 + 4l) of the stream
  const uint32_t mis = (uint32_t)((L.stream + st_off_rans) & 3u);
  const uint32_t *abuf = (const uint32_t *)(arena + (L.stream + st_off_rans - mis));
  uint32_t chunk = 0x7FFFFFFFu, W = 0;
  uint32_t mine = 0;
  WAIT_VM0();
  // 64 positions per outer iteration; the slot (low 12 bits of the state) of position j is parked in lane j
  // and stored with one coalesced store.  The per-symbol loop is hand-scheduled (RAnsDecoder.cs:56-65):
  //   * the state x and a 4-byte reservoir live in one SGPR pair {res, x}; a renormalisation byte is a
  //     single s_lshl_b64 of the pair (the stream is consumed from its tail, so an aligned little-endian
  //     dword holds the next 4 bytes most-significant first);
  //   * both halves of the slot table are read under one s_set_gpr_idx_on and selected by bit 11;
  //   * the block refills the reservoir itself (v_readlane from the 256-byte window in W) and leaves to C only for a new
  //     window, the unaligned first / last bytes of the stream, or when 64 positions are done.
  // The tables are pinned to v[16:79] so that the indexed v_mov can name their first register.
  uint64_t P = (uint64_t)x << 32;
  uint32_t rc = 0;                             // valid bytes in the reservoir (top-aligned in P's low dword)
  uint32_t dl = 0;                             // dword of the window the reservoir came from: the block refills itself from
                                               // the dword below it (whole dwords, same 256-byte window) and leaves to C otherwise
  bool exhausted = false;
  for (uint32_t i0 = 0; i0 < num_values; i0 += WAVE) {
    const uint32_t cnt = uni(num_values - i0 < WAVE ? num_values - i0 : WAVE);
    uint32_t j = 0;
    while (j < cnt) {
      if (!exhausted) {
        {
          // The 64 steps of a block are unrolled, so position J is parked in lane J with v_writelane (inline
          // lane number) and there is no loop counter: 4 scalar + 5 vector instructions per symbol.  A step is
          // entered by a computed jump (all steps have the same size), which is how decoding resumes at
          // position j after the reservoir was refilled.  The last block of a stream runs all 64 steps as well:
          // positions past the end decode whatever the state yields and are not stored.  The register-index window stays
          // open over the whole block (one s_set_gpr_idx_idx per step instead of an on/off pair): it indexes src1 only,
          // and every other vector instruction of a step has a constant or an SGPR there.  The packed table word is
          // consumed by two SDWA operations (freq = low word into the 24-bit multiply, rem - cum = high word into the add),
          // both reading the indexed register directly.  v_writelane sits between the VALU write of the new state and
          // its v_readlane (the wait state that hazard needs).
          uint32_t k6, va, vf;
          uint32_t js = uni(j);
          asm volatile(
              " s_getpc_b64 s[22:23]\n"
              "Lupc%=:\n"
              " s_mul_i32 %[k6], %[j], Lus1_%=-Lus0_%=\n"
              " s_add_u32 s22, s22, Lus0_%=-Lupc%=\n"
              " s_addc_u32 s23, s23, 0\n"
              " s_add_u32 s22, s22, %[k6]\n"
              " s_addc_u32 s23, s23, 0\n"
              " s_set_gpr_idx_on s22, gpr_idx(SRC1)\n"
              " s_setpc_b64 s[22:23]\n"
              ".irp J,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63\n"
              "Lus\\J\\()_%=:\n"
              " s_cmpk_lt_u32 s21, 0x4000\n"
              " s_cbranch_scc1 Lur\\J\\()_%=\n"
              "Lok\\J\\()_%=:\n"
              " s_bfe_u32 %[k6], s21, 0x60006\n"
              " s_set_gpr_idx_idx %[k6]\n"
              " v_lshrrev_b32_e64 %[vf], 12, s21\n"
              " v_mul_u32_u24_sdwa %[va], %[vf], v16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
              " v_add_u32_sdwa %[va], %[va], v16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
              " v_writelane_b32 %[mine], s21, \\J\n"
              " v_readlane_b32 s21, %[va], s21\n"
              ".endr\n"
              " s_movk_i32 %[j], 64\n"
              " s_branch Luend%=\n"
              ".irp J,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63\n"
              "Lur\\J\\()_%=:\n"
              " s_sub_u32 %[rc], %[rc], 1\n"
              " s_cbranch_scc1 Lue\\J\\()_%=\n"
              " s_lshl_b64 s[20:21], s[20:21], 8\n"
              " s_cmpk_lt_u32 s21, 0x4000\n"
              " s_cbranch_scc0 Lok\\J\\()_%=\n"
              " s_branch Lur\\J\\()_%=\n"
              "Lue\\J\\()_%=:\n"
              " s_cmp_lt_u32 %[off], 4\n"
              " s_cbranch_scc1 Lsl\\J\\()_%=\n"
              " s_sub_u32 %[dl], %[dl], 1\n"
              " s_cbranch_scc1 Lsl\\J\\()_%=\n"
              " v_readlane_b32 s20, %[W], %[dl]\n"
              " s_sub_u32 %[off], %[off], 4\n"
              " s_mov_b32 %[rc], 3\n"
              " s_lshl_b64 s[20:21], s[20:21], 8\n"
              " s_cmpk_lt_u32 s21, 0x4000\n"
              " s_cbranch_scc0 Lok\\J\\()_%=\n"
              " s_branch Lur\\J\\()_%=\n"
              "Lsl\\J\\()_%=:\n"
              " s_movk_i32 %[j], \\J\n"
              " s_branch Luempty%=\n"
              ".endr\n"
              "Luempty%=:\n"
              " s_mov_b32 %[rc], 0\n"
              "Luend%=:\n"
              " s_set_gpr_idx_off\n"
              : "+{s[20:21]}"(P), [rc] "+s"(rc), [j] "+s"(js), [mine] "+v"(mine), [k6] "=&s"(k6), [va] "=&v"(va), [vf] "=&v"(vf),
                [off] "+s"(off), [dl] "+s"(dl)
              : "{v[16:47]}"(tab_lo), "{v[48:79]}"(tab_hi), [W] "v"(W)
              : "vcc", "scc", "s22", "s23");
          j = js;
        }
        if (j < cnt) {                         // the state needs a byte and the reservoir is empty
          if (off == 0) exhausted = true;      // RAnsDecoder.cs:58-61: no bytes left, the state stays as it is
          else {
            const uint32_t end = off + mis;    // one past the next byte, in abuf coordinates
            const uint32_t d = (end - 1) >> 2, r = end - 4 * d, ch = d >> 6;
            if (ch != chunk) { chunk = ch; W = abuf[(size_t)ch * 64 + lane]; WAIT_VM0(); }
            const uint32_t res = rdlane(W, d & 63u) << (8 * (4 - r));
            rc = uni(r < off ? r : off);
            off -= rc;
            dl = uni(d & 63u);
            P = (P & 0xFFFFFFFF00000000ull) | res;
          }
        }
      } else {
        const uint32_t xs = uni((uint32_t)(P >> 32));
        mine = (lane == j) ? xs : mine;
        const uint32_t k5 = (xs >> 6) & 31u, l6 = xs & 63u;
        const uint32_t e0 = rdlane(tab_lo[k5], l6), e1 = rdlane(tab_hi[k5], l6);
        const uint32_t e = (xs & 2048u) ? e1 : e0;
        P = (uint64_t)((e & 0xFFFFu) * (xs >> 12) + (e >> 16)) << 32;
        ++j;
      }
    }
    if (lane < cnt) out[i0 + lane] = mine;
  }
  WAIT_VM0();
  __syncthreads();
  if (VLIST) {
    // 5. slot -> symbol of the list
    for (uint32_t i = lane; i < num_values; i += WAVE) out[i] = slot_sym[out[i] & 4095u];
    WAIT_VM0();
    __syncthreads();
    if (lane == 0) atomicOr(&D->val_lists_done, 1u << ai);
    return;
  }
  if (TAGS) {
    // 5. slot -> tag, as bytes into the output region; the bit total (tag x components, SymbolDecoding.cs:41-47) and the largest tag
    uint8_t *tags = arena + L.out[ai];
    uint64_t bits = 0;
    uint32_t worst = 0;
    const uint32_t nc = a.nc_portable;
    for (uint32_t i = lane; i < num_values; i += WAVE) {
      const uint32_t v = slot_sym[out[i] & 4095u];
      tags[i] = (uint8_t)v;
      bits += (uint64_t)v * nc;
      worst = v > worst ? v : worst;
    }
    for (int d = 32; d >= 1; d >>= 1) {
      bits += ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), d, WAVE) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)bits, d, WAVE);
      const uint32_t w2 = (uint32_t)__shfl_xor((int)worst, d, WAVE);
      worst = w2 > worst ? w2 : worst;
    }
    WAIT_VM0();
    __syncthreads();
    if (lane == 0) { a.table = (bits & 0x00FFFFFFFFFFFFFFull) | ((uint64_t)(worst > 255u ? 255u : worst) << 56); a.tags_done = 1; }
    return;
  }
  // 5. slot -> symbol (lane parallel), then zig-zag unless the transform's corrections are positive (D-4)
  const bool positive = a.have_scheme && (a.pred_transform == 2 || a.pred_transform == 3);
  for (uint32_t i = lane; i < num_values; i += WAVE) {
    uint32_t v = slot_sym[out[i] & 4095u];
    out[i] = positive ? v : ((v & 1u) ? (uint32_t)(-(int32_t)(v >> 1) - 1) : (v >> 1));
  }
  early_tail(arena, L, D, ai, flags);
  late_handoff(arena, L, D, ai, flags, 1u);
}

__global__ __launch_bounds__(WAVE, 6) void k_symbols_reg(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t flags) {
  uint32_t mesh = blockIdx.x, ai = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || ai >= D->num_attributes) return;
  reg_decode_stream<REG_ATTR>(arena, layouts[mesh], D, ai, flags);
}
// The tag stream in front of which the walk of a mesh stopped (see dsa_locate.h): one wave per mesh.
__global__ __launch_bounds__(WAVE, 6) void k_tags(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  uint32_t mesh = blockIdx.x;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || !D->values_pending || D->resume_att >= D->num_attributes) return;
  if (D->values_pending == 2) return;                                // (not stopped in front of a tag stream)
  reg_decode_stream<REG_TAGS>(arena, layouts[mesh], D, D->resume_att, 0u);
}
// The six context lists of valence-coded connectivity, a wave per list, in front of the connectivity waves that read them.
__global__ __launch_bounds__(WAVE, 6) void k_valence_lists(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  uint32_t mesh = blockIdx.x, c = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || D->encoder_type == 0 || D->traversal_type != 2) return;
  reg_decode_stream<REG_VLIST>(arena, layouts[mesh], D, c, 0u);
}

// Launched once per tier so that the LDS footprint of the cumulative table does not cap occupancy:
//   TIER 0: alphabets <= 64 (table in one register per lane) + tagged / fixed-width sources
//   TIER 1: alphabets <= 960      TIER 2: alphabets <= SYM_MAX_LDS and the large-alphabet fallback
// Which raw streams k_symbols_reg takes (12-bit precision, at most 4096 symbols more than one of which occurs, table scratch in the attribute's output region).
__device__ __forceinline__ bool sym_reg_eligible(const AttrDesc &a, const MeshLayout &L, uint32_t ai) {
  return a.source == SRC_RAW && a.precision_bits == 12 && a.num_symbols <= REG_MAX_SYMS && a.num_distinct > 1 &&
         L.out_cap[ai] >= 4096 * 6 + REG_MAX_SYMS * 4;
}
// Which raw streams k_symbols_wide takes: any precision, at most 2048 symbols to search -- those of the alphabet, or, for a sparse
// large alphabet (14-bit positions: 16 384 ids, about 2 000 used), its non-zero ones -- and room for the tables in global memory.
__device__ __forceinline__ bool sym_wide_eligible(const AttrDesc &a, const MeshLayout &L, uint32_t ai) {
  if (a.source != SRC_RAW || a.num_distinct <= 1 || sym_reg_eligible(a, L, ai)) return false;
  const bool compact = a.num_symbols > SYM_MAX_LDS;
  const uint32_t nse = compact ? a.num_distinct : a.num_symbols;
  if (nse <= 64 || nse > WIDE_MAX_SYMS || a.precision_bits > 16) return false;     // {cum, freq} packed in 16 + 16 bits
  return compact ? (a.table != 0 && 2ull * nse + 1 <= (unsigned long long)a.num_symbols + 2) : (L.out_cap[ai] >= 4ull * (nse + 1));
}

// =========================================================================
// k_symbols_wide: rANS decode of raw streams of any precision whose search table (<= 2048 cumulative frequencies) fits the
// register file: 32 VGPRs hold it, block b of 64 entries in register b, fetched by a uniform index (v_movrels) where k_symbols<2>
// reads LDS.  No LDS at all -- beside the chain waves a CU has little -- and no LDS round trip in the chain of a symbol.  The
// 15-bit-precision streams of 14-bit positions go here (compacted to their non-zero symbols), RAnsDecoder.cs:56-99.
// =========================================================================
__global__ __launch_bounds__(WAVE) void k_symbols_wide(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t flags) {
  const uint32_t mesh = blockIdx.x, ai = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || ai >= D->num_attributes) return;
  const MeshLayout &L = layouts[mesh];
  const AttrDesc &a = D->att[ai];
  if (a.source == SRC_BYTES) return;
  if (lanes::ln_sym_eligible(a, L, ai, flags)) return;
  if (sym_filtered(a, flags)) return;
  if (!sym_wide_eligible(a, L, ai)) return;
  const uint32_t lane = lane_id();
  const uint32_t num_values = a.num_entries * a.nc_portable;
  if (num_values == 0) return;
  const uint8_t *stream = arena + L.stream;
  const uint32_t stream_len = L.stream_len;
  uint32_t *out = (uint32_t *)(arena + L.work[ai]);
  const bool compact = a.num_symbols > SYM_MAX_LDS;
  const uint32_t nsym = compact ? a.num_distinct : a.num_symbols;
  uint32_t *symtab = compact ? (uint32_t *)(arena + a.table) : nullptr;
  uint32_t *gtab = compact ? symtab + nsym : (uint32_t *)(arena + L.out[ai]);          // nsym + 1 entries
  const uint32_t P = a.precision_bits, precision = 1u << P, l_base = precision * 4;
  // 1. probability table -> global scratch (lane 0), keeping the non-zero entries of a sparse alphabet (RAnsSymbolDecoder.cs:21-48)
  if (lane == 0) {
    Rd r(stream, stream_len, a.off_table);
    uint32_t k = 0;
    bool ok = true;
    for (uint32_t i = 0; i < a.num_symbols && ok; ++i) {
      const uint32_t pd = r.u8(), token = pd & 3u;
      if (token == 3u) {
        const uint32_t offset = pd >> 2;
        if (i + offset >= a.num_symbols) ok = false;
        if (!compact) for (uint32_t j = 0; j <= offset && i + j < a.num_symbols; ++j) gtab[i + j] = 0;
        i += offset;
      } else {
        uint32_t pr = pd >> 2;
        for (uint32_t j = 0; j < token; ++j) pr |= r.u8() << (8 * (j + 1) - 2);
        if (compact) { if (pr) { if (k >= nsym) ok = false; else { gtab[k] = pr; symtab[k] = i; ++k; } } }
        else gtab[i] = pr;
      }
    }
    if (!ok || !r.ok || (compact && k != nsym)) fail(D, ST_INVALID, 400);
    WAIT_VM0();
    __threadfence_block();
  }
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  // 2. cumulative frequencies, in place; entry nsym = precision
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nsym; b0 += WAVE) {
    const uint32_t i = b0 + lane;
    const uint32_t pr = i < nsym ? gtab[i] : 0u;
    uint32_t tot;
    const uint32_t ex = wave_excl_scan(pr, &tot);
    if (i < nsym) gtab[i] = carry + ex;
    carry += tot;
  }
  if (carry != precision) { if (lane == 0) fail(D, ST_INVALID, 401); return; }
  if (lane == 0) gtab[nsym] = precision;
  WAIT_VM0();
  __threadfence_block();
  __syncthreads();
  // 3. the table into registers: register r, lane l = {cum, freq} of entry 64 r + l in 16 + 16 bits (precision <= 16; a frequency of
  // 2^16 would be a one-symbol alphabet, which never comes here); beyond the alphabet 0xFFFFFFFF compares false.  Up to 15 bits of
  // precision the cumulative frequency is the HIGH half: one unsigned compare of the whole word with {rem, 0xFFFF} finds the
  // entries at or below rem, and the padding is above every such key (the hand-scheduled block below); at 16 bits the low half.
  const uint32_t nblocks = (nsym + WAVE - 1) / WAVE;
  const bool hand = uni(P) <= 15u;
  v32u tab;
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const uint32_t i = (uint32_t)r * WAVE + lane;
    const uint32_t c0 = i < nsym ? gtab[i] : 0x1FFFFu, c1 = i < nsym ? gtab[i + 1] : 0x1FFFFu;      // gtab[nsym] = precision
    // entries of zero frequency at the end of a non-compact table have cum = precision: at 16 bits that is 0x10000, whose low half
    // (0) would count for every rem -- they are padding as well (the reference's slot table, RAnsSymbolDecoder.cs:50-59, never maps to them)
    tab[r] = (i < nsym && c0 < 0x10000u && !(hand && c0 >= precision)) ? (hand ? ((c0 << 16) | (c1 - c0)) : (c0 | ((c1 - c0) << 16))) : 0xFFFFFFFFu;
  }
  const uint32_t coarse = lane < nblocks ? gtab[lane * WAVE] : 0xFFFFFFFFu;     // lane l: first cumulative frequency of block l
  // 4. initial state from the stream tail
  const uint8_t *buf = stream + a.off_rans;
  uint32_t x, off;
  {
    uint32_t st = 0, o = 0;
    const bool ok = rans_init(buf, a.size_rans, l_base, &st, &o);
    if (!ok) { if (lane == 0) fail(D, ST_INVALID, 402); return; }
    x = uni(st); off = uni(o);
  }
  const uintptr_t base_addr = (uintptr_t)buf;
  const uint32_t mis = (uint32_t)(base_addr & 3u);
  const uint32_t *abuf = (const uint32_t *)(base_addr - mis);
  uint32_t chunk = 0xFFFFFFFFu, W = 0;
  const uint32_t mask = precision - 1;
  uint32_t mine = 0;
  if (hand) {
    // The search as a hand-scheduled block in the manner of k_symbols_reg's (see there for the state / reservoir pair in s[20:21], the
    // computed entry into 64 unrolled steps, the refill from the window W): per symbol 17 scalar + 5 vector instructions (10 for the
    // most frequent symbol, see below) where the compiler's loop below issues 48 -- rem = x & mask; block b = count of block starts <= rem (one compare on `coarse`, s_bcnt1);
    // the block's register fetched through the index window (v_or with the indexed src1; v15 + b names it: b is 1-based);
    // entry = count of words <= {rem, 0xFFFF}; its word by v_readlane; x = freq (x >> P) + rem - cum on the scalar unit.
    // Parked per position: 64 b + entry (the post-pass takes the 64 off).
    uint64_t PR = (uint64_t)x << 32;
    uint32_t rc = 0, dl = 0;
    bool exhausted = false;
    const uint32_t lb = uni(l_base), mk = uni(mask), pb = uni(P);
    // the most frequent symbol is tried first, on the scalar unit alone (x = freq (x >> P) + rem - cum if rem - cum < freq: 10
    // instructions with the renormalisation test) -- two thirds of the corrections of 14-bit positions are one symbol
    uint32_t best = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const uint32_t w = tab[r], k = w == 0xFFFFFFFFu ? 0u : (((w & 0xFFFFu) << 11) | (uint32_t)(31 - r) << 6 | (63u - lane));
      best = k > best ? k : best;
    }
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)best, d, WAVE); best = o > best ? o : best; }
    const uint32_t idom = uni((31u - ((best >> 6) & 31u)) * WAVE + (63u - (best & 63u)));
    const uint32_t fdom = uni(best >> 11), cdom = uni(gtab[idom]), tdom = uni(idom + WAVE);
    WAIT_VM0();
    for (uint32_t i0 = 0; i0 < num_values; i0 += WAVE) {
      const uint32_t cnt = uni(num_values - i0 < WAVE ? num_values - i0 : WAVE);
      uint32_t j = 0;
      while (j < cnt) {
        if (!exhausted) {
          {
            uint32_t k6, rem, key, bb, q, t, jj, e, f, tmp;
            uint32_t js = uni(j);
            rc = uni(rc); off = uni(off); dl = uni(dl);
            PR = ((uint64_t)uni((uint32_t)(PR >> 32)) << 32) | uni((uint32_t)PR);
            asm volatile(
                " s_getpc_b64 s[22:23]\n"
                "Lwpc%=:\n"
                " s_mul_i32 %[k6], %[j], Lws1_%=-Lws0_%=\n"
                " s_add_u32 s22, s22, Lws0_%=-Lwpc%=\n"
                " s_addc_u32 s23, s23, 0\n"
                " s_add_u32 s22, s22, %[k6]\n"
                " s_addc_u32 s23, s23, 0\n"
                " s_set_gpr_idx_on s22, gpr_idx(SRC1)\n"
                " s_setpc_b64 s[22:23]\n"
                ".irp J,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63\n"
                "Lws\\J\\()_%=:\n"
                " s_cmp_lt_u32 s21, %[lb]\n"
                " s_cbranch_scc1 Lwr\\J\\()_%=\n"
                "Lwk\\J\\()_%=:\n"
                " s_and_b32 %[rem], s21, %[mk]\n"
                " s_sub_u32 %[e], %[rem], %[cdom]\n"
                " s_cmp_lt_u32 %[e], %[fdom]\n"
                " s_cbranch_scc0 Lwq\\J\\()_%=\n"
                " s_lshr_b32 %[q], s21, %[pb]\n"
                " s_mul_i32 %[q], %[q], %[fdom]\n"
                " v_writelane_b32 %[mine], %[tdom], \\J\n"
                " s_add_u32 s21, %[q], %[e]\n"
                "Lwn\\J\\()_%=:\n"
                ".endr\n"
                " s_movk_i32 %[j], 64\n"
                " s_branch Lwend%=\n"
                ".irp J,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63\n"
                "Lwq\\J\\()_%=:\n"
                " v_cmp_le_u32_e64 vcc, %[coarse], %[rem]\n"
                " s_pack_ll_b32_b16 %[key], 0xffff, %[rem]\n"
                " s_lshr_b32 %[q], s21, %[pb]\n"
                " s_bcnt1_i32_b64 %[bb], vcc\n"
                " s_set_gpr_idx_idx %[bb]\n"
                " s_lshl_b32 %[t], %[bb], 6\n"
                " v_or_b32_e32 %[tmp], 0, v15\n"
                " v_cmp_le_u32_e64 vcc, %[tmp], %[key]\n"
                " s_bcnt1_i32_b64 %[jj], vcc\n"
                " s_sub_u32 %[jj], %[jj], 1\n"
                " v_readlane_b32 %[e], %[tmp], %[jj]\n"
                " s_add_u32 %[t], %[t], %[jj]\n"
                " v_writelane_b32 %[mine], %[t], \\J\n"
                " s_and_b32 %[f], %[e], 0xffff\n"
                " s_lshr_b32 %[e], %[e], 16\n"
                " s_mul_i32 %[q], %[q], %[f]\n"
                " s_sub_u32 %[rem], %[rem], %[e]\n"
                " s_add_u32 s21, %[q], %[rem]\n"
                " s_branch Lwn\\J\\()_%=\n"
                ".endr\n"
                ".irp J,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63\n"
                "Lwr\\J\\()_%=:\n"
                " s_sub_u32 %[rc], %[rc], 1\n"
                " s_cbranch_scc1 Lwe\\J\\()_%=\n"
                " s_lshl_b64 s[20:21], s[20:21], 8\n"
                " s_cmp_lt_u32 s21, %[lb]\n"
                " s_cbranch_scc0 Lwk\\J\\()_%=\n"
                " s_branch Lwr\\J\\()_%=\n"
                "Lwe\\J\\()_%=:\n"
                " s_cmp_lt_u32 %[off], 4\n"
                " s_cbranch_scc1 Lwl\\J\\()_%=\n"
                " s_sub_u32 %[dl], %[dl], 1\n"
                " s_cbranch_scc1 Lwl\\J\\()_%=\n"
                " v_readlane_b32 s20, %[W], %[dl]\n"
                " s_sub_u32 %[off], %[off], 4\n"
                " s_mov_b32 %[rc], 3\n"
                " s_lshl_b64 s[20:21], s[20:21], 8\n"
                " s_cmp_lt_u32 s21, %[lb]\n"
                " s_cbranch_scc0 Lwk\\J\\()_%=\n"
                " s_branch Lwr\\J\\()_%=\n"
                "Lwl\\J\\()_%=:\n"
                " s_movk_i32 %[j], \\J\n"
                " s_branch Lwempty%=\n"
                ".endr\n"
                "Lwempty%=:\n"
                " s_mov_b32 %[rc], 0\n"
                "Lwend%=:\n"
                " s_set_gpr_idx_off\n"
                : "+{s[20:21]}"(PR), [rc] "+s"(rc), [j] "+s"(js), [mine] "+v"(mine), [k6] "=&s"(k6), [rem] "=&s"(rem), [key] "=&s"(key), [bb] "=&s"(bb),
                  [q] "=&s"(q), [t] "=&s"(t), [jj] "=&s"(jj), [e] "=&s"(e), [f] "=&s"(f), [tmp] "=&v"(tmp), [off] "+s"(off), [dl] "+s"(dl)
                : "{v[16:47]}"(tab), [W] "v"(W), [coarse] "v"(coarse), [lb] "s"(lb), [mk] "s"(mk), [pb] "s"(pb), [cdom] "s"(cdom), [fdom] "s"(fdom), [tdom] "s"(tdom)
                : "vcc", "scc", "s22", "s23");
            j = js;
          }
          if (j < cnt) {                         // the state needs a byte and the reservoir is empty
            if (off == 0) exhausted = true;      // RAnsDecoder.cs:58-61: no bytes left, the state stays as it is
            else {
              const uint32_t end = off + mis;    // one past the next byte, in abuf coordinates
              const uint32_t d = (end - 1) >> 2, r = end - 4 * d, ch = d >> 6;
              if (ch != chunk) { chunk = ch; W = abuf[(size_t)ch * 64 + lane]; WAIT_VM0(); }
              const uint32_t res = uni(rdlane(W, d & 63u) << (8 * (4 - r)));
              rc = uni(r < off ? r : off);
              off -= rc;
              dl = uni(d & 63u);
              PR = (PR & 0xFFFFFFFF00000000ull) | res;
            }
          }
        } else {
          const uint32_t xs = uni((uint32_t)(PR >> 32)), rem = xs & mk;
          const uint32_t b = uni((uint32_t)__popcll(__ballot(coarse <= rem)) - 1u);
          const uint32_t v = tab[b];
          const uint32_t jx = uni((uint32_t)__popcll(__ballot(v <= ((rem << 16) | 0xFFFFu))) - 1u);
          const uint32_t e = rdlane(v, jx);
          mine = (lane == j) ? (b + 1u) * WAVE + jx : mine;
          PR = (uint64_t)uni((e & 0xFFFFu) * (xs >> pb) + rem - (e >> 16)) << 32;
          ++j;
        }
      }
      if (lane < cnt) out[i0 + lane] = mine - WAVE;
    }
  } else {
  uint32_t res = 0, rc = 0;               // up to four bytes of the stream, the next one in the top bits (RAnsDecoder.cs:56-67 reads them one by one)
  for (uint32_t i = 0; i < num_values; ++i) {
    while (x < l_base) {
      if (rc == 0) {
        if (off == 0) break;              // no bytes left: the state stays as it is
        const uint32_t end = off + mis;   // one past the next byte, in abuf coordinates
        const uint32_t d = (end - 1) >> 2, r = end - 4 * d, ch = d >> 6;
        if (ch != chunk) { chunk = ch; W = abuf[(size_t)ch * 64 + lane]; }   // arena padding makes the over-read safe
        res = rdlane(W, d & 63u) << (8 * (4 - r));
        rc = uni(r < off ? r : off);
        off -= rc;
      }
      x = (x << 8) | (res >> 24);
      res <<= 8;
      --rc;
    }
    const uint32_t rem = x & mask;
    const uint32_t b = uni((uint32_t)__popcll(__ballot(coarse <= rem)) - 1u);
    const uint32_t v = tab[b];
    // (the padding beyond the alphabet, 0xFFFFFFFF, must not count: at 16-bit precision its low half, 65535, is a value rem takes)
    const uint32_t j = (uint32_t)__popcll(__ballot((v & 0xFFFFu) <= rem && v != 0xFFFFFFFFu)) - 1u;
    const uint32_t e = rdlane(v, j);
    x = (e >> 16) * (x >> P) + rem - (e & 0xFFFFu);
    mine = (i & 63u) == lane ? b * WAVE + j : mine;
    if ((i & 63u) == 63u) out[i - 63u + lane] = mine;
  }
  const uint32_t tail = num_values & 63u;
  if (tail && lane < tail) out[num_values - tail + lane] = mine;
  }
  WAIT_VM0();
  __threadfence_block();
  __syncthreads();
  // 5. compact index -> symbol id, then zig-zag unless the transform's corrections are positive (D-4)
  const bool positive = a.have_scheme && (a.pred_transform == 2 || a.pred_transform == 3);
  for (uint32_t i = lane; i < num_values; i += WAVE) {
    uint32_t v = out[i];
    if (compact) v = symtab[v];
    out[i] = positive ? v : ((v & 1u) ? (uint32_t)(-(int32_t)(v >> 1) - 1) : (v >> 1));
  }
}

// One (mesh, attribute) item of k_symbols<TIER>.
template <int TIER>
__device__ __forceinline__ void symbols_tier_item(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t mesh, uint32_t ai, uint32_t flags, uint32_t *lds_cum) {
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || ai >= D->num_attributes) return;
  const MeshLayout &L = layouts[mesh];
  const AttrDesc &a = D->att[ai];
  if (a.source == SRC_BYTES) return;
  if (lanes::ln_sym_eligible(a, L, ai, flags)) return;             // k_symbols_lanes
  if (sym_filtered(a, flags)) return;                              // the other launch of the early / late pair
  // a sparse large alphabet (14-bit positions: 16 384 ids, a few thousand of them used) is searched through its non-zero
  // symbols; the table k_locate reserved for the serial fallback holds the compact -> symbol map instead
  const bool compact = a.source == SRC_RAW && a.num_symbols > SYM_MAX_LDS && a.num_distinct <= SYM_MAX_LDS && a.num_distinct >= 1 && a.table != 0;
  {
    const uint32_t ns = a.source == SRC_RAW ? a.num_symbols : 0u;
    if (sym_reg_eligible(a, L, ai)) return;    // k_symbols_reg
    if ((flags & SYM_WIDE) && sym_wide_eligible(a, L, ai)) return;   // k_symbols_wide
    const uint32_t nse = compact ? a.num_distinct : ns;
    const int tier = nse <= 64 ? 0 : (nse <= 960 ? 1 : 2);
    if (tier != TIER) return;
  }
  const uint8_t *s = arena + L.stream;
  uint32_t *work = (uint32_t *)(arena + L.work[ai]);
  const uint32_t lane = lane_id();
  const uint32_t nc = a.nc_portable;
  const uint32_t num_values = a.num_entries * nc;
  if (num_values == 0) return;
  if (a.source == SRC_RAW) {
    if (a.num_symbols <= SYM_MAX_LDS) rans_decode_wave(D, s, L.stream_len, a, num_values, work, lds_cum, nullptr);
    else if (compact) rans_decode_wave(D, s, L.stream_len, a, num_values, work, lds_cum, (uint32_t *)(arena + a.table));
    else rans_decode_serial(D, s, L.stream_len, a, num_values, work, (uint32_t *)(arena + a.table));
  } else if (a.source == SRC_TAGGED) {
    // SymbolDecoding.cs:38-47: entry e owns nc fields of tags[e] bits, packed LSB-first in entry order
    const uint8_t *tags = arena + L.out[ai];
    const uint8_t *bits = s + a.off_bits;
    const uint32_t nbytes = L.stream_len - a.off_bits;
    uint64_t base_bits = 0;
    for (uint32_t e0 = 0; e0 < a.num_entries; e0 += WAVE) {
      uint32_t e = e0 + lane;
      uint32_t t = e < a.num_entries ? tags[e] : 0u;
      uint32_t tot;
      uint32_t ex = wave_excl_scan(t * nc, &tot);
      if (e < a.num_entries) {
        uint64_t bp = base_bits + ex;
        for (uint32_t c = 0; c < nc; ++c) { work[e * nc + c] = t ? read_bits(bits, nbytes, bp, t) : 0u; bp += t; }
      }
      base_bits += tot;
    }
  } else {   // SRC_FIXED
    const uint8_t *raw = s + a.off_raw;
    for (uint32_t i = lane; i < num_values; i += WAVE) {
      uint32_t v = 0;
      for (uint32_t k = 0; k < a.fixed_bytes; ++k) v |= (uint32_t)raw[(size_t)i * a.fixed_bytes + k] << (8 * k);
      work[i] = v;
    }
  }
  __syncthreads();
  if (status_of(D) != ST_OK) return;
  // zig-zag unless the transform's corrections are positive (D-4); BitUtilities.cs:94-103
  bool positive = a.have_scheme && (a.pred_transform == 2 || a.pred_transform == 3);
  if (!positive)
    for (uint32_t i = lane; i < num_values; i += WAVE) {
      uint32_t v = work[i];
      work[i] = (v & 1u) ? (uint32_t)(-(int32_t)(v >> 1) - 1) : (v >> 1);
    }
}
// The tiers are launched for every batch, and almost always find nothing to do; a wave needs its registers and LDS (136 VGPRs,
// 17 KB for tier 2) even to find that out, which beside the chain and decoder waves it gets one at a time.  So a launch is a
// fixed number of waves that walk the (mesh, attribute) items with a stride, instead of one wave per item.
#define SYM_TIER_BLOCKS 2048u
// The cumulative table is dynamic LDS (sym_tier_lds_bytes at launch): with a size the compiler can see it allocates the registers
// that size's occupancy would allow -- 136 for the 16 KB tier where the kernel needs 46 -- and the waves of a tier that has
// nothing to do then wait, 8 ms on the bench batch, for SIMDs with that many registers free.
constexpr uint32_t sym_tier_lds_bytes(int tier) { return ((tier == 0 ? 64u : (tier == 1 ? 960u : (uint32_t)SYM_MAX_LDS)) + WAVE + 1u) * 4u; }
extern __shared__ __attribute__((aligned(16))) uint32_t sh_tier[];
template <int TIER>
__global__ __launch_bounds__(WAVE, 8) void k_symbols(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t na, uint32_t flags) {
  uint32_t *lds_cum = sh_tier;
  const uint64_t items = (uint64_t)n * na;
  for (uint64_t it = blockIdx.x; it < items; it += gridDim.x) {
    symbols_tier_item<TIER>(arena, layouts, descs, (uint32_t)(it % n), (uint32_t)(it / n), flags, lds_cum);
    __syncthreads();                                     // the next item reuses the LDS table
  }
}

// k_register_gate: waves that do nothing but need 136 vector registers to be placed.  On a stream it holds back what follows until
// SIMDs have that many free -- on a machine filled by an earlier launch of 80-register decoders beside the 64-register chain
// waves, until that launch has no block left waiting for a slot (dsa_api.hip: the crowded-batch schedule of the symbol kernels).
__global__ __launch_bounds__(WAVE) void k_register_gate() { asm volatile("v_mov_b32 v135, 0" ::: "v135"); }

// =========================================================================
// k_predict: inverse prediction, in place on the work buffer.
// =========================================================================
// Wrap-transform schemes (Difference / Parallelogram + Wrap) are decoded 64 entries per step:
//   * lane i owns entry p0+i.  Lane 0's prediction only uses finished entries and is computed with the
//     reference's exact formula.  Lane i >= 1 joins the run if its prediction is "previous entry +
//     (finished entry - finished entry)" (parallelogram whose Next or Previous operand is entry p-1, or
//     the delta fallback), so that along the run   o[p] = adjust(o[p-1] + g[p] + corr[p]).
//   * while no prediction is clamped this is a prefix sum modulo max_dif: one wave scan gives every
//     o[p] of the run;
//   * every lane then re-evaluates the reference's sequential step (clamp, add, single +-max_dif) from
//     its neighbour's value and the run is cut at the first lane that disagrees -- so the result is the
//     sequential result by induction, and each step finishes at least one entry.
// The octahedral normal transform is decoded 64 entries per chunk with the corrections in registers
// (readlane) and coalesced loads/stores; its step is a short scalar program.
__device__ __forceinline__ uint32_t addmod(uint32_t a, uint32_t b, uint32_t m) { uint32_t r = a + b; return r >= m ? r - m : r; }
// phase 0: schemes that need no traversal data (difference / octahedral delta) -- launched behind the symbol
// kernels on their stream; phase 1: parallelogram schemes, after the traversal.
__device__ __forceinline__ bool wrap_fast_ok(const AttrDesc &a, uint32_t flags);
__device__ __forceinline__ bool pw_dequant_fused(const AttrDesc &a, uint32_t flags);
#define OS_FLAG 16u   // the canonicalised octahedral delta is k_predict_oct_streams' (crowded batches)
__device__ __forceinline__ bool oct_stream_eligible(const AttrDesc &a) {
  return a.have_scheme && a.source != SRC_BYTES && a.pred_transform == 3 && a.pred_kind == 0 && a.corner_data == 0 && a.num_entries != 0 && !a.early_done &&
         a.oct_max_q >= 3 && a.oct_max_q < (1 << OCT_PK_MAX_BITS);      // the packed step's range; finer octahedra stay with k_predict
}
// The body of k_predict for one attribute on one wave (also the tail of an entropy-decode wave, see early_tail).
__device__ __forceinline__ void predict_wave(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t phase, uint32_t flags) {
  const AttrDesc &a = D->att[ai];
  if (!a.have_scheme || a.source == SRC_BYTES) return;
  if (wrap_fast_ok(a, flags)) return;                    // k_predict_wrap
  if (lanes::ln_oct_eligible(a, flags)) return;          // k_predict_oct_lanes
  if ((flags & OS_FLAG) && oct_stream_eligible(a)) return;   // k_predict_oct_streams
  if (a.pred_kind == 2 || a.pred_kind == 3 || a.pred_kind == 4) return;            // k_predict_geometric, k_texcoords, k_multipara
  if (att_is_late(a) != (phase == 1) || pred_filtered(a, flags)) return;
  int32_t *w = (int32_t *)(arena + L.work[ai]);
  const uint32_t nc = a.nc_portable, entries = a.num_entries;
  const uint32_t lane = lane_id();
  if (entries == 0) return;
  if (a.pred_transform == 1) {
    if (nc > 4) { if (lane == 0) fail(D, ST_NOTIMPL, 500); return; }
    const int32_t mn = a.wrap_min, mx = a.wrap_max, max_dif = 1 + mx - mn;
    const uint32_t M = (uint32_t)max_dif;
    const bool para_mode = a.pred_kind == 1;
    const uint32_t *para = att_para(arena, L, D, a);
    // Every lane's prediction has the form  base + (o[ga] - o[gb])  with finished entries ga, gb (or none):
    //   lane 0:   base = o[p-1] (delta, or a parallelogram that contains entry p-1), o[next] for a parallelogram of
    //             three older entries, 0 for the first entry -- read from memory;
    //   lane i>0: base = the value of lane i-1 in this run -- only entries whose prediction contains p-1 (or falls
    //             back to delta) and whose other operands are finished can join the run.
    // Small ranges (M < 2^25) scan plain sums with one DPP add per step and reduce modulo M once.
    const bool small_m = M < (1u << 25);
    const float inv_m = 1.0f / (float)M;
    uint32_t p0 = 0;
    int32_t last[4] = {0, 0, 0, 0};
    bool have_last = false;
    while (p0 < entries) {
      const uint32_t p = p0 + lane;
      const bool live = p < entries;
      uint32_t en = DSA_INVALID, ep = 0, eo = 0;
      if (live && para_mode && p > 0) { en = para[3 * p]; ep = para[3 * p + 1]; eo = para[3 * p + 2]; }
      uint32_t ga = DSA_INVALID, gb = DSA_INVALID, bidx = DSA_INVALID;
      bool chain = false;
      if (live) {
        if (en == DSA_INVALID) { chain = true; }                                            // delta: pred = o[p-1]
        else if (en == p - 1) { ga = ep; gb = eo; chain = ep < p0 && eo < p0; }
        else if (ep == p - 1) { ga = en; gb = eo; chain = en < p0 && eo < p0; }
        else { ga = ep; gb = eo; bidx = en; }                                               // three older entries
        if (bidx == DSA_INVALID && p > 0) bidx = p - 1;
      }
      // run = lane 0 + leading chain lanes
      const uint64_t not_chain = __ballot(!chain) & ~1ull;
      uint32_t run = not_chain ? (uint32_t)__builtin_ctzll(not_chain) : WAVE;
      if (p0 + run > entries) run = entries - p0;
      const bool in_run = lane < run;
      int32_t corr[4], g[4], o[4];
      uint32_t xr[4];
      bool irregular = false;
      // lane 0's base is usually the value the previous step ended on: it is carried in `last` instead of being
      // read back from memory behind that step's store
      const bool base_from_mem = lane == 0 && bidx != DSA_INVALID && !(have_last && bidx == p0 - 1);
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c) {
        corr[c] = 0; g[c] = 0; o[c] = 0; xr[c] = 0;
        if (c >= nc) continue;
        if (in_run) {
          corr[c] = w[p * nc + c];
          if (ga != DSA_INVALID) g[c] = (int32_t)((uint32_t)w[ga * nc + c] - (uint32_t)w[gb * nc + c]);
        }
        // lane 0: the reference's step from memory operands (exact whatever happens to the rest of the run)
        int32_t base0 = base_from_mem ? w[bidx * nc + c] : 0;
        if (lane == 0 && bidx != DSA_INVALID && !base_from_mem) base0 = last[c];
        const int32_t o0 = wrap_original((int32_t)((uint32_t)base0 + (uint32_t)g[c]), corr[c], mn, mx, max_dif);
        if (lane == 0) {
          o[c] = o0;
          const uint32_t e0 = (uint32_t)o0 - (uint32_t)mn;
          if (o0 < mn || e0 >= M) irregular = true; else xr[c] = e0;
        } else if (in_run) {
          // residue of g + corr without division: real corrections keep |g + corr| < 2M
          const int64_t e64 = (int64_t)g[c] + (int64_t)corr[c];
          int32_t e = (int32_t)e64;
          if (e64 <= -2 * (int64_t)M || e64 >= 2 * (int64_t)M) irregular = true;
          else {
            if (e < 0) e += (int32_t)M;
            if (e < 0) e += (int32_t)M;
            if (e >= (int32_t)M) e -= (int32_t)M;
            xr[c] = (uint32_t)e;
          }
        }
      }
      {
        const uint64_t irr = __ballot(irregular && in_run);
        if (irr & 1ull) run = 1;                                   // lane 0 keeps its exact value, nobody chains on it
        else if (irr) { const uint32_t f = (uint32_t)__builtin_ctzll(irr); if (f < run) run = f; }
      }
      const bool in_scan = lane < run;
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c) {
        if (c >= nc) continue;
        uint32_t x = in_scan ? xr[c] : 0u;
        if (small_m) {
          x = wave_incl_scan(x, [](uint32_t a, uint32_t b) { return a + b; });             // < 64 M < 2^31
          uint32_t q = (uint32_t)((float)x * inv_m);
          int32_t rres = (int32_t)(x - q * M);
          if (rres < 0) rres += (int32_t)M;
          if (rres >= (int32_t)M) rres -= (int32_t)M;
          x = (uint32_t)rres;
        } else {
          x = wave_incl_scan(x, [M](uint32_t a, uint32_t b) { return addmod(a, b, M); });
        }
        if (lane > 0) o[c] = (int32_t)((uint32_t)mn + x);
      }
      // verify against the sequential step: pred = o[p-1] + g (uint32 arithmetic), clamp must be a no-op
      bool good = true;
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c) {
        if (c >= nc) continue;
        int32_t prev = (int32_t)lane_prev((uint32_t)o[c]);
        if (lane > 0 && in_scan) {
          int32_t pred = (int32_t)((uint32_t)prev + (uint32_t)g[c]);
          if (pred < mn || pred > mx || wrap_original(pred, corr[c], mn, mx, max_dif) != o[c]) good = false;
        }
      }
      uint64_t bad = __ballot(in_scan && !good);
      if (bad) run = (uint32_t)__builtin_ctzll(bad);     // >= 1: lane 0 is always exact
      if (lane < run) {
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) if (c < nc) w[p * nc + c] = o[c];
      }
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c) last[c] = (int32_t)rdlane((uint32_t)o[c], run - 1);
      have_last = true;
      p0 += run;
    }
    (void)para_mode;
  } else {
    // normal octahedron transforms are 2-component; only Difference reaches here (k_locate rejects the rest)
    if (a.pred_kind != 0) { if (lane == 0) fail(D, ST_NOTIMPL, 501); return; }
    OctParams o;
    int q = 32 - __clz(a.oct_max_q);
    int32_t max_value = (1 << q) - 2;
    o.center = max_value / 2;
    o.max_q = (1 << q) - 1;
    const bool canonical = a.pred_transform == 3;
    const uint32_t M = (uint32_t)o.max_q;
    // In the canonical frame of the predicted value (diamond inversion + rotation into the bottom-left
    // quadrant) the delta recurrence is  w[p] = mod_max(w[p-1] + corr[p])  as long as consecutive values
    // stay in the same class, i.e. a prefix sum modulo max_q.  Per step: lane 0 is computed exactly, lanes
    // 1..63 take the scan value mapped back with lane 0's class, then every lane re-evaluates the exact
    // transform from its neighbour's value and the run is cut at the first disagreement.  When runs get
    // short (values hopping between classes) the chunk falls back to the sequential loop.
    int32_t ps = 0, pt = 0;           // o[p0-1]
    uint32_t p0 = 0, short_runs = 0;
    while (p0 < entries) {
      const uint32_t p = p0 + lane;
      const bool live = p < entries;
      int2 cv = make_int2(0, 0);
      if (live) cv = ((const int2 *)w)[p];
      const uint32_t cnt = entries - p0 < WAVE ? entries - p0 : WAVE;
      if (short_runs < 2) {
        // class of the prediction of lane 0 and its canonical value
        int32_t us = ps - o.center, ut = pt - o.center;
        const int32_t aus = us < 0 ? -us : us, aut = ut < 0 ? -ut : ut;
        const bool in_d = (uint32_t)aus + (uint32_t)aut <= (uint32_t)o.center;
        if (!in_d) oct_invert_diamond(o.center, us, ut);
        bool bl = true;
        int rot = 0;
        if (canonical) {
          bl = (us == 0 && ut == 0) || (us < 0 && ut <= 0);
          if (us == 0) rot = ut == 0 ? 0 : (ut > 0 ? 3 : 1);
          else if (us > 0) rot = ut >= 0 ? 2 : 1;
          else rot = ut <= 0 ? 0 : 3;
          if (!bl) oct_rotate(us, ut, rot);
        }
        // canonical values along the run: w_i = mod_max(u + corr_0 + ... + corr_i)
        // us, ut are in [-center, center] whenever the previous value is a valid octahedral coordinate, so
        // the residues need no division; corrections outside [0, M) fail their lane's check below
        const uint32_t xs = (uint32_t)(us + o.center), xt = (uint32_t)(ut + o.center);
        if (xs >= M || xt >= M) { short_runs = 2; continue; }     // out-of-range state: exact sequential chunk
        uint32_t ss = (live && (uint32_t)cv.x < M) ? (uint32_t)cv.x : 0u, st = (live && (uint32_t)cv.y < M) ? (uint32_t)cv.y : 0u;
        if (lane == 0) { ss = addmod(ss, xs, M); st = addmod(st, xt, M); }
        if (M < (1u << 25)) {       // plain sums (< 64 M) with one DPP add per step, then a single modulo
          const float inv_m = 1.0f / (float)M;
          ss = wave_incl_scan(ss, [](uint32_t a, uint32_t b) { return a + b; });
          st = wave_incl_scan(st, [](uint32_t a, uint32_t b) { return a + b; });
          int32_t rs = (int32_t)(ss - (uint32_t)((float)ss * inv_m) * M), rt = (int32_t)(st - (uint32_t)((float)st * inv_m) * M);
          if (rs < 0) rs += (int32_t)M;
          if (rs >= (int32_t)M) rs -= (int32_t)M;
          if (rt < 0) rt += (int32_t)M;
          if (rt >= (int32_t)M) rt -= (int32_t)M;
          ss = (uint32_t)rs; st = (uint32_t)rt;
        } else {
          ss = wave_incl_scan(ss, [M](uint32_t a, uint32_t b) { return addmod(a, b, M); });
          st = wave_incl_scan(st, [M](uint32_t a, uint32_t b) { return addmod(a, b, M); });
        }
        int32_t os = (int32_t)ss - o.center, ot = (int32_t)st - o.center;     // w in [-center, center]
        if (canonical && !bl) oct_rotate(os, ot, (4 - rot) % 4);
        if (!in_d) oct_invert_diamond(o.center, os, ot);
        os += o.center; ot += o.center;
        // exact re-evaluation from the neighbour's value
        int32_t prs = (int32_t)lane_prev((uint32_t)os), prt = (int32_t)lane_prev((uint32_t)ot);
        if (lane == 0) { prs = ps; prt = pt; }
        int32_t es, et;
        oct_original(o, canonical, prs, prt, cv.x, cv.y, es, et);
        const bool good = live && es == os && et == ot && (uint32_t)cv.x < M && (uint32_t)cv.y < M;
        uint32_t K = leading_lanes(good);
        if (K == 0) {               // lane 0 disagrees only if a correction is out of range: take its exact value
          K = 1;
          os = es; ot = et;
        }
        if (lane < K) ((int2 *)w)[p] = make_int2(os, ot);
        ps = (int32_t)rdlane((uint32_t)os, K - 1); pt = (int32_t)rdlane((uint32_t)ot, K - 1);
        p0 += K;
        short_runs = K < 8 ? short_runs + 1 : 0;
        continue;
      }
      // sequential chunk (PredictionSchemeDeltaDecoder.cs:23-37), corrections in registers
      int32_t rs = 0, rt = 0;
      for (uint32_t i = 0; i < cnt; ++i) {
        const int32_t c0 = (int32_t)rdlane((uint32_t)cv.x, i), c1 = (int32_t)rdlane((uint32_t)cv.y, i);
        int32_t os, ot;
        oct_original(o, canonical, ps, pt, c0, c1, os, ot);
        if (lane == i) { rs = os; rt = ot; }
        ps = os; pt = ot;
      }
      if (live) ((int2 *)w)[p] = make_int2(rs, rt);
      p0 += cnt;
      short_runs = 0;
    }
  }
}

__global__ __launch_bounds__(WAVE) void k_predict(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t phase, uint32_t flags) {
  const uint32_t mesh = blockIdx.x, ai = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || ai >= D->num_attributes) return;
  if (phase == 0 && D->att[ai].early_done) return;
  // (issue priority for the early attributes' prediction -- the serial octahedral chains their strand ends on -- was measured: the
  // late symbols lose what it gains)
#ifdef DSA_PREDICT_PRIO
  if (phase == 0) __builtin_amdgcn_s_setprio(DSA_PREDICT_PRIO);
#endif
  predict_wave(arena, layouts[mesh], D, ai, phase, flags);
}

// =========================================================================
// k_predict_oct_streams: the canonicalised octahedral delta (PredictionSchemeDeltaDecoder.cs:23-37 over
// PredictionSchemeNormalOctahedronCanonicalizedDecodingTransform.ComputeOriginalValue) of a crowded batch, ONE LANE PER
// STREAM.  The chain has nothing a wave could scan when consecutive normals keep changing their class (which side of the
// diamond, which quadrant: height fields straddle an axis all the time) -- k_predict then retires three entries per 64-lane
// step, 2 G scalar + 2 G vector instructions per 4096-mesh batch at the tail of the early strand.  With both components in the
// 16-bit halves of one register (dsa_common.h: oct_pk_step, about 35 packed operations per entry, no branches but one wave-uniform
// one) on 64 streams per wave the batch needs 1/30 of the instructions, and the time of the kernel is the length of the longest
// stream's chain, whatever the batch size.
// A lone wave cannot hide a memory round trip behind other waves, so the corrections come through a ring in LDS filled by
// LDS-DMA (global_load_lds: no registers in flight): slot s = 64 lanes x 16 bytes = the next two entries of every lane's own
// stream, requested OS_AHEAD slots before they are used (a gather of 64 lines per instruction -- a cycle of the address unit
// each, 1/6 of what the recursion costs -- every line serving eight consecutive requests from L1 / L2).  Results leave with a
// 16-byte store per lane and slot.  The streams of a wave may differ in length: finished lanes keep requesting their last
// pair (the count of requests in flight must not depend on the lane) and store nothing.
// =========================================================================
#ifndef DSA_OCT_PRIO
#define DSA_OCT_PRIO 3
#endif
#define OS_RING 32u       // slots of 1 KB
#define OS_AHEAD 28u      // requests in flight (vmcnt counts 63 at most; stores share the counter)
static_assert(OS_AHEAD < OS_RING && OS_AHEAD < 60u, "the slot being filled is never the one being read");
__global__ __launch_bounds__(WAVE, 8) void k_predict_oct_streams(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  // the ring is dynamic LDS (OS_RING KB at launch): a size the compiler can see makes it budget registers for the two waves per
  // SIMD that much LDS allows (176), and the wave then finds no room beside the decoders
  extern __shared__ __attribute__((aligned(1024))) uint8_t ring[];
  const uint32_t lane = lane_id(), mesh = blockIdx.x * WAVE + lane, ai = blockIdx.y;
  uint32_t entries = 0;
  uint8_t *w = arena;                 // lanes without a stream read the front of the arena (always there) and never store
  OctParams o;
  uint32_t q = 2;
  if (mesh < n) {
    const MeshDesc *D = &descs[mesh];
    if (D->status == ST_OK && !D->general && ai < D->num_attributes && oct_stream_eligible(D->att[ai])) {
      const AttrDesc &a = D->att[ai];
      entries = a.num_entries;
      w = arena + layouts[mesh].work[ai];
      q = 32u - (uint32_t)__clz(a.oct_max_q);
    }
  }
  o.center = ((1 << q) - 2) / 2;
  o.max_q = (1 << q) - 1;
  const uint32_t pairs = (entries + 1u) / 2u;
  uint32_t most = pairs;
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)most, d, WAVE); most = x > most ? x : most; }
  most = uni(most);
  if (most == 0) return;
  __builtin_amdgcn_s_setprio(DSA_OCT_PRIO);      // a handful of long chains
  const uint32_t last = pairs ? pairs - 1u : 0u;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto request = [&](uint32_t pr) {
    const uint32_t at = pr < last ? pr : last;
    __builtin_amdgcn_global_load_lds((glb_void *)(w + 16ull * at), (lds_void *)(ring + (pr % OS_RING) * 1024u), 16, 0, 0);
  };
  for (uint32_t s = 0; s < OS_AHEAD; ++s) request(s);
  OctPkLane st;
  oct_pk_init(st, o, q);
  const uint32_t lds_lane = (uint32_t)(uintptr_t)(lds_void *)ring + lane * 16u;
  for (uint32_t i = 0; i < most; ++i) {
    request(i + OS_AHEAD);
    // the request for slot i is OS_AHEAD requests old: at most that many memory operations may still be under way (stores
    // issued in between only make the wait stricter)
    uint4 c;
    asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(lds_lane + (i % OS_RING) * 1024u), "n"(OS_AHEAD) : "memory");
    // both entries of the slot on the packed step (dsa_common.h), straight-line but for one wave-uniform choice per entry (does
    // any lane's previous value lie outside the diamond: the step then carries the two inversions); a lane that was not
    // entitled to it -- a correction outside [0, max_q], a value that left the square -- does its entries again by the
    // reference's function
    const bool entitled = !st.wild && (((c.x | c.y | c.z | c.w) >> q) == 0u);
    uint32_t A, SA;
    uint32_t OUT = oct_pk_outside(st.Cv, st.P, A, SA);
    const uint32_t C1 = c.x | (c.y << 16), C2 = c.z | (c.w << 16);
    const uint32_t P1 = __builtin_amdgcn_ballot_w64(OUT != 0u) ? oct_pk_step<true>(st.Cv, st.Mv, st.P, A, SA, OUT, C1) : oct_pk_step<false>(st.Cv, st.Mv, st.P, A, SA, OUT, C1);
    OUT = oct_pk_outside(st.Cv, P1, A, SA);
    const uint32_t P2 = __builtin_amdgcn_ballot_w64(OUT != 0u) ? oct_pk_step<true>(st.Cv, st.Mv, P1, A, SA, OUT, C2) : oct_pk_step<false>(st.Cv, st.Mv, P1, A, SA, OUT, C2);
    const uint32_t v1 = pk_add(P1, st.Cv), v2 = pk_add(P2, st.Cv);
    int32_t a0 = (int32_t)(v1 & 0xFFFFu), a1 = (int32_t)(v1 >> 16), b0 = (int32_t)(v2 & 0xFFFFu), b1 = (int32_t)(v2 >> 16);
    const bool both = 2u * i + 1u < entries;
    if (entitled) st.P = P2;
    else if (i < pairs) {
      oct_pk_entry(st, o, (int32_t)c.x, (int32_t)c.y, a0, a1);
      if (both) oct_pk_entry(st, o, (int32_t)c.z, (int32_t)c.w, b0, b1);
    }
    if (both) *(uint4 *)(w + 16ull * i) = make_uint4((uint32_t)a0, (uint32_t)a1, (uint32_t)b0, (uint32_t)b1);
    else if (i < pairs) *(uint2 *)(w + 16ull * i) = make_uint2((uint32_t)a0, (uint32_t)a1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing may land in the ring after the wave has given it back
}

// =========================================================================
// k_predict_wrap: the wrap-transform schemes (Difference / Parallelogram + Wrap) with few components and a range
// below 2^25 -- every quantised attribute -- on a leaner form of k_predict's step: one instantiation per component
// count, every load unconditional (indices clamped, a delta entry loads o[0] - o[0]), the scan as six DPP adds per
// component, residues by one multiply with 1/M.  Same invariant: lane 0 is the reference's exact step from
// finished entries, lanes 1.. join while their prediction is "previous entry + (finished - finished)", every lane
// re-evaluates the sequential step from its neighbour's value and the run is cut at the first disagreement.
// =========================================================================
__device__ __forceinline__ bool pw_dequant_fused(const AttrDesc &a, uint32_t flags) {
  return wrap_fast_ok(a, flags) && a.seq_type == 2 && a.nc == a.nc_portable && a.q_bits >= 1 && a.q_bits <= 30;
}
__device__ __forceinline__ bool wrap_fast_ok(const AttrDesc &a, uint32_t flags) {
  return (flags & PW_FLAG) && a.have_scheme && a.source != SRC_BYTES && a.pred_transform == 1 && a.pred_kind != 3 && a.pred_kind != 4 && a.nc_portable >= 1 && a.nc_portable <= 4 &&
         (uint32_t)(1 + a.wrap_max - a.wrap_min) < (1u << 25) && a.num_entries != 0;
}
// inclusive wave64 prefix sum, one v_add with a DPP operand per step where the backend fuses them
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);    // row_shr:1 (lanes without a source add 0)
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
  return x;
}
template <int NC>
struct PwVec { int32_t v[NC]; };

template <int NC, bool PARA>
__device__ __forceinline__ void predict_wrap_wave(int32_t *w, const uint32_t *para, uint32_t entries, int32_t mn, int32_t mx, float *out, float delta, const float *qmin) {
  typedef PwVec<NC> V;
  const V *wv = (const V *)w;
  const uint32_t lane = lane_id();
  const uint32_t M = (uint32_t)(1 + mx - mn);
  const int32_t max_dif = (int32_t)M;
  const float inv_m = 1.0f / (float)M;
  const uint32_t lastp = entries - 1;
  uint32_t p0 = 0;
  int32_t last[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) last[c] = 0;
  while (p0 < entries) {
    const uint32_t p = p0 + lane;
    const bool live = p <= lastp;
    const uint32_t pc = live ? p : lastp;
    uint32_t en = DSA_INVALID, ep = 0, eo = 0;
    if (PARA) { const PwVec<3> t = *(const PwVec<3> *)(para + 3 * (size_t)pc); en = (uint32_t)t.v[0]; ep = (uint32_t)t.v[1]; eo = (uint32_t)t.v[2]; }
    const bool is_delta = en == DSA_INVALID;
    const bool n_prev = en + 1 == pc, p_prev = ep + 1 == pc;                   // en = INVALID: en + 1 = 0, never pc (pc > 0 whenever para[pc] is a parallelogram)
    // prediction = base + (o[ga] - o[gb]); a delta entry takes o[0] - o[0]
    uint32_t ga = is_delta ? 0u : (n_prev ? ep : en), gb = is_delta ? 0u : eo;
    const bool near_prev = is_delta || n_prev || p_prev;
    // "zig-zag" entries -- a strip both of whose rows are new: prediction = o[p-1] + o[p-2] - o[p-3], every operand inside the run.
    // In first differences e[p] = o[p] - o[p-1] that is e[p] = e[p-2] + corr: a segmented scan over the lanes of one parity
    // (the other lanes, whose differences are known from finished entries, are its segment heads), then the sum as before.
    const bool zig = PARA && live && !is_delta && (n_prev || p_prev) && lane >= 2 && pc >= 3 && ga == pc - 2 && gb == pc - 3;
    bool chain = live && near_prev && (is_delta || (ga < p0 && gb < p0) || zig);
    // lane 0 whose parallelogram is made of three older entries: base = o[next], g = o[prev] - o[opposite]
    const bool far0 = lane == 0 && !near_prev;
    if (far0) { ga = ep; gb = eo; }
    const uint64_t not_chain = __ballot(!chain) & ~1ull;
    uint32_t run = not_chain ? (uint32_t)__builtin_ctzll(not_chain) : WAVE;
    run = run < entries - p0 ? run : entries - p0;
    const V corr = wv[pc], va = wv[ga <= lastp ? ga : lastp], vb = wv[gb <= lastp ? gb : lastp];
    int32_t base[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) base[c] = p0 ? last[c] : 0;
    if (__ballot(far0)) {                          // rare (a handful per mesh): scalar branch
      const V vf = wv[far0 && en <= lastp ? en : 0u];
#pragma unroll
      for (int c = 0; c < NC; ++c) if (far0) base[c] = vf.v[c];
    }
    int32_t g[NC], o[NC];
    bool irregular = false;
    const bool any_zig = PARA && (__ballot(zig && lane < run) != 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      g[c] = zig ? 0 : (int32_t)((uint32_t)va.v[c] - (uint32_t)vb.v[c]);      // (a zig-zag lane loaded entries that are not final: its g follows below)
      // lane 0: the reference's step from finished entries, exact whatever happens to the rest of the run
      const int32_t o0 = wrap_original((int32_t)((uint32_t)base[c] + (uint32_t)g[c]), corr.v[c], mn, mx, max_dif);
      const uint32_t x0 = (uint32_t)o0 - (uint32_t)mn;
      // lanes 1..: residue of g + corr modulo M (both below M in magnitude for real data; anything else ends the run)
      const bool wild = (uint32_t)(g[c] + max_dif - 1) > 2u * M - 2u || (uint32_t)(corr.v[c] + max_dif - 1) > 2u * M - 2u;
      const uint32_t t = (uint32_t)(g[c] + corr.v[c] + 2 * max_dif);      // in (0, 4M) when not wild
      uint32_t r = t - (uint32_t)((float)t * inv_m) * M;
      r = (int32_t)r < 0 ? r + M : r;
      r = r >= M ? r - M : r;
      if (any_zig) {                               // wave-uniform: most steps have no such lane
        // lane 0's difference from the entry before the run; zig-zag lanes add the difference two lanes below to their correction
        uint32_t e0 = x0 + M - (uint32_t)(last[c] - mn);
        e0 = e0 >= M ? e0 - M : e0;
        uint32_t pk = (lane == 0 ? e0 : r) | (zig ? 0u : 0x80000000u);      // value (< M <= 2^25; 32 of them stay below 2^31) | segment head
#pragma unroll
        for (int d = 2; d <= 32; d <<= 1) {
          const uint32_t pv = (uint32_t)__shfl_up((int)pk, d, WAVE);
          if (lane >= (uint32_t)d && !(pk >> 31)) pk = (pk + (pv & 0x7FFFFFFFu)) | (pv & 0x80000000u);
        }
        const uint32_t e = pk & 0x7FFFFFFFu;
        uint32_t er = e - (uint32_t)((float)e * inv_m) * M;
        er = (int32_t)er < 0 ? er + M : er;
        er = er >= M ? er - M : er;
        r = zig ? er : r;
      }
      uint32_t x = lane == 0 ? x0 : r;
      irregular = irregular || (lane == 0 ? x0 >= M : wild);
      o[c] = o0;
      g[c] = g[c];
      // plain sums of up to 64 residues stay below 2^31
      x = lane < run ? x : 0u;
      x = wave_incl_sum(x);
      uint32_t q = x - (uint32_t)((float)x * inv_m) * M;
      q = (int32_t)q < 0 ? q + M : q;
      q = q >= M ? q - M : q;
      if (lane != 0) o[c] = (int32_t)((uint32_t)mn + q);
    }
    {
      const uint64_t irr = __ballot(irregular && lane < run);
      if (irr) run = (irr & 1ull) ? 1u : (uint32_t)__builtin_ctzll(irr);     // lane 0 keeps its exact value, nobody chains on it
    }
    // every lane re-evaluates the sequential step from its neighbour's value (the sums above included lanes that the
    // irregular cut has just dropped: their successors fail here, as they must)
    if (any_zig) {                                 // what a zig-zag lane's parallelogram really is, from the candidates two and three lanes below
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int32_t o2 = __shfl_up(o[c], 2, WAVE);
        int32_t o3 = __shfl_up(o[c], 3, WAVE);
        o3 = lane == 2 ? last[c] : o3;
        if (zig) g[c] = (int32_t)((uint32_t)o2 - (uint32_t)o3);
      }
    }
    bool good = true;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int32_t prev = (int32_t)lane_prev((uint32_t)o[c]);
      const int32_t pred = (int32_t)((uint32_t)prev + (uint32_t)g[c]);
      good = good && pred >= mn && pred <= mx && wrap_original(pred, corr.v[c], mn, mx, max_dif) == o[c];
    }
    const uint64_t bad = __ballot(lane != 0 && lane < run && !good);
    if (bad) run = (uint32_t)__builtin_ctzll(bad);       // >= 1: lane 0 is always exact
    if (lane < run) {
      V r;
#pragma unroll
      for (int c = 0; c < NC; ++c) r.v[c] = o[c];
      ((V *)w)[p] = r;
      if (out) {     // dequantised on the way out (what k_finalize would do from memory): two f32 roundings, Dequantizer.cs:15-23
        PwVec<NC> f;
#pragma unroll
        for (int c = 0; c < NC; ++c) f.v[c] = __float_as_int(__fadd_rn(__fmul_rn((float)o[c], delta), qmin[c]));
        ((PwVec<NC> *)out)[p] = f;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) last[c] = (int32_t)rdlane((uint32_t)o[c], run - 1);
    p0 += run;
  }
}

__device__ __forceinline__ void predict_wrap_attribute(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t phase, uint32_t flags) {
  const AttrDesc &a = D->att[ai];
  if (!wrap_fast_ok(a, flags) || att_is_late(a) != (phase == 1) || pred_filtered(a, flags)) return;
  if (phase == 1 && (flags & LATE_HANDOFF) == 0 && (a.late_ready & 4u)) return;        // (the kernel's launch: predicted by one of its producers already)
  int32_t *w = (int32_t *)(arena + L.work[ai]);
  const uint32_t *para = att_para(arena, L, D, a);
  const uint32_t e = a.num_entries, nc = a.nc_portable;
  const int32_t mn = a.wrap_min, mx = a.wrap_max;
  // quantised floats leave this kernel dequantised (k_finalize skips them)
  const bool fused = pw_dequant_fused(a, flags);
  float *out = fused ? (float *)(arena + L.out[ai]) : nullptr;
  const float delta = fused ? __fdiv_rn(a.q_range, (float)(int32_t)((1u << a.q_bits) - 1u)) : 0.0f;
  float qmin[4];
  for (int c = 0; c < 4; ++c) qmin[c] = fused && (uint32_t)c < nc ? a.q_min[c] : 0.0f;
  if (a.pred_kind == 1) {
    if (nc == 1) predict_wrap_wave<1, true>(w, para, e, mn, mx, out, delta, qmin);
    else if (nc == 2) predict_wrap_wave<2, true>(w, para, e, mn, mx, out, delta, qmin);
    else if (nc == 3) predict_wrap_wave<3, true>(w, para, e, mn, mx, out, delta, qmin);
    else predict_wrap_wave<4, true>(w, para, e, mn, mx, out, delta, qmin);
  } else {
    if (nc == 1) predict_wrap_wave<1, false>(w, para, e, mn, mx, out, delta, qmin);
    else if (nc == 2) predict_wrap_wave<2, false>(w, para, e, mn, mx, out, delta, qmin);
    else if (nc == 3) predict_wrap_wave<3, false>(w, para, e, mn, mx, out, delta, qmin);
    else predict_wrap_wave<4, false>(w, para, e, mn, mx, out, delta, qmin);
  }
}
__global__ __launch_bounds__(WAVE) void k_predict_wrap(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t phase, uint32_t flags) {
  const uint32_t mesh = blockIdx.x, ai = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || ai >= D->num_attributes) return;
  if (phase == 0 && D->att[ai].early_done) return;
  predict_wrap_attribute(arena, layouts[mesh], D, ai, phase, flags);
}

// =========================================================================
// k_finalize: portable ints -> attribute values, point->entry maps.
// =========================================================================
// phase 0 (symbol stream): attributes whose values are complete once the symbols and a traversal-free scheme are
// done; phase 1 (main stream, last): parallelogram attributes and everything of the general-path meshes.
// One attribute's share of k_finalize for the threads (tid, tid + stride, ...): a 256-thread grid slice, or one wave.
__device__ __forceinline__ void finalize_attribute(uint8_t *arena, const MeshLayout &L, const MeshDesc *D, uint32_t ai, uint32_t phase, uint32_t flags, uint32_t tid, uint32_t stride) {
  const AttrDesc &a = D->att[ai];
  const bool late = D->general || att_is_late(a);
  if (late != (phase == 1)) return;
  if (!D->general && pw_dequant_fused(a, flags)) return;      // k_predict_wrap wrote the floats
  const int32_t *w = (const int32_t *)(arena + L.work[ai]);
  const uint32_t entries = a.num_entries;
  if (a.seq_type == 2) {            // AttributeQuantizationTransform.cs:179-199, Dequantizer.cs:15-23: two f32 roundings
    float *out = (float *)(arena + L.out[ai]);
    const uint32_t nc = a.nc;
    const float delta = __fdiv_rn(a.q_range, (float)(int32_t)((1u << a.q_bits) - 1u));
    for (uint32_t i = tid; i < entries * nc; i += stride) {
      float prod = __fmul_rn((float)w[i], delta);
      out[i] = __fadd_rn(prod, a.q_min[i % nc]);
    }
  } else if (a.seq_type == 3) {     // OctahedronToolBox.cs:139-142,220-239 (D-8 corrected)
    float *out = (float *)(arena + L.out[ai]);
    const int32_t max_value = (1 << a.q_bits) - 2;
    const float scale = __fdiv_rn(2.0f, (float)max_value);
    for (uint32_t e = tid; e < entries; e += stride) {
      float y = __fsub_rn(__fmul_rn((float)w[2 * e], scale), 1.0f);
      float z = __fsub_rn(__fmul_rn((float)w[2 * e + 1], scale), 1.0f);
      float x = __fsub_rn(__fsub_rn(1.0f, fabsf(y)), fabsf(z));
      float x_off = -x < 0.0f ? 0.0f : -x;
      y = __fadd_rn(y, y < 0.0f ? x_off : -x_off);
      z = __fadd_rn(z, z < 0.0f ? x_off : -x_off);
      float norm2 = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
      float ox = 0.0f, oy = 0.0f, oz = 0.0f;
      if ((double)norm2 >= 1e-6) {
        double d = __ddiv_rn(1.0, __dsqrt_rn((double)norm2));
        ox = (float)__dmul_rn((double)x, d); oy = (float)__dmul_rn((double)y, d); oz = (float)__dmul_rn((double)z, d);
      }
      out[3 * e] = ox; out[3 * e + 1] = oy; out[3 * e + 2] = oz;
    }
  } else if (a.seq_type == 1) {     // SequentialIntegerAttributeDecoder.cs:142-160: narrowing store
    const uint32_t width = data_type_length(a.data_type), total = entries * a.nc;
    uint8_t *out = arena + L.out[ai];
    for (uint32_t i = tid; i < total; i += stride) {
      int32_t v = w[i];
      if (width == 1) out[i] = (uint8_t)v;
      else if (width == 2) ((uint16_t *)out)[i] = (uint16_t)v;
      else ((uint32_t *)out)[i] = (uint32_t)v;
    }
  } else {                          // generic: bytes as stored
    const uint8_t *src = arena + L.stream + a.off_raw;
    uint8_t *out = arena + L.out[ai];
    const uint32_t total = entries * a.nc * data_type_length(a.data_type);
    for (uint32_t i = tid; i < total; i += stride) out[i] = src[i];
  }
}
__global__ __launch_bounds__(256) void k_finalize(uint8_t *arena, const MeshLayout *layouts, const MeshDesc *descs, uint32_t n, uint32_t phase, uint32_t flags) {
  const uint32_t mesh = blockIdx.y, ai = blockIdx.z;
  if (mesh >= n) return;
  const MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || ai >= D->num_attributes) return;
  if (phase == 0 && D->att[ai].early_done) return;
  finalize_attribute(arena, layouts[mesh], D, ai, phase, flags, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// =========================================================================
// GeometricNormal prediction on the fast kernels (MeshPredictionSchemeGeometricNormalDecoder.cs:44-82, what stock encoders pick
// for normals at their default level).  The prediction of an entry is the area-weighted sum of the face normals around its
// vertex, from the decoded quantised positions: nothing of it depends on other normals, so the entries of a mesh are
// independent -- but for the flip bit each carries, a serial rABS stream.
//   k_flip_bits         one lane per (mesh, attribute), from the start of the decode (it needs k_locate's offsets only): the bits,
//                       packed 32 to a word, into the mesh's vertex-stamp region (4 bytes per vertex that only the general path
//                       uses: room for the bits of 32 attributes; the attribute's own output region holds the symbol
//                       kernels' tables until the values are written)
//   k_predict_geometric one thread per entry, behind the traversal and the prediction of the positions: the fan of corners
//                       around the vertex on the face records (VertexCornersIterator.cs: left from the corner the entry was
//                       reached through, then right from it; D-10), the sums in the bitstream's 64-bit arithmetic, then
//                       geometric_normal_finish (dsa_common.h) with the entry's correction, in place.
// =========================================================================
__device__ __forceinline__ uint32_t *flip_bits_of(uint8_t *arena, const MeshLayout &L, uint32_t ai) {
  return (uint32_t *)(arena + L.vstamp) + (size_t)ai * ((L.cap_vertices + 31u) / 32u);
}
// The bit array of an attribute's serial side stream (flip bits of GeometricNormal, orientation bits of TexCoordsPortable) and its
// capacity in bits: the vertex-stamp slot of a vertex attribute, the `orient` region of its attribute data block for a corner
// attribute (whose entries are not bounded by the vertex count; k_locate sees to it that one attribute per block uses it).
__device__ __forceinline__ uint32_t *orient_bits_of(uint8_t *arena, const MeshLayout &L, const MeshDesc *D, uint32_t ai, uint32_t *capacity) {
  const AttrDesc &a = D->att[ai];
  if (a.corner_data == 0) { *capacity = L.cap_vertices; return flip_bits_of(arena, L, ai); }
  const SeamLayout g = seam_layout(L.cap_faces, L.cap_vertices, D->num_att_data, L.rec_compact != 0);
  *capacity = 3u * L.cap_faces;
  return (uint32_t *)(seam_block(arena, L, g, (uint32_t)a.corner_data - 1u) + g.orient);
}
// corner_pass: 0 the attributes located at the start of the decode, 1 those of corner-attribute decoders (behind k_seam_tables,
// which counts their entries = their flip bits) and what the walk located only then
__global__ __launch_bounds__(WAVE) void k_flip_bits(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t lanes_per_mesh, uint32_t corner_pass) {
  const uint32_t lane = lane_id();
  const uint32_t mesh = blockIdx.x * (WAVE / lanes_per_mesh) + lane / lanes_per_mesh, ai = lane % lanes_per_mesh;
  if (mesh >= n) return;
  const MeshLayout &L = layouts[mesh];
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || ai >= D->num_attributes) return;
  const AttrDesc &a = D->att[ai];
  if (!a.have_scheme || a.pred_kind != 2 || a.source == SRC_BYTES || att_behind_tables(a) != (corner_pass != 0)) return;
  Rabs rb;
  uint32_t endp;
  rb.start(arena + L.stream, L.stream_len, a.off_flips, &endp);
  if (!rb.ok) { fail(D, ST_INVALID, 655); return; }
  uint32_t cap;
  uint32_t *bits = orient_bits_of(arena, L, D, ai, &cap);
  const uint32_t entries = a.num_entries;
  if (entries > cap) { fail(D, ST_INVALID, 657); return; }
  (void)rabs_block_to_words<false>(rb, entries, bits);
}

// NARROW: the positions are quantised (below 2^30), so the edge vectors fit 32 bits and a term of the cross product is one
// 32 x 32 -> 64 multiplication (the same value as the bitstream's 64-bit product)
template <bool CP, bool NARROW>
__device__ __forceinline__ void geometric_entries(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t tid, uint32_t stride) {
  typedef typename std::conditional<NARROW, int32_t, int64_t>::type coord;
  typedef Rec<CP> R;
  const AttrDesc &a = D->att[ai];
  const uint32_t *frec = (const uint32_t *)(arena + L.frec);
  // a corner attribute (normals with seams): the fan around the entry's corner ends at the attribute's seams -- the opposites of the
  // attribute's own records --, the positions are those of the position vertices at the same corners
  const bool corner_att = a.corner_data != 0;
  const TravIO aio = corner_att ? trav_attribute(arena, L, D, (uint32_t)a.corner_data - 1u) : trav_position(arena, L, D);
  const uint32_t *arec = aio.frec;
  const uint32_t *d2c = aio.d2c;
  const int32_t *posv = (const int32_t *)(arena + L.para);       // k_vertex_positions
  uint32_t bits_cap;
  const uint32_t *bits = orient_bits_of(arena, L, D, ai, &bits_cap);
  auto load_rec = [&](uint32_t f) -> typename R::Raw {
    typename R::Raw r = R::load(frec, f);
    if (corner_att) { const typename R::Raw q = R::load(arec, f); r.o = q.o; }
    return r;
  };
  int32_t *w = (int32_t *)(arena + L.work[ai]);
  const uint32_t entries = a.num_entries, NV = D->num_vertices, F = D->num_faces;
  OctParams o;
  const int q = 32 - __clz(a.oct_max_q);
  o.center = ((1 << q) - 2) / 2;
  o.max_q = (1 << q) - 1;
  const bool canonical = a.pred_transform == 3;
  for (uint32_t p = tid; p < entries; p += stride) {
    const uint32_t ci = d2c[p];
    bool ok = (ci >> 2) < F && (ci & 3u) != 3u;
    auto position = [&](uint32_t v, coord dst[3]) {
      if (v >= NV) { ok = false; dst[0] = dst[1] = dst[2] = 0; return; }
      for (int k = 0; k < 3; ++k) dst[k] = posv[(size_t)v * 3 + k];
    };
    uint64_t nsum[3] = {0, 0, 0};
    if (ok) {
      // One record load per face of the fan (its vertices and its opposites), one new position per face: two faces in a row
      // share an edge, so going left the vertex behind the corner becomes the one ahead of it, going right the other way round.
      typename R::Raw rec = load_rec(ci >> 2);
      const typename R::Raw first = rec;
      coord center[3], pn[3], pp[3], pn0[3];
      position(R::vertex(rec, ci & 3u), center);
      position(R::vertex(rec, k_next(ci & 3u)), pn);
      position(R::vertex(rec, k_prev(ci & 3u)), pp);
      for (int k = 0; k < 3; ++k) pn0[k] = pn[k];
      uint32_t c = ci, steps = 0;
      bool left = true;
      while (ok) {
        // Where the fan goes next follows from the record alone: the next face's record is requested BEFORE this face's cross
        // product waits for its position -- one memory round trip per face in the chain of an entry instead of two.
        uint32_t nc = DSA_INVALID;
        bool nleft = left, done = false, restart = false;
        if (left) {
          const uint32_t ol = R::opp(rec, k_next(c & 3u));                    // SwingLeft: Next(Opposite(Next(c)))
          if (ol != DSA_INVALID) { nc = qnext(ol); done = nc == ci; }         // (all the way round)
          else { nleft = false; restart = true; }                             // a boundary: the rest of the fan to the right of the start corner
        }
        if (!nleft && !done) {
          const uint32_t orr = restart ? R::opp(first, k_prev(ci & 3u)) : R::opp(rec, k_prev(c & 3u));     // SwingRight: Previous(Opposite(Previous(c)))
          if (orr == DSA_INVALID) done = true; else nc = qprev(orr);
        }
        const bool nvalid = !done && (nc >> 2) < F && (nc & 3u) != 3u;
        const typename R::Raw nrec = load_rec(nvalid ? nc >> 2 : ci >> 2);
        coord u[3], v[3];
        for (int k = 0; k < 3; ++k) { u[k] = pn[k] - center[k]; v[k] = pp[k] - center[k]; }
        nsum[0] += (uint64_t)((int64_t)u[1] * v[2]) - (uint64_t)((int64_t)u[2] * v[1]);
        nsum[1] += (uint64_t)((int64_t)u[2] * v[0]) - (uint64_t)((int64_t)u[0] * v[2]);
        nsum[2] += (uint64_t)((int64_t)u[0] * v[1]) - (uint64_t)((int64_t)u[1] * v[0]);
        if (++steps > 3u * F + 1u) { ok = false; break; }
        if (done) break;
        if (!nvalid) { ok = false; break; }
        if (nleft) {
          for (int k = 0; k < 3; ++k) pn[k] = pp[k];                           // the shared edge's far vertex is now ahead of the corner
          position(R::vertex(nrec, k_prev(nc & 3u)), pp);
        } else {
          if (restart) for (int k = 0; k < 3; ++k) pn[k] = pn0[k];
          for (int k = 0; k < 3; ++k) pp[k] = pn[k];                           // the shared edge's far vertex is now behind the corner
          position(R::vertex(nrec, k_next(nc & 3u)), pn);
        }
        c = nc; rec = nrec; left = nleft;
      }
    }
    if (!ok) { fail(D, ST_INVALID, 650); continue; }
    const bool flip = (bits[p >> 5] >> (p & 31u)) & 1u;
    int32_t os, ot;
    geometric_normal_finish(o, canonical, nsum, flip, w[2 * p], w[2 * p + 1], os, ot);
    w[2 * p] = os; w[2 * p + 1] = ot;
  }
}
__global__ __launch_bounds__(256) void k_predict_geometric(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  const uint32_t mesh = blockIdx.y, ai = blockIdx.z;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || ai >= D->num_attributes) return;
  const AttrDesc &a = D->att[ai];
  if (!a.have_scheme || a.pred_kind != 2 || a.source == SRC_BYTES || a.num_entries == 0) return;
  const MeshLayout &L = layouts[mesh];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  const bool narrow = D->geo_wide == 0;      // every position below 2^30 in magnitude (any quantised attribute of a sound stream)
  if (L.rec_compact) { if (narrow) geometric_entries<true, true>(arena, L, D, ai, tid, stride); else geometric_entries<true, false>(arena, L, D, ai, tid, stride); }
  else { if (narrow) geometric_entries<false, true>(arena, L, D, ai, tid, stride); else geometric_entries<false, false>(arena, L, D, ai, tid, stride); }
}
// The portable positions by vertex id (vertex -> entry -> position: the attributes of a seam-free mesh share their sequence), for
// meshes with a GeometricNormal attribute, into the parallelogram-operand region, which the wrap prediction in front of this
// kernel was the last to read: k_predict_geometric looks up a position per face of every fan, this saves it a hop each time.
__global__ __launch_bounds__(256) void k_vertex_positions(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  const uint32_t mesh = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general) return;
  bool any = false;
  for (uint32_t k = 0; k < D->num_attributes; ++k) any = any || (D->att[k].have_scheme && D->att[k].pred_kind == 2 && D->att[k].source != SRC_BYTES);
  if (!any) return;
  uint32_t pa = DSA_INVALID;
  for (uint32_t k = 0; k < D->num_attributes; ++k) if (D->att[k].att_type == 0 && D->att[k].seq_type != 0) { pa = k; break; }
  if (pa == DSA_INVALID || D->att[pa].nc_portable != 3) { if (threadIdx.x == 0 && blockIdx.x == 0) fail(D, ST_INVALID, 656); return; }
  const MeshLayout &L = layouts[mesh];
  const int32_t *v2d = (const int32_t *)(arena + L.v2d);
  const int32_t *pos = (const int32_t *)(arena + L.work[pa]);
  int32_t *posv = (int32_t *)(arena + L.para);
  const uint32_t NV = D->num_vertices < L.cap_vertices ? D->num_vertices : L.cap_vertices, pos_entries = D->att[pa].num_entries;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < NV; v += gridDim.x * blockDim.x) {
    const int32_t d = v2d[v];
    const bool there = d >= 0 && (uint32_t)d < pos_entries;      // a vertex of a face always is (the traversal visits every face)
    bool big = false;
    for (int k = 0; k < 3; ++k) {
      const int32_t x = there ? pos[(size_t)d * 3 + k] : 0;
      big = big || x <= -(1 << 30) || x >= (1 << 30);
      posv[(size_t)v * 3 + k] = x;
    }
    if (big) D->geo_wide = 1;      // (every writer writes the same value)
  }
}

// k_faces: faces as point ids (Mesh.cs:15-69; MeshEdgeBreakerDecoder.cs:537-553,627-637) and the census of linked corners,
// behind k_chain on the third stream, beside the parallelogram prediction.  (As a pass of the connectivity wave itself it
// cost that wave 4 ms: 4096 waves reach it together and the 13 GB it moves are then on the critical path.)
__global__ __launch_bounds__(256) void k_faces(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  uint32_t mesh = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || D->encoder_type == 0) return;   // k_general writes the faces of its meshes
  const MeshLayout &L = layouts[mesh];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  const uint32_t *frec = (const uint32_t *)(arena + L.frec);
  const uint32_t *vrank = (const uint32_t *)(arena + L.vrank);
  int32_t *faces = (int32_t *)(arena + L.faces);
  const uint32_t NV = D->num_vertices;
  const bool compact = L.rec_compact != 0, seam_fast = D->seam_fast != 0;
  // Every link k_connectivity makes sets two corners; a corner linked twice ("corner already has an
  // opposite", MeshEdgeBreakerDecoder.cs:254,272,314,392) leaves fewer linked corners than 2 x links.
  uint32_t linked = 0, bad = 0;
  for (uint32_t f = tid; f < D->num_faces; f += stride) {
    uint4 vv, oo;
    if (compact) { const Rec<true>::Raw r = Rec<true>::load(frec, f); vv = make_uint4(Rec<true>::vertex(r, 0), Rec<true>::vertex(r, 1), Rec<true>::vertex(r, 2), 0u); oo = make_uint4(Rec<true>::opp(r, 0), Rec<true>::opp(r, 1), Rec<true>::opp(r, 2), 0u); }
    else { const Rec<false>::Raw r = Rec<false>::load(frec, f); vv = r.v; oo = r.o; }
    linked += (oo.x != DSA_INVALID) + (oo.y != DSA_INVALID) + (oo.z != DSA_INVALID);
    if (seam_fast) continue;               // points are not vertices there: k_seam_tables numbers them per corner
    if (vv.x < NV && vv.y < NV && vv.z < NV) {
      faces[3 * f] = (int32_t)vrank[vv.x]; faces[3 * f + 1] = (int32_t)vrank[vv.y]; faces[3 * f + 2] = (int32_t)vrank[vv.z];
    } else bad = 1;
  }
  for (int d = 32; d >= 1; d >>= 1) linked += __shfl_xor(linked, d, 64);
  if (lane_id() == 0 && linked) atomicAdd(&D->linked_corners, linked);
  if (__ballot(bad) && lane_id() == 0) fail(D, ST_INVALID, 263);
}

// k_point_maps: the point -> entry map of point clouds (linear sequencer); meshes get theirs from the traversal wave.
__global__ __launch_bounds__(256) void k_point_maps(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  uint32_t mesh = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general) return;   // k_general writes the maps of its meshes
  const MeshLayout &L = layouts[mesh];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  const uint32_t na = D->num_attributes;
  if (D->encoder_type != 0) return;
  for (uint32_t p = tid; p < D->num_points; p += stride)
    for (uint32_t ai = 0; ai < na; ++ai) ((uint32_t *)(arena + L.map[ai]))[p] = p;
}

// k_seal: last kernel of a decode, one thread per mesh: the census of linked corners against the links made, and the seam
// bits found by k_conn_checks against the number of interior edges.
__global__ __launch_bounds__(256) void k_seal(MeshDesc *descs, uint32_t n) {
  const uint32_t mesh = blockIdx.x * blockDim.x + threadIdx.x;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK) return;
  if (D->general) return;                              // the general path's phase 2 has compared its own census
  if (D->values_pending) { fail(D, ST_INVALID, 159); return; }          // (the walk of the attribute sections did not get to its end)
  if (__hip_atomic_load(&D->linked_corners, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != D->interior_corners) { fail(D, ST_INVALID, 263); return; }
  // a seam among the coded bits (k_conn_checks) of a mesh without seam scratch: the general path builds its tables.  (A mesh with
  // corner-attribute decoders has the scratch, and k_seam_tables has used the bits of every attribute data -- those of a
  // vertex-attribute decoder too, which split points as well: MeshEdgeBreakerDecoder.cs:537-638.)
  if (D->encoder_type != 0 && !D->seam_fast)
    for (uint32_t d = 0; d < D->num_att_data; ++d)
      if (D->seam_first[d] < D->interior_corners / 2) { fail(D, ST_NOTIMPL, DSA_SITE_RETRY_GENERAL); return; }
}

}  // namespace dsa
