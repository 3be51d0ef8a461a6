// draco-sharp_amd/csrc/dsa_general.h
//
// k_general: the whole decode of one Edgebreaker mesh by a single lane, for the streams the fast kernels
// of dsa_kernels.h do not take:
//   * valence traversal            Mesh/MeshEdgeBreakerTraversalValenceDecoder.cs:22-154
//   * attribute seams, attribute corner tables, corner attributes
//                                  Mesh/MeshEdgeBreakerDecoder.cs:502-638, Mesh/MeshAttributeCornerTable.cs:19-215
//   * TexCoordsPortable prediction PredictionSchemes/MeshPredictionSchemeTexCoordsPortable{Decoder,Predictor}.cs
// It is the sequential algorithm of the reference, statement by statement, on arrays in the batch arena: one
// wave per mesh with lane 0 active, meshes of a batch in parallel.  Correctness first: this is the parity path for
// real-world files such as the reference's house_04 sample; the throughput path is dsa_kernels.h.
// Everything after the header is parsed here (the host only sized the arena); results land where the fast path
// puts them (faces, work[] = portable values, map[], AttrDesc), so k_finalize and the C-ABI do not care which
// path produced them.
#pragma once
#include "dsa_common.h"

namespace dsa {
namespace gen {

#define GFAIL(site) do { fail(D, ST_INVALID, (site)); return false; } while (0)
#define GREQ(cond, site) do { if (!(cond)) GFAIL(site); } while (0)
#define GNOTIMPL(site) do { fail(D, ST_NOTIMPL, (site)); return false; } while (0)
#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t gclk() { return __builtin_amdgcn_s_memtime(); }
#else
inline uint64_t gclk() { return 0; }
#endif

__device__ __forceinline__ uint32_t cnx(uint32_t c) { return c == DSA_INVALID ? c : ((c % 3u == 2u) ? c - 2u : c + 1u); }
__device__ __forceinline__ uint32_t cpv(uint32_t c) { return c == DSA_INVALID ? c : ((c % 3u == 0u) ? c + 2u : c - 1u); }

// Mesh/CornerTable.cs:59-82,127-130,174-192,213-245,275-279
struct Ct {
  uint32_t *opp, *c2v, *vcorner;
  uint32_t F, C, nv, vmax;
  __device__ __forceinline__ uint32_t opposite(uint32_t c) const { return c < C ? opp[c] : DSA_INVALID; }
  __device__ __forceinline__ uint32_t vertex(uint32_t c) const { return c < C ? c2v[c] : DSA_INVALID; }
  __device__ __forceinline__ uint32_t left_most(uint32_t v) const { return v < nv ? vcorner[v] : DSA_INVALID; }
  __device__ __forceinline__ uint32_t swing_right(uint32_t c) const { return cpv(opposite(cpv(c))); }
  __device__ __forceinline__ uint32_t swing_left(uint32_t c) const { return cnx(opposite(cnx(c))); }
  __device__ __forceinline__ uint32_t right_corner(uint32_t c) const { return opposite(cnx(c)); }
  __device__ __forceinline__ uint32_t left_corner(uint32_t c) const { return opposite(cpv(c)); }
  __device__ __forceinline__ bool is_on_boundary(uint32_t v) const { return swing_left(left_most(v)) == DSA_INVALID; }
  __device__ __forceinline__ uint32_t num_faces() const { return F; }
  __device__ __forceinline__ void set_opp(uint32_t a, uint32_t b) { if (a < C) opp[a] = b; if (b < C) opp[b] = a; }
};

// Mesh/MeshAttributeCornerTable.cs:19-30 (ctor), :80-93 (AddSeamEdge), :95-155 (RecomputeVertices), :157-215
struct Act {
  const Ct *ct;
  uint8_t *edge_seam, *vert_seam;
  uint32_t *c2v, *v2lm;
  uint32_t nv;
  __device__ __forceinline__ uint32_t opposite(uint32_t c) const { return (c >= ct->C || edge_seam[c]) ? DSA_INVALID : ct->opp[c]; }
  __device__ __forceinline__ uint32_t vertex(uint32_t c) const { return c < ct->C ? c2v[c] : DSA_INVALID; }
  __device__ __forceinline__ uint32_t left_most(uint32_t v) const { return v < nv ? v2lm[v] : DSA_INVALID; }
  __device__ __forceinline__ uint32_t swing_right(uint32_t c) const { return cpv(opposite(cpv(c))); }
  __device__ __forceinline__ uint32_t swing_left(uint32_t c) const { return cnx(opposite(cnx(c))); }
  __device__ __forceinline__ uint32_t right_corner(uint32_t c) const { return opposite(cnx(c)); }
  __device__ __forceinline__ uint32_t left_corner(uint32_t c) const { return opposite(cpv(c)); }
  __device__ __forceinline__ bool is_on_boundary(uint32_t v) const { uint32_t c = left_most(v); return c == DSA_INVALID || swing_left(c) == DSA_INVALID; }
  __device__ __forceinline__ uint32_t num_faces() const { return ct->F; }
  __device__ __forceinline__ void add_seam_edge(uint32_t c) {
    edge_seam[c] = 1;
    uint32_t a = ct->vertex(cnx(c)), b = ct->vertex(cpv(c));
    if (a < ct->vmax) vert_seam[a] = 1;
    if (b < ct->vmax) vert_seam[b] = 1;
    uint32_t o = ct->opposite(c);
    if (o != DSA_INVALID && o < ct->C) {
      edge_seam[o] = 1;
      a = ct->vertex(cnx(o)); b = ct->vertex(cpv(o));
      if (a < ct->vmax) vert_seam[a] = 1;
      if (b < ct->vmax) vert_seam[b] = 1;
    }
  }
};

// Entropy/RAnsSymbolDecoder.cs:12-59 + RAnsDecoder.cs:20-99, serial; cum[0..ns] cumulative frequencies.
// Scratch for the symbol coder of a general mesh: tables of the common case (12-bit precision, <= 2048 symbols)
// live in LDS (host check: plain memory), where a lookup costs ~100 cycles instead of a dependent chain of global
// loads; larger tables stay in the arena region `cum` and are searched.
#define GEN_LUT_SLOTS 4096    // full resolution: one slot per value of the 12-bit remainder; crowded launches use half (lut_shift 1)
#define GEN_LUT_SYMS 2048    // alphabets up to 2^11 symbols at 12 precision bits keep their tables in LDS (u16 in crowded launches: 8 KB with the half-resolution LUT)
struct RansScratch {
  uint16_t *lut;          // [GEN_LUT_SLOTS >> lut_shift] slot -> symbol holding the slot's first value
  uint32_t lut_shift;     // 0: exact; 1: one slot per two values (4 KB instead of 8), read() steps on when needed
  uint16_t *fast_cum;     // [GEN_LUT_SYMS + 1], 12-bit precision only (lut_shift 1)
  uint32_t *fast_cum32;   // the same as 32-bit words (lut_shift 0: two independent reads per symbol, no stepping)
  uint32_t *cum;          // arena, cum_cap entries
  uint64_t cum_cap;
};
struct Rans {
  uint32_t pb, l_base, ns, state, off;
  const uint8_t *buf;
  const uint32_t *cum;
  const uint16_t *cum16 = nullptr;
  uint32_t lut_shift = 0;
  const uint16_t *lut;    // nullptr: binary search over cum
  const uint16_t *bucket_lut = nullptr;   // bucket of the remainder -> symbol holding the bucket's first value
  uint32_t bucket_shift = 0, num_buckets = 0;
  uint64_t win = 0;       // the eight stream bytes below `off + win_n`, so that a renormalisation byte costs a memory
  uint32_t win_n = 0;     // round trip only once in eight
  __device__ __forceinline__ uint32_t read() {
    while (state < l_base && off > 0) {
      if (win_n == 0) {
        if (off < 8) { state = state * 256u + buf[--off]; continue; }
        __builtin_memcpy(&win, buf + off - 8, 8);
        win_n = 8;
      }
      --off; --win_n;
      state = state * 256u + (uint32_t)((win >> (8 * win_n)) & 0xFFu);
    }
    const uint32_t rem = state & ((1u << pb) - 1u);
    if (lut && lut_shift == 0) {       // 12-bit precision, exact LUT and table in LDS
      const uint32_t s = lut[rem], c0 = cum[s];
      state = (state >> 12) * (cum[s + 1] - c0) + rem - c0;
      return s;
    }
    if (lut) {                         // 12-bit precision, half-resolution LUT (crowded launches: half the LDS)
      uint32_t s = lut[rem >> 1], c1 = cum16[s + 1];
      if (c1 <= rem) { ++s; c1 = cum16[s + 1]; while (c1 <= rem) { ++s; c1 = cum16[s + 1]; } }   // odd value in the next symbol (zero-frequency symbols are stepped over)
      const uint32_t c0 = cum16[s];
      state = (state >> 12) * (c1 - c0) + rem - c0;
      return s;
    }
    uint32_t lo = 0, hi = ns;          // largest s with cum[s] <= rem
    if (bucket_lut) {                  // other precisions: the LDS table narrows the search to the symbols that meet the
      const uint32_t b = rem >> bucket_shift;                       // remainder's bucket (a few, where the full search takes 11-20 round trips)
      lo = bucket_lut[b];
      hi = (b + 1 < num_buckets ? bucket_lut[b + 1] : ns - 1) + 1;
    }
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= rem) lo = mid; else hi = mid; }
    const uint32_t c0 = cum[lo];
    state = (state >> pb) * (cum[lo + 1] - c0) + rem - c0;
    return lo;
  }
};

__device__ __forceinline__ bool rans_create(MeshDesc *D, Rd &r, uint32_t max_bit_length, const RansScratch &rs, Rans &x) {
  uint32_t *cum = rs.cum;
  const uint64_t cum_cap = rs.cum_cap;
  x.pb = rans_precision_bits(max_bit_length);
  x.l_base = 4u << x.pb;
  const uint64_t ns = r.varint();
  GREQ(r.ok && ns >= 1 && ns <= (1u << 20) && ns + 2 <= cum_cap, 600);
  x.ns = (uint32_t)ns;
  GREQ(read_prob_table(r, x.ns, cum + 1), 601);
  cum[0] = 0;
  for (uint32_t i = 0; i < x.ns; ++i) {
    GREQ(cum[i + 1] <= (1u << x.pb), 602);
    cum[i + 1] += cum[i];
    GREQ(cum[i + 1] <= (1u << x.pb), 602);
  }
  GREQ(cum[x.ns] == (1u << x.pb), 603);
  x.cum = cum;
  x.lut = nullptr;
  if (x.pb == 12 && x.ns <= GEN_LUT_SYMS && rs.lut) {
    if (rs.lut_shift == 0) { for (uint32_t i = 0; i <= x.ns; ++i) rs.fast_cum32[i] = cum[i]; x.cum = rs.fast_cum32; }
    else for (uint32_t i = 0; i <= x.ns; ++i) rs.fast_cum[i] = (uint16_t)cum[i];
    const uint32_t sh = rs.lut_shift, half = (1u << sh) - 1u;
    for (uint32_t i = 0; i < x.ns; ++i) for (uint32_t j = (cum[i] + half) >> sh; (j << sh) < cum[i + 1]; ++j) rs.lut[j] = (uint16_t)i;   // slots whose first value lies in [cum[i], cum[i+1])
    x.lut_shift = sh;
    x.cum16 = rs.fast_cum; x.lut = rs.lut;
  } else if (rs.lut && x.ns <= 65536) {
    const uint32_t nb = GEN_LUT_SLOTS >> rs.lut_shift, lg = rs.lut_shift ? 11u : 12u;
    x.bucket_shift = x.pb - lg; x.num_buckets = nb;
    uint32_t sym = 0;
    for (uint32_t j = 0; j < nb; ++j) {
      const uint32_t start = j << x.bucket_shift;
      while (cum[sym + 1] <= start) ++sym;
      rs.lut[j] = (uint16_t)sym;
    }
    x.bucket_lut = rs.lut;
  }
  // RAnsSymbolDecoder.cs:53-59, RAnsDecoder.cs:20-54
  const uint64_t size = r.varint();
  GREQ(r.ok && size >= 1 && size <= (uint64_t)(r.n - r.pos), 604);
  x.buf = r.p + r.pos;
  r.skip(size);
  const uint32_t o = (uint32_t)size, t = x.buf[o - 1] >> 6;
  if (t == 0) { x.off = o - 1; x.state = x.buf[o - 1] & 0x3Fu; }
  else if (t == 1) { GREQ(o >= 2, 605); x.off = o - 2; x.state = ((uint32_t)x.buf[o - 2] | ((uint32_t)x.buf[o - 1] << 8)) & 0x3FFFu; }
  else if (t == 2) { GREQ(o >= 3, 605); x.off = o - 3; x.state = ((uint32_t)x.buf[o - 3] | ((uint32_t)x.buf[o - 2] << 8) | ((uint32_t)x.buf[o - 1] << 16)) & 0x3FFFFFu; }
  else { GREQ(o >= 4, 605); x.off = o - 4; x.state = ((uint32_t)x.buf[o - 4] | ((uint32_t)x.buf[o - 3] << 8) | ((uint32_t)x.buf[o - 2] << 16) | ((uint32_t)x.buf[o - 1] << 24)) & 0x3FFFFFFFu; }
  x.state += x.l_base;
  GREQ(x.state < x.l_base * 256u, 606);
  return true;
}

// Entropy/SymbolDecoding.cs:7-67 (tagged path per the bitstream, D-1)
// zigzag: store ConvertSymbolToSignedInt(symbol) instead of the symbol (SequentialIntegerAttributeDecoder.cs:86-99),
// saving the separate pass over the values.
__device__ __forceinline__ uint32_t unzigzag(uint32_t sv) { return (sv & 1u) ? (uint32_t)(-(int32_t)(sv >> 1) - 1) : (sv >> 1); }
__device__ __forceinline__ bool decode_symbols(MeshDesc *D, Rd &r, uint32_t num_values, uint32_t nc, uint32_t *out, const RansScratch &rs, bool zigzag = false) {
  if (num_values == 0) return true;
  const uint32_t scheme = r.u8();
  GREQ(r.ok && scheme <= 1, 610);
  Rans x;
  if (scheme == 0) {
    if (!rans_create(D, r, 5, rs, x)) return false;
    const uint8_t *bits = r.p + r.pos;
    const uint32_t nbytes = r.n - r.pos;
    uint64_t bitpos = 0;
    uint32_t vid = 0;
    for (uint32_t i = 0; i < num_values; i += nc) {
      const uint32_t len = x.read();
      GREQ(len <= 32, 611);
      for (uint32_t j = 0; j < nc; ++j) {
        GREQ(vid < num_values, 612);
        const uint32_t sv = len ? read_bits(bits, nbytes, bitpos, len) : 0u;
        out[vid++] = zigzag ? unzigzag(sv) : sv;
        bitpos += len;
      }
    }
    r.skip((bitpos + 7) >> 3);
    GREQ(r.ok, 613);
  } else {
    const uint32_t mbl = r.u8();
    GREQ(r.ok && mbl >= 1 && mbl <= 18, 614);
    if (!rans_create(D, r, mbl, rs, x)) return false;
    if (zigzag) for (uint32_t i = 0; i < num_values; ++i) out[i] = unzigzag(x.read());
    else for (uint32_t i = 0; i < num_values; ++i) out[i] = x.read();
  }
  return true;
}

// Traverser/DepthFirstTraverser.cs:9-99 + MeshAttributeIndicesEncodingObserver.cs:14-21 +
// MeshTraversalSequencer.cs:13-31 on either corner table.
template <class T>
__device__ __forceinline__ bool traverse(MeshDesc *D, const T &t, uint32_t num_verts, const int32_t *c2p, uint8_t *fvis, uint8_t *vvis, uint32_t *stack,
                         uint32_t stack_cap, uint32_t *d2c, int32_t *v2d, uint32_t *pids, uint32_t cap_entries, uint32_t *num_entries,
                         const uint8_t *bnd, uint32_t bnd_size) {
  const uint32_t F = t.num_faces();
  for (uint32_t f = 0; f < F; ++f) fvis[f] = 0;
  for (uint32_t v = 0; v < num_verts; ++v) { vvis[v] = 0; v2d[v] = -1; }
  uint32_t count = 0;
#define G_VISIT(v_, c_) { GREQ(count < cap_entries, 620); vvis[v_] = 1; pids[count] = (uint32_t)c2p[c_]; d2c[count] = (c_); v2d[v_] = (int32_t)count; ++count; }
  for (uint32_t f0 = 0; f0 < F; ++f0) {
    if (fvis[f0]) continue;
    uint32_t corner = 3 * f0, sp = 0;
    stack[sp++] = corner;
    const uint32_t nv = t.vertex(cnx(corner)), pv = t.vertex(cpv(corner));
    GREQ(nv < num_verts && pv < num_verts, 621);
    if (!vvis[nv]) G_VISIT(nv, cnx(corner));
    if (!vvis[pv]) G_VISIT(pv, cpv(corner));
    while (sp > 0) {
      corner = stack[sp - 1];
      uint32_t face = corner == DSA_INVALID ? DSA_INVALID : corner / 3;
      if (corner == DSA_INVALID || face >= F || fvis[face]) { --sp; continue; }
      for (;;) {
        fvis[face] = 1;
        // the three reads a face needs start together: its tip and the corners behind its two other edges
        const uint32_t v = t.vertex(corner);
        const uint32_t rc = t.right_corner(corner), lc = t.left_corner(corner);
        GREQ(v < num_verts, 622);
        if (!vvis[v]) {
          const bool on_boundary = bnd ? (v < bnd_size ? bnd[v] != 0 : true) : t.is_on_boundary(v);
          G_VISIT(v, corner);
          if (!on_boundary) {
            corner = rc;
            GREQ(corner != DSA_INVALID && corner / 3 < F, 623);
            face = corner / 3;
            continue;
          }
        }
        const uint32_t rf = rc == DSA_INVALID ? DSA_INVALID : rc / 3, lf = lc == DSA_INVALID ? DSA_INVALID : lc / 3;
        const bool rdone = rf >= F || fvis[rf], ldone = lf >= F || fvis[lf];
        if (rdone) {
          if (ldone) { --sp; break; }
          corner = lc; face = lf;
        } else {
          if (ldone) { corner = rc; face = rf; }
          else { GREQ(sp < stack_cap, 624); stack[sp - 1] = lc; stack[sp++] = rc; break; }
        }
      }
    }
  }
#undef G_VISIT
  *num_entries = count;
  return true;
}

// Traverser/MaxPredictionDegreeTraverser.cs:22-152 (D-28: with the degree list sized) + the same observer and
// sequencer.  The three priority stacks are linked lists threaded through one u32 per corner: an edge into a face is
// pushed at most once in a consistent table (only the face on its other side pushes it), and a second push is refused.
template <class T>
__device__ __forceinline__ bool traverse_prediction_degree(MeshDesc *D, const T &t, uint32_t num_verts, const int32_t *c2p, uint8_t *fvis, uint8_t *vvis,
                                           uint32_t *next, uint32_t *degree, uint32_t *d2c, int32_t *v2d, uint32_t *pids,
                                           uint32_t cap_entries, uint32_t *num_entries) {
  const uint32_t F = t.num_faces(), NEVER = 0xFFFFFFFEu, END = 0xFFFFFFFDu, USED = 0xFFFFFFFCu;
  for (uint32_t f = 0; f < F; ++f) fvis[f] = 0;
  for (uint32_t c = 0; c < 3 * F; ++c) next[c] = NEVER;
  for (uint32_t v = 0; v < num_verts; ++v) { vvis[v] = 0; v2d[v] = -1; degree[v] = 0; }
  uint32_t count = 0, head[3] = {END, END, END};
  int best = 0;
#define G_VISIT(v_, c_) { GREQ(count < cap_entries, 620); vvis[v_] = 1; pids[count] = (uint32_t)c2p[c_]; d2c[count] = (c_); v2d[v_] = (int32_t)count; ++count; }
#define G_PUSH(c_, pr_) { GREQ((c_) < 3 * F && next[c_] == NEVER, 625); next[c_] = head[pr_]; head[pr_] = (c_); if ((pr_) < best) best = (pr_); }
#define G_PRIORITY(c_, out_) { const uint32_t tip_ = t.vertex(c_); GREQ(tip_ < num_verts, 622); out_ = vvis[tip_] ? 0 : (++degree[tip_] > 1 ? 1 : 2); }
  for (uint32_t f0 = 0; f0 < F; ++f0) {
    if (fvis[f0]) continue;                            // a visited start face would be pushed, popped and dropped
    uint32_t corner = 3 * f0;
    G_PUSH(corner, 0);
    best = 0;
    const uint32_t nv = t.vertex(cnx(corner)), pv = t.vertex(cpv(corner)), tv = t.vertex(corner);
    GREQ(nv < num_verts && pv < num_verts && tv < num_verts, 621);
    if (!vvis[nv]) G_VISIT(nv, cnx(corner));
    if (!vvis[pv]) G_VISIT(pv, cpv(corner));
    if (!vvis[tv]) G_VISIT(tv, corner);
    for (;;) {
      corner = DSA_INVALID;                            // PopNextCornerToTraverse
      for (int i = best; i < 3; ++i)
        if (head[i] != END) { corner = head[i]; head[i] = next[corner]; next[corner] = USED; best = i; break; }
      if (corner == DSA_INVALID) break;
      if (fvis[corner / 3]) continue;
      for (;;) {
        fvis[corner / 3] = 1;
        const uint32_t v = t.vertex(corner);
        GREQ(v < num_verts, 622);
        if (!vvis[v]) G_VISIT(v, corner);
        const uint32_t rc = t.right_corner(corner), lc = t.left_corner(corner);
        const bool rdone = rc == DSA_INVALID || rc / 3 >= F || fvis[rc / 3], ldone = lc == DSA_INVALID || lc / 3 >= F || fvis[lc / 3];
        if (!ldone) {
          int pr;
          G_PRIORITY(lc, pr);
          if (rdone && pr <= best) { corner = lc; continue; }
          G_PUSH(lc, pr);
        }
        if (!rdone) {
          int pr;
          G_PRIORITY(rc, pr);
          if (pr <= best) { corner = rc; continue; }
          G_PUSH(rc, pr);
        }
        break;
      }
    }
  }
#undef G_VISIT
#undef G_PUSH
#undef G_PRIORITY
  *num_entries = count;
  return true;
}

// Parallelogram operands of every entry of a sequence (MeshPredictionSchemeParallelogramDecoder.cs:56-89), element
// parallel: para[3p..] = entries (next, prev, opposite) of the parallelogram across the entry's corner, next = INVALID
// when the entry falls back to delta.  Takes the table chase out of the serial prediction chain.
template <class T>
__device__ __forceinline__ void parallelogram_operands(const T &t, const uint32_t *d2c, const int32_t *v2d, uint32_t num_verts, uint32_t entries, uint32_t *para,
                                       uint32_t lane, uint32_t nl) {
  for (uint32_t p = lane; p < entries; p += nl) {
    uint32_t en = DSA_INVALID, ep = 0, eo = 0;
    const uint32_t oci = p > 0 ? t.opposite(d2c[p]) : DSA_INVALID;
    if (oci != DSA_INVALID) {
      const uint32_t a = t.vertex(oci), b = t.vertex(cnx(oci)), c = t.vertex(cpv(oci));
      if (a < num_verts && b < num_verts && c < num_verts) {
        const int32_t vo = v2d[a], vn = v2d[b], vp = v2d[c];
        if (vo >= 0 && vn >= 0 && vp >= 0 && vo < (int32_t)p && vn < (int32_t)p && vp < (int32_t)p) { en = (uint32_t)vn; ep = (uint32_t)vp; eo = (uint32_t)vo; }
      }
    }
    para[3 * p] = en; para[3 * p + 1] = ep; para[3 * p + 2] = eo;
  }
}

// MeshPredictionSchemeParallelogramDecoder.cs:29-54 + wrap transform, in place on corr -> values, on the operands above
__device__ __forceinline__ void parallelogram_wrap(const uint32_t *para, uint32_t entries, uint32_t nc, int32_t *w, int32_t mn, int32_t mx, int32_t max_dif) {
  for (uint32_t c = 0; c < nc; ++c) w[c] = wrap_original(0, w[c], mn, mx, max_dif);
  for (uint32_t p = 1; p < entries; ++p) {
    const uint32_t vn = para[3 * p], vp = para[3 * p + 1], vo = para[3 * p + 2];
    const bool ok = vn != DSA_INVALID;
    for (uint32_t c = 0; c < nc; ++c) {
      const int32_t pred = ok ? (int32_t)((uint32_t)w[vn * nc + c] + (uint32_t)w[vp * nc + c] - (uint32_t)w[vo * nc + c]) : w[(p - 1) * nc + c];
      w[p * nc + c] = wrap_original(pred, w[p * nc + c], mn, mx, max_dif);
    }
  }
}

// MeshPredictionSchemeParallelogramDecoder.cs:56-89 (TryComputeParallelogramPrediction)
template <class T>
__device__ __forceinline__ bool parallelogram_prediction(const T &t, const int32_t *v2d, uint32_t p, uint32_t ci, const int32_t *w, uint32_t nc, int32_t *pred) {
  const uint32_t oci = t.opposite(ci);
  if (oci == DSA_INVALID) return false;
  const uint32_t a = t.vertex(oci), b = t.vertex(cnx(oci)), c = t.vertex(cpv(oci));
  if (a == DSA_INVALID || b == DSA_INVALID || c == DSA_INVALID) return false;
  const int32_t vo = v2d[a], vn = v2d[b], vp = v2d[c];
  if (!(vo >= 0 && vn >= 0 && vp >= 0 && vo < (int32_t)p && vn < (int32_t)p && vp < (int32_t)p)) return false;
  for (uint32_t k = 0; k < nc; ++k) pred[k] = (int32_t)((uint32_t)w[vn * nc + k] + (uint32_t)w[vp * nc + k] - (uint32_t)w[vo * nc + k]);
  return true;
}

// MeshPredictionSchemeMultiParallelogramDecoder.cs:24-73 (D-27: the sum is cleared per entry) and
// MeshPredictionSchemeConstrainedMultiParallelogramDecoder.cs:28-108 (D-14 resolved to the bitstream), in place.
// constrained: up to four parallelograms, left swing first, each kept or dropped by the next crease flag of the
// context (= parallelograms found - 1); the four flag streams are read as the entries go by.
template <class T>
__device__ __forceinline__ bool multi_parallelogram_wrap(MeshDesc *D, const T &t, const uint32_t *d2c, const int32_t *v2d, uint32_t entries, uint32_t nc, int32_t *w,
                                         int32_t mn, int32_t mx, int32_t max_dif, bool constrained, Rabs *crease, uint32_t *crease_left, uint32_t max_steps) {
  if (nc > 4) GNOTIMPL(660);                          // wider integer attributes: not on the device yet
  int32_t cand[4][4], sum[4];
  for (uint32_t c = 0; c < nc; ++c) w[c] = wrap_original(0, w[c], mn, mx, max_dif);
  for (uint32_t p = 1; p < entries; ++p) {
    const uint32_t start = d2c[p];
    uint32_t c = start, found = 0, used = 0, steps = 0;
    for (uint32_t k = 0; k < nc; ++k) sum[k] = 0;
    if (!constrained) {
      while (c != DSA_INVALID) {
        GREQ(++steps <= max_steps, 661);
        if (parallelogram_prediction(t, v2d, p, c, w, nc, cand[0])) { for (uint32_t k = 0; k < nc; ++k) sum[k] = (int32_t)((uint32_t)sum[k] + (uint32_t)cand[0][k]); ++found; }
        c = t.swing_right(c);
        if (c == start) c = DSA_INVALID;
      }
      used = found;
    } else {
      bool first_pass = true;
      while (c != DSA_INVALID) {
        GREQ(++steps <= max_steps, 661);
        if (parallelogram_prediction(t, v2d, p, c, w, nc, cand[found])) { if (++found == 4) break; }
        c = first_pass ? t.swing_left(c) : t.swing_right(c);
        if (c == start) break;
        if (c == DSA_INVALID && first_pass) { first_pass = false; c = t.swing_right(start); }
      }
      for (uint32_t i = 0; i < found; ++i) {
        const uint32_t context = found - 1;
        GREQ(crease_left[context] > 0, 662);
        --crease_left[context];
        if (!crease[context].next()) { ++used; for (uint32_t k = 0; k < nc; ++k) sum[k] = (int32_t)((uint32_t)sum[k] + (uint32_t)cand[i][k]); }
      }
    }
    for (uint32_t k = 0; k < nc; ++k) {
      const int32_t pred = used ? sum[k] / (int32_t)used : w[(p - 1) * nc + k];
      w[p * nc + k] = wrap_original(pred, w[p * nc + k], mn, mx, max_dif);
    }
  }
  return true;
}

__device__ __forceinline__ uint64_t int_sqrt(uint64_t number) {   // Core/MathUtilities.cs:5-25
  if (number == 0) return 0;
  uint64_t act = number, root = 1;
  while (act >= 2) { root *= 2; act /= 4; }
  do { root = (root + number / root) / 2; } while (root * root > number);
  return root;
}

// MeshPredictionSchemeTexCoordsPortableDecoder.cs:50-85 + ...PortablePredictor.cs:46-150, in place
template <class T>
__device__ __forceinline__ bool texcoords_portable_wrap(MeshDesc *D, const T &t, const uint32_t *d2c, const int32_t *v2d, uint32_t entries, int32_t *w,
                                        const uint32_t *entry_to_point, const int32_t *pos, const uint32_t *pos_map, uint32_t num_points,
                                        uint32_t pos_entries, const uint8_t *orient, uint32_t num_orient, int32_t mn, int32_t mx, int32_t max_dif) {
  uint32_t left = num_orient;
  for (uint32_t p = 0; p < entries; ++p) {
    const int32_t data_id = (int32_t)p;
    const uint32_t ci = d2c[p];
    const uint32_t vnx = t.vertex(cnx(ci)), vpv = t.vertex(cpv(ci));
    const int32_t next_id = vnx != DSA_INVALID ? v2d[vnx] : -1, prev_id = vpv != DSA_INVALID ? v2d[vpv] : -1;
    int32_t pred[2] = {0, 0};
    bool done = false;
    if (prev_id >= 0 && next_id >= 0 && prev_id < data_id && next_id < data_id) {
      const int64_t n_uv[2] = {w[next_id * 2], w[next_id * 2 + 1]}, p_uv[2] = {w[prev_id * 2], w[prev_id * 2 + 1]};
      if (p_uv[0] == n_uv[0] && p_uv[1] == n_uv[1]) { pred[0] = (int32_t)p_uv[0]; pred[1] = (int32_t)p_uv[1]; done = true; }
      else {
        int64_t tip[3], np[3], pp[3];
        const int32_t ids[3] = {data_id, next_id, prev_id};
        int64_t *dst[3] = {tip, np, pp};
        for (int q = 0; q < 3; ++q) {
          const uint32_t point = entry_to_point[ids[q]];
          GREQ(point < num_points, 630);
          const uint32_t e = pos_map[point];
          GREQ(e < pos_entries, 631);
          for (int k = 0; k < 3; ++k) dst[q][k] = pos[(size_t)e * 3 + k];
        }
        const int64_t pn[3] = {pp[0] - np[0], pp[1] - np[1], pp[2] - np[2]};
        const int64_t pn_norm2 = pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2];
        if (pn_norm2 != 0) {
          const int64_t cn[3] = {tip[0] - np[0], tip[1] - np[1], tip[2] - np[2]};
          const int64_t cn_dot_pn = pn[0] * cn[0] + pn[1] * cn[1] + pn[2] * cn[2];
          const int64_t pn_uv[2] = {p_uv[0] - n_uv[0], p_uv[1] - n_uv[1]};
          const int64_t x_uv[2] = {n_uv[0] * pn_norm2 + cn_dot_pn * pn_uv[0], n_uv[1] * pn_norm2 + cn_dot_pn * pn_uv[1]};
          int64_t x_pos[3];
          for (int k = 0; k < 3; ++k) x_pos[k] = np[k] + (cn_dot_pn * pn[k]) / pn_norm2;
          const int64_t cx[3] = {tip[0] - x_pos[0], tip[1] - x_pos[1], tip[2] - x_pos[2]};
          const uint64_t cx_norm2 = (uint64_t)(cx[0] * cx[0] + cx[1] * cx[1] + cx[2] * cx[2]);
          int64_t cx_uv[2] = {pn_uv[1], -pn_uv[0]};
          const int64_t norm = (int64_t)int_sqrt(cx_norm2 * (uint64_t)pn_norm2);
          cx_uv[0] *= norm; cx_uv[1] *= norm;
          GREQ(left > 0, 632);
          const bool orientation = orient[--left] != 0;
          int64_t pu, pv;
          if (orientation) { pu = (x_uv[0] + cx_uv[0]) / pn_norm2; pv = (x_uv[1] + cx_uv[1]) / pn_norm2; }
          else { pu = (x_uv[0] - cx_uv[0]) / pn_norm2; pv = (x_uv[1] - cx_uv[1]) / pn_norm2; }
          pred[0] = (int32_t)pu; pred[1] = (int32_t)pv;
          done = true;
        }
      }
    }
    if (!done) {
      int32_t data_offset = 0;
      bool zero = false;
      if (prev_id >= 0 && prev_id < data_id) data_offset = prev_id * 2;
      if (next_id >= 0 && next_id < data_id) data_offset = next_id * 2;
      else {
        if (data_id > 0) data_offset = (data_id - 1) * 2;
        else zero = true;
      }
      if (!zero) { pred[0] = w[data_offset]; pred[1] = w[data_offset + 1]; }
    }
    w[p * 2] = wrap_original(pred[0], w[p * 2], mn, mx, max_dif);
    w[p * 2 + 1] = wrap_original(pred[1], w[p * 2 + 1], mn, mx, max_dif);
  }
  return true;
}

// MeshPredictionSchemeGeometricNormalDecoder.cs:44-82 + ...GeometricNormalPredictorArea.cs:16-63 +
// OctahedronToolBox.cs:28-77,121-137 with the bitstream's 64-bit arithmetic (D-9, D-10, D-23..D-25), in place on corr -> values.
// The flip bits (one per entry, a serial rABS stream) are decoded by the values stage into a byte array; the prediction
// itself needs only the decoded positions, so it runs afterwards on the whole wave (ATT_NORMALS).
template <class T>
__device__ __forceinline__ bool geometric_normal_oct(MeshDesc *D, const T &t, const uint32_t *d2c, const int32_t *v2d, uint32_t entries, int32_t *w,
                                     const uint32_t *entry_to_point, uint32_t v2d_size, const int32_t *pos, const uint32_t *pos_map,
                                     uint32_t num_points, uint32_t pos_entries, const uint8_t *flips, const OctParams &o, bool canonical, uint32_t max_steps,
                                     uint32_t lane, uint32_t nl) {
  for (uint32_t p = lane; p < entries; p += nl) {                // entries are independent of each other: element parallel
    const uint32_t ci = d2c[p];
    int64_t center[3] = {0, 0, 0};
    // position of the vertex at a corner: corner -> vertex -> data id -> point -> position entry
#define G_POS(corner, dst)                                                         \
    {                                                                              \
      const uint32_t v_ = t.vertex(corner);                                        \
      GREQ(v_ != DSA_INVALID && v_ < v2d_size, 650);                               \
      const int32_t d_ = v2d[v_];                                                  \
      GREQ(d_ >= 0 && (uint32_t)d_ < entries, 651);                                \
      const uint32_t point_ = entry_to_point[d_];                                  \
      GREQ(point_ < num_points, 652);                                              \
      const uint32_t e_ = pos_map[point_];                                         \
      GREQ(e_ < pos_entries, 653);                                                 \
      for (int k_ = 0; k_ < 3; ++k_) (dst)[k_] = pos[(size_t)e_ * 3 + k_];         \
    }
    G_POS(ci, center);
    uint64_t n[3] = {0, 0, 0};
    uint32_t c = ci, steps = 0;
    bool left = true;
    while (c != DSA_INVALID) {                          // VertexCornersIterator.cs: left from the corner, then right from it
      GREQ(++steps <= max_steps, 654);
      int64_t pn[3], pp[3];
      G_POS(cnx(c), pn);
      G_POS(cpv(c), pp);
      uint64_t a[3], b[3];
      for (int k = 0; k < 3; ++k) { a[k] = (uint64_t)(pn[k] - center[k]); b[k] = (uint64_t)(pp[k] - center[k]); }
      n[0] += a[1] * b[2] - a[2] * b[1];
      n[1] += a[2] * b[0] - a[0] * b[2];
      n[2] += a[0] * b[1] - a[1] * b[0];
      if (left) {
        c = t.swing_left(c);
        if (c == DSA_INVALID) { c = t.swing_right(ci); left = false; }
        else if (c == ci) break;
      } else c = t.swing_right(c);
    }
#undef G_POS
    int32_t os, ot;
    geometric_normal_finish(o, canonical, n, flips[p] != 0, w[2 * p], w[2 * p + 1], os, ot);
    w[2 * p] = os; w[2 * p + 1] = ot;
  }
  return true;
}

// What the prediction schemes of one attributes decoder see.  ct == nullptr: no corner table (linear sequencing of a
// sequential mesh); act != nullptr: the attribute corner table of the decoder's attribute data.
struct ValueCtx {
  const Ct *ct;
  const Act *act;
  const uint32_t *d2c;
  const int32_t *v2d;
  const uint32_t *pids;
  uint8_t *orient;
  uint32_t orient_cap, num_points;
  uint32_t num_verts, num_corners;   // size of v2d; bound on a corner fan
  const uint32_t *para_ct, *para_act; // parallelogram operands of the sequence on the position / attribute corner table
};

// Values of attribute ai (SequentialAttributeDecoder.cs:47-52,75-86 / SequentialIntegerAttributeDecoder.cs:23-101):
// symbols -> corrections -> portable values in work[ai].
__device__ __forceinline__ bool decode_values(uint8_t *arena, const MeshLayout &L, MeshDesc *D, Rd &r, uint32_t ai, uint32_t entries, const RansScratch &rs,
                              const ValueCtx &vc) {
  AttrDesc &a = D->att[ai];
  a.num_entries = entries;
  a.pred_method = -2; a.pred_transform = -1; a.have_scheme = 0; a.pred_kind = 0;
  if (a.seq_type == 0) {                               // SequentialAttributeDecoder.cs:75-86
    a.source = SRC_BYTES;
    a.off_raw = r.pos;
    const uint64_t bytes = (uint64_t)data_type_length(a.data_type) * a.nc * entries;
    GREQ(bytes <= L.out_cap[ai], 140);
    r.skip(bytes);
    GREQ(r.ok, 141);
    return true;
  }
  const uint32_t nc = a.seq_type == 3 ? 2u : a.nc;
  a.nc_portable = (uint8_t)nc;
  const uint64_t num_values = (uint64_t)entries * nc;
  GREQ(num_values <= L.work_cap[ai], 142);
  int32_t *w = (int32_t *)(arena + L.work[ai]);
  const int method = (int8_t)r.u8();
  GREQ(r.ok && method >= -2 && method < 7, 143);
  a.pred_method = (int8_t)method;
  int tt = -1;
  if (method != -2) {
    tt = (int8_t)r.u8();
    GREQ(r.ok && tt >= -1 && tt < 4, 144);
    a.pred_transform = (int8_t)tt;
    a.have_scheme = a.seq_type == 3 ? (tt == 2 || tt == 3) : (tt == 1);
  }
  const uint32_t compressed = r.u8();
  GREQ(r.ok, 145);
  a.source = SRC_RAW;
  const bool positive = a.have_scheme && (tt == 2 || tt == 3);      // D-4: zig-zag unless the transform's corrections are positive
  const uint64_t t_sym = gclk();
  if (compressed > 0) {
    if (!decode_symbols(D, r, (uint32_t)num_values, nc, (uint32_t *)w, rs, !positive)) return false;
  } else {                                             // SequentialIntegerAttributeDecoder.cs:68-84 (D-13)
    const uint32_t nb = r.u8();
    GREQ(r.ok && nb >= 1 && nb <= 4, 158);
    for (uint64_t k = 0; k < num_values; ++k) { uint32_t v = 0; for (uint32_t q = 0; q < nb; ++q) v |= r.u8() << (8 * q); ((uint32_t *)w)[k] = positive ? v : unzigzag(v); }
    GREQ(r.ok, 160);
  }
  D->dbg[10] += (uint32_t)((gclk() - t_sym) >> 4);     // diagnostics: entropy decode of all attributes, in units of 16 clocks
  if (!a.have_scheme) return true;
  // scheme selection, PredictionSchemeDecoderFactory.cs:9-76
  int eff = method;
  if (vc.ct == nullptr) eff = 0;                       // no corner table (linear sequencing): every scheme falls back to delta
  // which mesh schemes exist depends on the transform (D-26): wrap carries the parallelogram family and the texture
  // coordinate schemes, the octahedral transforms carry only the geometric normal scheme; the rest is delta
  else if (tt == 1) { if (method == 1 || method == 2 || method == 4 || method == 5) eff = method; else if (method == 0 || method == 6) eff = 0; else GNOTIMPL(161); }
  else eff = method == 6 ? 6 : 0;
  a.pred_kind = (int8_t)eff;
  uint8_t *orient = nullptr;
  uint32_t num_orient = 0;
  if (eff == 5) {                                      // MeshPredictionSchemeTexCoordsPortableDecoder.cs:66-85
    GREQ(vc.orient != nullptr, 680);                   // orientation scratch lives in the attribute data block
    const int32_t num_or = (int32_t)r.u32();
    GREQ(r.ok && num_or >= 0 && (uint32_t)num_or <= vc.orient_cap, 681);
    Rabs rd;
    uint32_t endp;
    rd.start(arena + L.stream, L.stream_len, r.pos, &endp);
    GREQ(rd.ok, 682);
    r.pos = endp;
    orient = vc.orient;
    bool last = true;
    for (int32_t k = 0; k < num_or; ++k) { if (rd.next() == 0) last = !last; orient[k] = last ? 1 : 0; }
    num_orient = (uint32_t)num_or;
  }
  Rabs crease[4];
  uint32_t crease_left[4] = {0, 0, 0, 0};
  if (eff == 4) {                                      // MeshPredictionSchemeConstrainedMultiParallelogramDecoder.cs:110-134 (v2.2: no mode byte)
    for (int i = 0; i < 4; ++i) {
      const uint64_t num_flags = r.varint();
      GREQ(r.ok && num_flags <= vc.num_corners, 663);
      crease_left[i] = (uint32_t)num_flags;
      if (num_flags > 0) {
        uint32_t endp;
        crease[i].start(arena + L.stream, L.stream_len, r.pos, &endp);
        GREQ(crease[i].ok, 664);
        r.pos = endp;
      }
    }
  }
  if (tt == 1) {                                       // PredictionSchemeWrapDecodingTransform.cs:69-75
    a.wrap_min = (int32_t)r.u32(); a.wrap_max = (int32_t)r.u32();
    GREQ(r.ok && a.wrap_min <= a.wrap_max, 162);
    const int64_t dif = (int64_t)a.wrap_max - (int64_t)a.wrap_min;
    GREQ(dif < 0x7FFFFFFF, 163);
    const int32_t mn = a.wrap_min, mx = a.wrap_max, max_dif = (int32_t)(1 + dif);
    if (num_values == 0) return true;
    if (eff == 0) {                                    // PredictionSchemeDeltaDecoder.cs:23-37
      for (uint32_t c = 0; c < nc; ++c) w[c] = wrap_original(0, w[c], mn, mx, max_dif);
      for (uint64_t k = nc; k < num_values; ++k) w[k] = wrap_original(w[k - nc], w[k], mn, mx, max_dif);
    } else if (eff == 1) {
      parallelogram_wrap(vc.act ? vc.para_act : vc.para_ct, entries, nc, w, mn, mx, max_dif);
    } else if (eff == 2 || eff == 4) {
      if (vc.act) return multi_parallelogram_wrap(D, *vc.act, vc.d2c, vc.v2d, entries, nc, w, mn, mx, max_dif, eff == 4, crease, crease_left, vc.num_corners + 1);
      return multi_parallelogram_wrap(D, *vc.ct, vc.d2c, vc.v2d, entries, nc, w, mn, mx, max_dif, eff == 4, crease, crease_left, vc.num_corners + 1);
    } else {
      GREQ(nc == 2, 683);
      // parent = portable positions, SequentialAttributeDecoder.cs:58-73
      int pa = -1;
      for (uint32_t q = 0; q < ai; ++q) if (D->att[q].att_type == 0 && D->att[q].seq_type != 0) { pa = (int)q; break; }
      GREQ(pa >= 0 && D->att[pa].nc_portable == 3, 684);
      const int32_t *pos = (const int32_t *)(arena + L.work[pa]);
      const uint32_t *pos_map = (const uint32_t *)(arena + L.map[pa]);
      bool ok;
      if (vc.act) ok = texcoords_portable_wrap(D, *vc.act, vc.d2c, vc.v2d, entries, w, vc.pids, pos, pos_map, vc.num_points, D->att[pa].num_entries, orient, num_orient, mn, mx, max_dif);
      else ok = texcoords_portable_wrap(D, *vc.ct, vc.d2c, vc.v2d, entries, w, vc.pids, pos, pos_map, vc.num_points, D->att[pa].num_entries, orient, num_orient, mn, mx, max_dif);
      if (!ok) return false;
    }
  } else {                                             // normal octahedron transforms (D-19)
    const int32_t max_q = (int32_t)r.u32();
    if (tt == 3) (void)r.u32();
    GREQ(r.ok && max_q > 0 && (max_q & 1) == 1, 164);
    a.oct_max_q = max_q;
    OctParams o;
    const int q = 32 - __builtin_clz((uint32_t)max_q);
    GREQ(q >= 2 && q <= 30, 165);
    const int32_t max_value = (1 << q) - 2;
    o.center = max_value / 2; o.max_q = (1 << q) - 1;
    if (eff == 6) {                                    // MeshPredictionSchemeGeometricNormalDecoder.cs:71-82: flip bits behind the transform data
      Rabs rd;
      uint32_t endp;
      rd.start(arena + L.stream, L.stream_len, r.pos, &endp);
      GREQ(rd.ok, 655);
      r.pos = endp;
      if (num_values == 0) return true;
      int pa = -1;                                     // parent = portable positions, SequentialAttributeDecoder.cs:58-73
      for (uint32_t k = 0; k < ai; ++k) if (D->att[k].att_type == 0 && D->att[k].seq_type != 0) { pa = (int)k; break; }
      GREQ(pa >= 0 && D->att[pa].nc_portable == 3, 656);
      GREQ(entries <= L.out_cap[ai], 657);             // the flip bits wait in the attribute's output region (written last, by k_finalize)
      uint8_t *flips = arena + L.out[ai];
      for (uint32_t p = 0; p < entries; ++p) flips[p] = (uint8_t)rd.next();
      return true;                                     // corrections stay in w; normals_stage() finishes the attribute
    }
    if (num_values == 0) return true;
    int32_t ps = 0, pt = 0;
    for (uint32_t e = 0; e < entries; ++e) {
      int32_t os, ot;
      oct_original(o, tt == 3, ps, pt, w[2 * e], w[2 * e + 1], os, ot);
      w[2 * e] = os; w[2 * e + 1] = ot;
      ps = os; pt = ot;
    }
  }
  return true;
}

// ATT_NORMALS: the GeometricNormal attributes of one decoder, on the whole wave.
__device__ __forceinline__ bool normals_stage(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t first_att, uint32_t num_atts, uint32_t entries,
                              const ValueCtx &vc, uint32_t lane, uint32_t nl) {
  for (uint32_t ai = first_att; ai < first_att + num_atts; ++ai) {
    const AttrDesc &a = D->att[ai];
    if (a.seq_type != 3 || !a.have_scheme || a.pred_kind != 6 || entries == 0) continue;
    int pa = -1;
    for (uint32_t k = 0; k < ai; ++k) if (D->att[k].att_type == 0 && D->att[k].seq_type != 0) { pa = (int)k; break; }
    GREQ(pa >= 0 && D->att[pa].nc_portable == 3, 656);
    OctParams o;
    const int q = 32 - __builtin_clz((uint32_t)a.oct_max_q);
    o.max_q = (1 << q) - 1; o.center = ((1 << q) - 2) / 2;
    int32_t *w = (int32_t *)(arena + L.work[ai]);
    const int32_t *pos = (const int32_t *)(arena + L.work[pa]);
    const uint32_t *pos_map = (const uint32_t *)(arena + L.map[pa]);
    const bool canonical = a.pred_transform == 3;
    const bool ok = vc.act ? geometric_normal_oct(D, *vc.act, vc.d2c, vc.v2d, entries, w, vc.pids, vc.num_verts, pos, pos_map, vc.num_points, D->att[pa].num_entries, arena + L.out[ai], o, canonical, vc.num_corners + 1, lane, nl)
                           : geometric_normal_oct(D, *vc.ct, vc.d2c, vc.v2d, entries, w, vc.pids, vc.num_verts, pos, pos_map, vc.num_points, D->att[pa].num_entries, arena + L.out[ai], o, canonical, vc.num_corners + 1, lane, nl);
    if (!ok) return false;
  }
  return true;
}

// AttributeQuantizationTransform.cs:110-121 / AttributeOctahedronTransform.cs:39-42 (D-5)
__device__ __forceinline__ bool decode_transform_params(MeshDesc *D, Rd &r, uint32_t ai) {
  AttrDesc &a = D->att[ai];
  if (a.seq_type == 2) {
    for (uint32_t c = 0; c < a.nc; ++c) a.q_min[c] = r.f32();
    a.q_range = r.f32();
    a.q_bits = (uint8_t)r.u8();
    GREQ(r.ok && a.q_bits >= 1 && a.q_bits <= 30, 135);
  } else if (a.seq_type == 3) {
    a.q_bits = (uint8_t)r.u8();
    GREQ(r.ok && a.q_bits >= 2 && a.q_bits <= 30, 136);
  }
  return true;
}

// Attribute descriptors of one decoder (AttributesDecoder.cs:19-63 + SequentialAttributeDecodersController.cs:16-27)
__device__ __forceinline__ bool decode_descriptors(const MeshLayout &L, MeshDesc *D, Rd &r, uint32_t decoder, uint32_t &natt, uint32_t *first_att, uint32_t *num_atts) {
  *first_att = natt;
  const uint64_t k = r.varint();
  GREQ(r.ok, 129);
  if (natt + k > DSA_MAX_ATT) GNOTIMPL(129);            // a valid stream with more attributes than the device path carries
  GREQ(natt + k <= L.cap_attributes, 129);
  *num_atts = (uint32_t)k;
  for (uint32_t j = 0; j < (uint32_t)k; ++j) {
    AttrDesc &a = D->att[natt + j];
    a.att_type = (uint8_t)r.u8(); a.data_type = (uint8_t)r.u8(); a.nc = (uint8_t)r.u8(); a.normalized = r.u8() != 0;
    GREQ(r.ok && a.att_type < 5 && a.data_type != 0 && a.data_type < 12 && a.nc != 0, 130);
    a.unique_id = (uint32_t)r.varint();
    a.decoder_id = (int8_t)decoder;
  }
  for (uint32_t j = 0; j < (uint32_t)k; ++j) {
    AttrDesc &a = D->att[natt + j];
    a.seq_type = (uint8_t)r.u8();
    GREQ(r.ok && a.seq_type <= 3, 131);
    if (a.seq_type == 2) GREQ(a.data_type == 9 && a.nc <= 4, 132);
    if (a.seq_type == 3) GREQ(a.data_type == 9 && a.nc == 3, 133);
    if (a.seq_type == 1) { const uint32_t w = data_type_length(a.data_type); GREQ(w == 1 || w == 2 || w == 4, 134); }
  }
  natt += (uint32_t)k;
  return true;
}

// Mesh/MeshSequentialDecoder.cs:8-123: faces as point indices (compressed: differences with the sign in the LSB
// through the symbol coder -- D-22: the C# tests that bit inverted; raw: u8 / u16 / varint / u32 by point count), one
// attributes decoder, linear sequencing (entry i = point i).
__device__ __forceinline__ bool decode_sequential_mesh(uint8_t *arena, const MeshLayout &L, MeshDesc *D, Rd &r, RansScratch rs) {
  const uint64_t nf64 = r.varint(), np64 = r.varint();
  GREQ(r.ok && nf64 <= 0x7FFFFFFFu / 3 && np64 <= 0x7FFFFFFFu, 111);
  const uint32_t F = (uint32_t)nf64, NP = (uint32_t)np64;
  GREQ(F == L.cap_faces && NP == L.cap_vertices, 116);
  const GenLayout g = gen_layout(F, NP, 0, 0, L.stream_len);
  GREQ(g.total <= L.gen_bytes, 640);
  rs.cum = (uint32_t *)(arena + L.gen + g.cum); rs.cum_cap = g.cum_entries;
  int32_t *faces = (int32_t *)(arena + L.faces);
  const uint32_t method = r.u8();
  GREQ(r.ok && method <= 1, 690);
  if (method == 0) {
    if (!decode_symbols(D, r, 3 * F, 1, (uint32_t *)faces, rs)) return false;
    int32_t last = 0;
    for (uint32_t k = 0; k < 3 * F; ++k) {
      const uint32_t e = (uint32_t)faces[k];
      int32_t diff = (int32_t)(e >> 1);
      if (e & 1u) { GREQ(diff <= last, 691); diff = -diff; }
      else GREQ(diff <= 0x7FFFFFFF - last, 692);
      last += diff;
      faces[k] = last;
    }
  } else {
    for (uint32_t k = 0; k < 3 * F; ++k) {
      if (NP < 256) faces[k] = (int32_t)r.u8();
      else if (NP < (1u << 16)) faces[k] = (int32_t)r.u16();
      else if (NP < (1u << 21)) faces[k] = (int32_t)(uint32_t)r.varint();
      else faces[k] = (int32_t)r.u32();
    }
    GREQ(r.ok, 693);
  }
  D->num_faces = F; D->num_points = NP; D->num_vertices = NP; D->num_all_vertices = NP; D->num_enc_vertices = NP; D->num_entries = NP;
  D->off_attributes = r.pos;
  const uint32_t ndec = r.u8();
  GREQ(r.ok, 122);
  if (ndec > DSA_MAX_ATT) GNOTIMPL(122);
  D->num_decoders = ndec;
  uint32_t natt = 0, first[DSA_MAX_ATT], count[DSA_MAX_ATT];
  for (uint32_t i = 0; i < ndec; ++i) if (!decode_descriptors(L, D, r, i, natt, &first[i], &count[i])) return false;
  D->num_attributes = natt;
  ValueCtx vc = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, NP, 0, 0, nullptr, nullptr};
  for (uint32_t i = 0; i < ndec; ++i) {
    for (uint32_t ai = first[i]; ai < first[i] + count[i]; ++ai) {
      uint32_t *map = (uint32_t *)(arena + L.map[ai]);
      for (uint32_t p = 0; p < NP; ++p) map[p] = p;           // LinearSequencer.cs:3-19
    }
    for (uint32_t ai = first[i]; ai < first[i] + count[i]; ++ai) if (!decode_values(arena, L, D, r, ai, NP, rs, vc)) return false;
    for (uint32_t ai = first[i]; ai < first[i] + count[i]; ++ai) if (!decode_transform_params(D, r, ai)) return false;
  }
  D->end_pos = r.pos;
  return true;
}

struct DecoderInfo { int att_data_id; uint32_t element_type, traversal_method, first_att, num_atts, num_entries; };

// Pointers and sizes of one general mesh, rebuilt by every phase from the layout and the descriptor.
struct MeshCtx {
  uint32_t F, C, VMAX, NVMAX, nad;
  GenLayout g;
  uint8_t *G;
  Ct ct;
  Act act[DSA_MAX_ATT_DATA];
  uint8_t *is_hole;
  int32_t *c2p;
};
__device__ __forceinline__ bool mesh_ctx(uint8_t *arena, const MeshLayout &L, MeshDesc *D, MeshCtx &m) {
  m.F = D->num_faces; m.C = 3 * m.F; m.VMAX = L.cap_vertices; m.NVMAX = m.C > m.VMAX ? m.C : m.VMAX; m.nad = D->num_att_data;
  m.g = gen_layout(m.F, m.VMAX, L.cap_splits, m.nad, L.stream_len);
  GREQ(m.g.total <= L.gen_bytes, 640);
  m.G = arena + L.gen;
  // arrays (the fast path's regions are free for a general mesh)
  m.ct.opp = (uint32_t *)(arena + L.frec); m.ct.c2v = m.ct.opp + m.C; m.ct.vcorner = (uint32_t *)(arena + L.vrec);
  m.ct.F = m.F; m.ct.C = m.C; m.ct.nv = D->num_all_vertices; m.ct.vmax = m.VMAX;
  m.is_hole = arena + L.vvis;
  m.c2p = (int32_t *)(arena + L.faces);
  for (uint32_t d = 0; d < m.nad; ++d) {
    uint8_t *blk = m.G + m.g.data + (uint64_t)d * m.g.data_stride;
    m.act[d].ct = &m.ct;
    m.act[d].edge_seam = blk + m.g.edge_seam; m.act[d].vert_seam = blk + m.g.vert_seam;
    m.act[d].c2v = (uint32_t *)(blk + m.g.c2v); m.act[d].v2lm = (uint32_t *)(blk + m.g.v2lm);
    m.act[d].nv = D->gen_act_nv[d];
  }
  return true;
}

// Lane abstraction of the cooperative phase: a wave on the device, a single "lane" in the host check.
#if defined(__HIPCC__)
#define G_NL 64u
__device__ __forceinline__ uint32_t g_lane() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t g_excl_scan(uint32_t x, uint32_t *total) { return wave_excl_scan(x, total); }
__device__ __forceinline__ bool g_any(bool b) { return __ballot(b) != 0; }
#else
#define G_NL 1u
inline uint32_t g_lane() { return 0; }
inline uint32_t g_excl_scan(uint32_t x, uint32_t *total) { *total = x; return 0; }
inline bool g_any(bool b) { return b; }
#endif

// Phase 1 (one lane): connectivity header, topology splits, Edgebreaker symbols (standard or valence), start faces,
// vertex compaction, attribute seam bits.  Leaves the reader position in D->end_pos for phase 3.
__device__ __forceinline__ bool mesh_connectivity(uint8_t *arena, const MeshLayout &L, MeshDesc *D, Rd &r, RansScratch rs) {
  const uint8_t *s = arena + L.stream;
  const uint64_t t0 = gclk();
  // ---------------------------------------------------------------- MeshEdgeBreakerDecoder.cs:25-134
  D->traversal_type = (uint8_t)r.u8();
  GREQ(r.ok && D->traversal_type <= 2, 109);
  const bool valence = D->traversal_type == 2, predictive = D->traversal_type == 1;
  const uint64_t nv64 = r.varint(), nf64 = r.varint();
  GREQ(r.ok && nf64 <= 0x7FFFFFFFu / 3 && nv64 <= nf64 * 3, 111);
  GREQ(nv64 * (nv64 - 1) / 2 >= 3 * nf64 / 2, 112);
  const uint32_t nad = r.u8();
  const uint64_t nsym64 = r.varint();
  GREQ(r.ok && nf64 >= nsym64 && nf64 <= nsym64 + nsym64 / 3, 113);
  const uint64_t nss64 = r.varint();
  GREQ(r.ok && nss64 <= nsym64, 114);
  if (nad > DSA_MAX_ATT_DATA) GNOTIMPL(115);
  const uint32_t F = (uint32_t)nf64, C = 3 * F, num_symbols = (uint32_t)nsym64;
  const uint32_t VMAX = (uint32_t)(nv64 + nss64);
  GREQ(F == L.cap_faces && VMAX == L.cap_vertices, 116);
  D->num_enc_vertices = (uint32_t)nv64; D->num_faces = F; D->num_att_data = (uint8_t)nad;
  D->num_symbols = num_symbols; D->num_split_symbols = (uint32_t)nss64;
  const uint64_t nsplits64 = r.varint();
  GREQ(r.ok && nsplits64 <= F && nsplits64 <= L.cap_splits, 117);
  const uint32_t S = (uint32_t)nsplits64;
  D->num_splits = S;
  D->num_all_vertices = 0;
  for (uint32_t d = 0; d < DSA_MAX_ATT_DATA; ++d) D->gen_act_nv[d] = 0;
  MeshCtx m;
  if (!mesh_ctx(arena, L, D, m)) return false;
  const GenLayout &g = m.g;
  uint8_t *G = m.G;
  Ct ct = m.ct;                 // by value: the hot loop below keeps the table pointers and the vertex count in registers
  uint8_t *is_hole = m.is_hole;
  uint32_t *valences = (uint32_t *)(arena + L.vstamp);
  uint32_t *ctx_syms = (uint32_t *)(arena + L.fstamp);
  uint32_t *invalid_list = (uint32_t *)(arena + L.vrank);
  uint32_t *stack = (uint32_t *)(G + g.stack);
  uint32_t *splits = (uint32_t *)(G + g.splits), *active = (uint32_t *)(G + g.active);
  rs.cum = (uint32_t *)(G + g.cum); rs.cum_cap = g.cum_entries;
  for (uint32_t c = 0; c < C; ++c) { ct.opp[c] = DSA_INVALID; ct.c2v[c] = DSA_INVALID; }
  for (uint32_t v = 0; v < VMAX; ++v) is_hole[v] = 1;
  if (S > 0) for (uint32_t f = 0; f < F; ++f) active[f] = DSA_INVALID;   // topologySplitActiveCorners as a direct map
  // topology splits, :136-230
  {
    uint32_t last = 0;
    for (uint32_t i = 0; i < S; ++i) {
      const uint32_t source = (uint32_t)r.varint() + last;
      const uint32_t delta = (uint32_t)r.varint();
      GREQ(r.ok && delta <= source, 200);
      splits[3 * i] = source; splits[3 * i + 1] = source - delta; splits[3 * i + 2] = 0;
      last = source;
    }
    if (S > 0) {
      const uint8_t *bits = r.p + r.pos;
      const uint32_t nbytes = r.n - r.pos;
      for (uint32_t i = 0; i < S; ++i) splits[3 * i + 2] = read_bits(bits, nbytes, i, 1);
      r.skip(((uint64_t)S + 7) >> 3);
      GREQ(r.ok, 118);
    }
  }
  // traversal decoder start, MeshEdgeBreakerTraversalDecoder.cs:27-61 / ...ValenceDecoder.cs:22-69
  const uint8_t *sym_bits = nullptr;
  uint32_t sym_nbytes = 0;
  uint64_t sym_bitpos = 0;
  if (!valence) {                                          // standard and predictive: the explicit symbols
    const uint64_t size = r.varint();
    GREQ(r.ok && size <= (uint64_t)(r.n - r.pos), 119);
    sym_bits = r.p + r.pos; sym_nbytes = (uint32_t)size;
    r.skip(size);
  }
  Rabs start_face, seams[DSA_MAX_ATT_DATA];
  uint32_t endp;
  start_face.start(s, L.stream_len, r.pos, &endp);
  GREQ(start_face.ok, 201);
  r.pos = endp;
  for (uint32_t i = 0; i < nad; ++i) {                     // decoded by phase 2 (mesh_seams), which has the whole wave
    D->gen_seam_pos[i] = r.pos;
    seams[i].start(s, L.stream_len, r.pos, &endp);
    GREQ(seams[i].ok, 260);
    r.pos = endp;
  }
  Rabs prediction;
  int predicted_symbol = -1;
  if (predictive) {                                        // MeshEdgeBreakerTraversalPredictiveDecoder.cs:19-27
    const int32_t nss = (int32_t)r.u32();
    GREQ(r.ok && nss >= 0 && (uint32_t)nss < VMAX, 645);
    for (uint32_t v = 0; v < VMAX; ++v) valences[v] = 0;
    prediction.start(s, L.stream_len, r.pos, &endp);
    GREQ(prediction.ok, 646);
    r.pos = endp;
  }
  uint32_t ctx_off[6] = {0, 0, 0, 0, 0, 0};
  int32_t ctx_cnt[6] = {0, 0, 0, 0, 0, 0};
  if (valence) {
    for (uint32_t v = 0; v < VMAX; ++v) valences[v] = 0;
    uint32_t total = 0;
    for (int i = 0; i < 6; ++i) {
      const uint64_t num = r.varint();
      GREQ(r.ok && num <= F && total + num <= F, 641);
      ctx_off[i] = total;
      if (num > 0) {
        if (!decode_symbols(D, r, (uint32_t)num, 1, ctx_syms + total, rs)) return false;
        ctx_cnt[i] = (int32_t)num;
      }
      total += (uint32_t)num;
    }
  }
  // ---------------------------------------------------------------- symbols, :232-442
  const uint64_t t1 = gclk();
  int last_symbol = -1, active_context = -1;
  uint32_t sp = 0, num_faces = 0, num_invalid = 0, splits_left = S;
  const bool remove_invalid = nad == 0;
  // What the next symbol needs is almost always what the last one made: the active corner is the top of the stack
  // (kept in `top`), and its face is the newest one, whose three vertices stay in `lv` (any other corner goes to
  // memory).  With that a C symbol is two dependent round trips (left-most corner of its pivot, then the vertex
  // behind it) and R / L / E none, instead of one per table the reference touches.  Valence traversal: the next
  // symbol of each of the six context lists is fetched ahead (`nxt`), so choosing the list costs no round trip.
  uint32_t top = DSA_INVALID, lf = DSA_INVALID, lv[3] = {DSA_INVALID, DSA_INVALID, DSA_INVALID};
#define G_VTX(c_) (((c_) < C && (c_) / 3 == lf) ? lv[(c_) % 3] : ct.vertex(c_))
#define G_FACE(f_, v0_, v1_, v2_) { ct.c2v[3 * (f_)] = (v0_); ct.c2v[3 * (f_) + 1] = (v1_); ct.c2v[3 * (f_) + 2] = (v2_); lf = (f_); lv[0] = (v0_); lv[1] = (v1_); lv[2] = (v2_); }
  uint64_t win = 0, win_base = 0, win_end = 0;            // window over the explicit symbol bits
  uint32_t n_links = 0;                                    // "corner already has an opposite" is not looked up per symbol: every link sets
                                                           // two corners, so phase 2 counts the linked corners and compares (link census)
  uint32_t nxt[6] = {0, 0, 0, 0, 0, 0};
  if (valence) for (int i = 0; i < 6; ++i) if (ctx_cnt[i] > 0) nxt[i] = ctx_syms[ctx_off[i] + ctx_cnt[i] - 1];
  for (uint32_t sid = 0; sid < num_symbols; ++sid) {
    const uint32_t face = num_faces++;
    bool check_split = false;
    uint32_t sym;
    if (predictive && predicted_symbol != -1 && prediction.next() != 0) {   // ...PredictiveDecoder.cs:34-46: predicted and confirmed
      sym = (uint32_t)predicted_symbol;
      last_symbol = predicted_symbol;
    } else if (!valence) {                                 // MeshEdgeBreakerTraversalDecoder.cs:89-99
      // three bits at most, from a 64-bit window over the symbol bytes (refilled every twenty symbols or so)
      if (sym_bitpos + 3 > win_end) {
        const uint64_t byte = sym_bitpos >> 3;
        win = 0;
        if (byte + 8 <= sym_nbytes) __builtin_memcpy(&win, sym_bits + byte, 8);
        else for (uint64_t k = byte; k < sym_nbytes; ++k) win |= (uint64_t)sym_bits[k] << (8 * (k - byte));
        win_base = byte * 8; win_end = win_base + 64;
      }
      const uint32_t three = (uint32_t)(win >> (sym_bitpos - win_base)) & 7u;
      sym = (three & 1u) ? three : 0u;
      sym_bitpos += (three & 1u) ? 3 : 1;
      GREQ(sym_bitpos <= (uint64_t)sym_nbytes * 8, 246);
      last_symbol = (int)sym;
    } else {                                               // ...ValenceDecoder.cs:77-98
      if (active_context != -1) {
        const int32_t cnt = --ctx_cnt[active_context];
        GREQ(cnt >= 0, 642);
        const uint32_t id = nxt[active_context];
        if (cnt > 0) nxt[active_context] = ctx_syms[ctx_off[active_context] + cnt - 1];   // not needed before this list's next turn
        GREQ(id <= 4, 643);
        last_symbol = id == 0 ? 0 : (id == 1 ? 1 : (id == 2 ? 3 : (id == 3 ? 5 : 7)));
      } else last_symbol = 7;
      sym = (uint32_t)last_symbol;
    }
    const uint32_t corner = 3 * face;
    if (sym == 0) {                    // C
      GREQ(sp > 0, 210);
      const uint32_t ca = top;
      const uint32_t vx = G_VTX(cnx(ca));
      GREQ(vx < ct.nv, 213);
      const uint32_t lm = ct.left_most(vx);
      GREQ(lm < C, 213);
      const uint32_t cb = cnx(lm);
      GREQ(ca != cb && ca < C, 213);
      ct.set_opp(ca, corner + 1);
      ct.set_opp(cb, corner + 2);
      n_links += 2;
      const uint32_t va_prev = G_VTX(cpv(ca)), vb_next = G_VTX(cnx(cb));
      GREQ(va_prev < ct.nv && vb_next < ct.nv && vx != va_prev && vx != vb_next, 213);
      G_FACE(face, vx, vb_next, va_prev);
      ct.vcorner[va_prev] = corner + 2;
      is_hole[vx] = 0;
      top = corner;
    } else if (sym == 5 || sym == 3) { // R / L
      GREQ(sp > 0, 220);
      const uint32_t ca = top;
      GREQ(ca < C, 263);
      ++n_links;
      uint32_t oc, cr;
      if (sym == 5) { oc = corner + 2; cr = corner; }
      else { oc = corner + 1; cr = corner + 2; }
      ct.set_opp(oc, ca);
      GREQ(ct.nv < VMAX, 220);
      const uint32_t nv = ct.nv++;
      ct.vcorner[nv] = oc;
      const uint32_t vr = G_VTX(cpv(ca)), vl = G_VTX(cnx(ca));
      GREQ(vr < ct.nv && vl < ct.nv, 220);
      if (sym == 5) { G_FACE(face, vr, vl, nv); } else { G_FACE(face, vl, nv, vr); }
      ct.vcorner[vr] = cr;
      top = corner;
      check_split = true;
    } else if (sym == 1) {             // S
      GREQ(sp > 0, 230);
      const uint32_t cb = top;
      --sp;
      if (S > 0 && active[sid] != DSA_INVALID) { GREQ(sp < F, 232); stack[sp++] = active[sid]; }
      GREQ(sp > 0, 232);
      const uint32_t ca = stack[sp - 1];
      GREQ(ca != cb && ca < C && cb < C, 233);
      ct.set_opp(ca, corner + 2);
      ct.set_opp(cb, corner + 1);
      n_links += 2;
      const uint32_t vp = ct.vertex(cpv(ca)), vq = ct.vertex(cnx(ca)), vb_prev = ct.vertex(cpv(cb));
      GREQ(vp < ct.nv && vq < ct.nv && vb_prev < ct.nv, 234);
      G_FACE(face, vp, vq, vb_prev);
      ct.vcorner[vb_prev] = corner + 2;
      uint32_t cn = cnx(cb);
      const uint32_t vn = ct.vertex(cn);
      GREQ(vn < ct.nv, 234);
      if (valence || predictive) valences[vp] += valences[vn];   // ...ValenceDecoder.cs:151-154 / ...PredictiveDecoder MergeVertices
      ct.vcorner[vp] = ct.left_most(vn);
      const uint32_t first = cn;
      uint32_t guard = 0;
      while (cn != DSA_INVALID) {
        ct.c2v[cn] = vp;
        cn = ct.swing_left(cn);
        GREQ(cn != first && ++guard <= C, 235);
      }
      ct.vcorner[vn] = DSA_INVALID;
      if (remove_invalid) { GREQ(num_invalid < VMAX, 236); invalid_list[num_invalid++] = vn; }
      for (int k = 0; k < 3; ++k) lv[k] = ct.c2v[corner + k];         // the ring walk above may have relabelled the new face's own corners
      top = corner;
    } else if (sym == 7) {             // E
      GREQ(ct.nv + 3 <= VMAX && sp < F, 240);
      const uint32_t v0 = ct.nv;
      ct.nv += 3;
      G_FACE(face, v0, v0 + 1, v0 + 2);
      ct.vcorner[v0] = corner; ct.vcorner[v0 + 1] = corner + 1; ct.vcorner[v0 + 2] = corner + 2;
      if (sp > 0) stack[sp - 1] = top;                      // the old top goes to memory only when something is pushed over it
      ++sp;
      top = corner;
      check_split = true;
    } else GFAIL(241);
    if (valence || predictive) {       // NewActiveCornerReached, ...ValenceDecoder.cs:100-149 / ...PredictiveDecoder.cs:48-92
      const uint32_t a = lv[0], b = lv[1], c = lv[2];        // the active corner is corner 0 of the newest face
      GREQ(a < VMAX && b < VMAX && c < VMAX, 644);
      switch (last_symbol) {
        case 0: case 1: valences[b] += 1; valences[c] += 1; break;
        case 5: valences[a] += 1; valences[b] += 1; valences[c] += 2; break;
        case 3: valences[a] += 1; valences[b] += 2; valences[c] += 1; break;
        case 7: valences[a] += 2; valences[b] += 2; valences[c] += 2; break;
        default: break;
      }
      const int32_t v = (int32_t)valences[b];
      active_context = (v < 2 ? 2 : (v > 7 ? 7 : v)) - 2;
      predicted_symbol = (last_symbol == 0 || last_symbol == 5) ? (v < 6 ? 5 : 0) : -1;
    }
    if (check_split) {                 // :363-375, IsTopologySplit :450-471
      const int32_t enc_id = (int32_t)(num_symbols - sid - 1);
      while (splits_left > 0) {
        const uint32_t src = splits[3 * (splits_left - 1)];
        GREQ((int64_t)src <= (int64_t)enc_id, 243);       // encoderSplitSymbolId < 0 in the reference
        if ((int64_t)src != (int64_t)enc_id) break;
        const uint32_t edge = splits[3 * (splits_left - 1) + 2], enc_split = splits[3 * (splits_left - 1) + 1];
        --splits_left;
        GREQ(enc_split < num_symbols, 244);
        const uint32_t nc = edge == 1 ? cnx(top) : cpv(top);   // 1 = right face edge
        const uint32_t key = num_symbols - enc_split - 1;
        active[key] = nc;                                      // dictionary semantics: overwrite
      }
    }
  }
#undef G_VTX
#undef G_FACE
  if (sp > 0) stack[sp - 1] = top;
  // start faces, :378-415
  const uint64_t t2 = gclk();
  while (sp > 0) {
    const uint32_t corner = stack[--sp];
    const bool interior = start_face.next() != 0;
    if (!interior) continue;
    GREQ(num_faces < F && corner < C, 251);
    const uint32_t ca = corner;
    const uint32_t vn = ct.vertex(cnx(ca));
    GREQ(vn < ct.nv && ct.left_most(vn) < C, 252);
    const uint32_t cb = cnx(ct.left_most(vn));
    const uint32_t vx = ct.vertex(cnx(cb));
    GREQ(vx < ct.nv && ct.left_most(vx) < C, 253);
    const uint32_t cc = cnx(ct.left_most(vx));
    GREQ(ca != cb && ca != cc && cb != cc, 254);
    GREQ(ct.opp[ca] == DSA_INVALID && ct.opp[cb] == DSA_INVALID && ct.opp[cc] == DSA_INVALID, 255);
    const uint32_t vp = ct.vertex(cnx(cc));
    GREQ(vp < ct.nv, 262);
    const uint32_t face = num_faces++, nc = 3 * face;
    ct.set_opp(nc, ca); ct.set_opp(nc + 1, cb); ct.set_opp(nc + 2, cc);
    n_links += 3;
    ct.c2v[nc] = vx; ct.c2v[nc + 1] = vp; ct.c2v[nc + 2] = vn;
    is_hole[vx] = 0; is_hole[vp] = 0; is_hole[vn] = 0;
  }
  GREQ(num_faces == F, 256);
  // isolated-vertex compaction, :417-441 (D-10)
  uint32_t num_vertices = ct.nv;
  for (uint32_t k = 0; k < num_invalid; ++k) {
    const uint32_t inv = invalid_list[k];
    GREQ(num_vertices > 0, 257);
    uint32_t src = num_vertices - 1;
    while (ct.left_most(src) == DSA_INVALID) { GREQ(num_vertices > 1, 258); src = --num_vertices - 1; }
    if (src < inv) continue;
    const uint32_t start = ct.left_most(src);
    uint32_t c = start, guard = 0;
    bool left = true;
    while (c != DSA_INVALID) {
      GREQ(c < C && ct.c2v[c] == src && ++guard <= C, 259);
      ct.c2v[c] = inv;
      if (left) {
        c = ct.swing_left(c);
        if (c == DSA_INVALID) { c = ct.swing_right(start); left = false; }
        else if (c == start) c = DSA_INVALID;
      } else c = ct.swing_right(c);
    }
    ct.vcorner[inv] = ct.left_most(src);
    ct.vcorner[src] = DSA_INVALID;
    is_hole[inv] = is_hole[src];
    is_hole[src] = 0;
    num_vertices--;
  }
  const uint32_t num_conn_vertices = num_vertices;
  D->num_vertices = num_conn_vertices;
  D->num_all_vertices = ct.nv;
  D->interior_corners = 2 * n_links;                       // checked against the corner table by phase 2
  const uint64_t t3 = gclk();
  D->end_pos = r.pos;
  const uint64_t t4 = gclk();     // diagnostics (tools/dbg_phases.py): header + tables, symbols, start faces + compaction, seams
  D->dbg[0] = (uint32_t)(t1 - t0); D->dbg[1] = (uint32_t)(t2 - t1); D->dbg[2] = (uint32_t)(t3 - t2); D->dbg[3] = (uint32_t)(t4 - t3);
  return true;
}

// Phase 2a (cooperative): attribute seams, MeshEdgeBreakerDecoder.cs:502-535.  One bit per interior edge and attribute
// data, in corner order (an edge belongs to the lower of its two faces); boundary edges are seams of every attribute
// data.  Only the rABS recurrences are serial (lane d decodes stream d into a byte per edge); which corner owns the
// k-th bit is a prefix sum, and marking is idempotent stores.  Scratch: the c2v / v2lm regions of the attribute data
// blocks, which phase 2b fills afterwards.
__device__ __forceinline__ bool mesh_seams(uint8_t *arena, const MeshLayout &L, MeshDesc *D, MeshCtx &m) {
  const Ct &ct = m.ct;
  Act *act = m.act;
  const uint32_t C = m.C, nad = m.nad, VMAX = m.VMAX, lane = g_lane();
  if (nad == 0) return true;
  for (uint32_t d = 0; d < nad; ++d) {
    for (uint32_t c = lane; c < C; c += G_NL) act[d].edge_seam[c] = 0;
    for (uint32_t v = lane; v < VMAX; v += G_NL) act[d].vert_seam[v] = 0;
  }
  uint32_t *kidx = act[0].v2lm;                            // corner -> index of its bit
  uint32_t base = 0;
  for (uint32_t c0 = 0; c0 < C; c0 += G_NL) {
    const uint32_t c = c0 + lane;
    bool owns = false;
    if (c < C) { const uint32_t oc = ct.opp[c]; owns = oc != DSA_INVALID && oc / 3 >= c / 3; }
    uint32_t total;
    const uint32_t k = base + g_excl_scan(owns ? 1u : 0u, &total);
    if (owns) kidx[c] = k;
    base += total;
  }
  const uint32_t num_bits = base;
  bool bad = false;
  for (uint32_t d = lane; d < nad; d += G_NL) {
    Rabs rd;
    uint32_t endp;
    rd.start(arena + L.stream, L.stream_len, D->gen_seam_pos[d], &endp);
    if (!rd.ok) { bad = true; continue; }
    uint8_t *bits = (uint8_t *)act[d].c2v;
    for (uint32_t k = 0; k < num_bits; ++k) bits[k] = (uint8_t)rd.next();
  }
  if (g_any(bad)) { if (lane == 0) fail(D, ST_INVALID, 260); return false; }
#if defined(__HIPCC__)
  __threadfence_block();
#endif
  for (uint32_t c = lane; c < C; c += G_NL) {
    const uint32_t oc = ct.opp[c];
    if (oc == DSA_INVALID) { for (uint32_t d = 0; d < nad; ++d) act[d].add_seam_edge(c); continue; }
    if (oc / 3 < c / 3) continue;
    const uint32_t k = kidx[c];
    for (uint32_t d = 0; d < nad; ++d) if (((const uint8_t *)act[d].c2v)[k]) act[d].add_seam_edge(c);
  }
#if defined(__HIPCC__)
  __threadfence_block();
#endif
  for (uint32_t d = 0; d < nad; ++d)
    for (uint32_t c = lane; c < C; c += G_NL) act[d].c2v[c] = DSA_INVALID;
  return true;
}

// Phase 2 (cooperative: every lane owns vertices): attribute vertices per corner (RecomputeVertices,
// MeshAttributeCornerTable.cs:95-155) and points per corner (AssignPointsToCorners, MeshEdgeBreakerDecoder.cs:537-638).
// Both number things vertex by vertex and, inside a vertex, in ring order; the counts of a chunk of vertices are
// prefix-summed, so every lane can write the ids of its own vertex.
__device__ __forceinline__ bool mesh_tables(uint8_t *arena, const MeshLayout &L, MeshDesc *D) {
  MeshCtx m;
  if (!mesh_ctx(arena, L, D, m)) return false;
  const Ct &ct = m.ct;
  const uint32_t C = m.C, nad = m.nad, lane = g_lane();
  {   // link census (MeshEdgeBreakerDecoder.cs "corner already has an opposite"): a corner linked twice leaves a stale
      // link on its first partner, so fewer corners carry a link than phase 1 made
    uint32_t linked = 0;
    for (uint32_t c0 = 0; c0 < C; c0 += G_NL) {
      const uint32_t c = c0 + lane;
      uint32_t total;
      (void)g_excl_scan((c < C && ct.opp[c] != DSA_INVALID) ? 1u : 0u, &total);
      linked += total;
    }
    if (linked != D->interior_corners) { if (lane == 0) fail(D, ST_INVALID, 263); return false; }
  }
  if (!mesh_seams(arena, L, D, m)) return false;
  for (uint32_t d = 0; d < nad; ++d) {
    Act &A = m.act[d];
    uint32_t base = 0;
    for (uint32_t v0 = 0; v0 < ct.nv; v0 += G_NL) {
      const uint32_t v = v0 + lane;
      uint32_t cnt = 0, first_c = DSA_INVALID;
      bool bad = false;
      if (v < ct.nv) {
        const uint32_t c = ct.left_most(v);
        if (c != DSA_INVALID) {
          if (c >= C) bad = true;
          else {
            first_c = c;
            if (A.vert_seam[v]) {
              uint32_t a = A.swing_left(first_c), guard = 0;
              while (a != DSA_INVALID) {
                first_c = a;
                a = A.swing_left(a);
                if (a == c || ++guard > C) { bad = true; break; }
              }
            }
            cnt = 1;
            uint32_t a = ct.swing_right(first_c), guard = 0;
            while (!bad && a != DSA_INVALID && a != first_c) {
              if (++guard > C) { bad = true; break; }
              if (A.edge_seam[cnx(a)]) ++cnt;
              a = ct.swing_right(a);
            }
          }
        }
      }
      if (g_any(bad)) { if (lane == 0) fail(D, ST_INVALID, 651); return false; }
      uint32_t total;
      uint32_t id = base + g_excl_scan(cnt, &total);
      if (g_any(cnt && id + cnt > C)) { if (lane == 0) fail(D, ST_INVALID, 650); return false; }
      if (cnt) {
        A.c2v[first_c] = id;
        A.v2lm[id] = first_c;
        uint32_t a = ct.swing_right(first_c);
        while (a != DSA_INVALID && a != first_c) {
          if (A.edge_seam[cnx(a)]) { ++id; A.v2lm[id] = a; }
          A.c2v[a] = id;
          a = ct.swing_right(a);
        }
      }
      base += total;
    }
    A.nv = base;
    if (lane == 0) D->gen_act_nv[d] = base;
  }
  // ---------------------------------------------------------------- AssignPointsToCorners, :537-638
  int32_t *c2p = m.c2p;
  const uint8_t *is_hole = m.is_hole;
  if (nad == 0) {
    for (uint32_t c = lane; c < C; c += G_NL) c2p[c] = (int32_t)ct.c2v[c];
    if (lane == 0) D->num_points = D->num_vertices;
    return true;
  }
  for (uint32_t c = lane; c < C; c += G_NL) c2p[c] = 0;
  uint32_t base = 0;
  for (uint32_t v0 = 0; v0 < ct.nv; v0 += G_NL) {
    const uint32_t v = v0 + lane;
    uint32_t cnt = 0, dedup_first = DSA_INVALID;
    bool bad = false;
    if (v < ct.nv) {
      const uint32_t c = ct.left_most(v);
      if (c != DSA_INVALID) {
        if (c >= C) bad = true;
        else {
          dedup_first = c;
          if (!is_hole[v]) {
            for (uint32_t d = 0; d < nad && !bad; ++d) {
              const uint32_t vv = ct.c2v[c];
              if (vv >= m.VMAX || !m.act[d].vert_seam[vv]) continue;           // IsCornerOnSeam
              const uint32_t vid = m.act[d].vertex(c);
              uint32_t a = ct.swing_right(c), guard = 0;
              bool seam_found = false;
              while (a != c) {
                if (a == DSA_INVALID || ++guard > C) { bad = true; break; }
                if (m.act[d].vertex(a) != vid) { dedup_first = a; seam_found = true; break; }
                a = ct.swing_right(a);
              }
              if (seam_found) break;
            }
          }
          cnt = 1;
          uint32_t prev_c = dedup_first, a = ct.swing_right(dedup_first), guard = 0;
          while (!bad && a != DSA_INVALID && a != dedup_first) {
            if (a >= C || ++guard > C) { bad = true; break; }
            for (uint32_t d = 0; d < nad; ++d) if (m.act[d].vertex(a) != m.act[d].vertex(prev_c)) { ++cnt; break; }
            prev_c = a;
            a = ct.swing_right(a);
          }
        }
      }
    }
    if (g_any(bad)) { if (lane == 0) fail(D, ST_INVALID, 661); return false; }
    uint32_t total;
    uint32_t id = base + g_excl_scan(cnt, &total);
    if (g_any(cnt && id + cnt > L.cap_points)) { if (lane == 0) fail(D, ST_INVALID, 662); return false; }
    if (cnt) {
      c2p[dedup_first] = (int32_t)id;
      uint32_t prev_c = dedup_first, a = ct.swing_right(dedup_first);
      while (a != DSA_INVALID && a != dedup_first) {
        bool seam = false;
        for (uint32_t d = 0; d < nad; ++d) if (m.act[d].vertex(a) != m.act[d].vertex(prev_c)) { seam = true; break; }
        if (seam) ++id;
        c2p[a] = (int32_t)id;
        prev_c = a;
        a = ct.swing_right(a);
      }
    }
    base += total;
  }
  if (lane == 0) D->num_points = base;
  // Boundary flag of every vertex of every table a depth-first traversal may walk (DepthFirstTraverser.cs:47 asks it for
  // each new vertex: two dependent table reads there, one byte here).  Position table: in the is_hole region (done
  // with); attribute tables: in the orientation scratch of their block (used only by the values stage).
#if defined(__HIPCC__)
  __threadfence_block();
#endif
  {
    uint8_t *bnd = m.is_hole;
    for (uint32_t v = lane; v < ct.nv; v += G_NL) bnd[v] = ct.is_on_boundary(v) ? 1 : 0;
    for (uint32_t d = 0; d < nad; ++d) {
      uint8_t *ba = m.G + m.g.data + (uint64_t)d * m.g.data_stride + m.g.orient;
      for (uint32_t v = lane; v < m.act[d].nv; v += G_NL) ba[v] = m.act[d].is_on_boundary(v) ? 1 : 0;
    }
  }
  return true;
}

// Phase 3: the attribute section -- per decoder the traversal order, point maps, values -- in three stages, so that
// the element-parallel one has the whole wave: 0 (one lane) traversal orders, 1 (cooperative) point maps, 2 (one lane)
// values.  Every stage re-reads the few bytes of decoder triples and descriptors.
enum { ATT_SEQUENCE = 0, ATT_MAPS = 1, ATT_VALUES = 2, ATT_NORMALS = 3 };
__device__ __forceinline__ bool mesh_attributes(uint8_t *arena, const MeshLayout &L, MeshDesc *D, RansScratch rs, int stage) {
  MeshCtx m;
  if (!mesh_ctx(arena, L, D, m)) return false;
  if (stage == ATT_SEQUENCE) D->off_attributes = D->end_pos;
  Rd r(arena + L.stream, L.stream_len, D->off_attributes);
  const bool cooperative = stage == ATT_MAPS || stage == ATT_NORMALS;
  const uint32_t lane = cooperative ? g_lane() : 0, nl = cooperative ? G_NL : 1;
  const GenLayout &g = m.g;
  uint8_t *G = m.G;
  Ct &ct = m.ct;
  Act *act = m.act;
  int32_t *c2p = m.c2p;
  const uint32_t F = m.F, C = m.C, VMAX = m.VMAX, nad = m.nad, num_points = D->num_points;
  rs.cum = (uint32_t *)(G + g.cum); rs.cum_cap = g.cum_entries;
  // ---------------------------------------------------------------- attribute section, ConnectivityDecoder.cs:16-44
  const uint32_t ndec = r.u8();
  GREQ(r.ok, 122);
  if (ndec > DSA_MAX_ATT) GNOTIMPL(122);
  D->num_decoders = ndec;
  DecoderInfo dec[DSA_MAX_ATT];
  int data_decoder[DSA_MAX_ATT_DATA];
  bool data_conn_used[DSA_MAX_ATT_DATA];
  for (uint32_t d = 0; d < DSA_MAX_ATT_DATA; ++d) { data_decoder[d] = -1; data_conn_used[d] = true; }
  bool pos_seen = false;
  for (uint32_t i = 0; i < ndec; ++i) {                   // MeshEdgeBreakerDecoder.cs:640-708
    dec[i].att_data_id = (int8_t)r.u8();
    dec[i].element_type = r.u8();
    const uint32_t traversal_method = r.u8();
    GREQ(r.ok && traversal_method < 2, 123);                // (any element type but "vertex" is a corner attribute, :666-701)
    if (dec[i].att_data_id >= 0) {
      GREQ((uint32_t)dec[i].att_data_id < nad && data_decoder[dec[i].att_data_id] < 0, 124);
      data_decoder[dec[i].att_data_id] = (int)i;
    } else { GREQ(!pos_seen, 125); pos_seen = true; }
    if (dec[i].element_type == 0) { if (dec[i].att_data_id >= 0) data_conn_used[dec[i].att_data_id] = false; }
    else GREQ(traversal_method == 0 && dec[i].att_data_id >= 0, 126);
    dec[i].traversal_method = traversal_method;           // 1: prediction degree (vertex attributes only, checked above)
  }
  uint32_t natt = 0;
  for (uint32_t i = 0; i < ndec; ++i) if (!decode_descriptors(L, D, r, i, natt, &dec[i].first_att, &dec[i].num_atts)) return false;
  D->num_attributes = natt;
  // ---------------------------------------------------------------- per decoder: sequence, values
  uint8_t *fvis = G + g.fvis, *vvis = G + g.vvis;
  uint32_t *dfs = (uint32_t *)(G + g.dfs);
  const uint32_t NVMAX = m.NVMAX;
  struct Shared { uint32_t *d2c, *pids, entries; int32_t *v2d; const uint32_t *map; } shared[2] = {{nullptr, nullptr, 0, nullptr, nullptr}, {nullptr, nullptr, 0, nullptr, nullptr}};
  uint64_t acc_trav = 0, acc_map = 0, acc_val = 0;
  for (uint32_t i = 0; i < ndec; ++i) {
    const uint64_t ta = gclk();
    // MeshTraversalSequencer + DepthFirstTraverser on the decoder's corner table
    uint32_t *d2c, *pids;
    int32_t *v2d;
    uint32_t cap_entries, nverts;
    const int dd = dec[i].att_data_id;
    uint32_t *para;                                       // operands on the table the decoder's prediction schemes use
    if (dd < 0) {
      d2c = (uint32_t *)(arena + L.d2c); v2d = (int32_t *)(arena + L.v2d); pids = (uint32_t *)(arena + L.vrank);
      para = (uint32_t *)(arena + L.para);
      cap_entries = VMAX; nverts = ct.nv;
    } else {
      uint8_t *blk = G + g.data + (uint64_t)dd * g.data_stride;
      d2c = (uint32_t *)(blk + g.d2c); v2d = (int32_t *)(blk + g.v2d); pids = (uint32_t *)(blk + g.pids);
      para = (uint32_t *)(blk + g.para);
      cap_entries = NVMAX; nverts = act[dd].nv > ct.nv ? act[dd].nv : ct.nv;
    }
    const bool use_act = dd >= 0 && data_conn_used[dd];
    uint32_t entries = 0;
    const bool corner_att = dec[i].element_type != 0;
    // Every vertex-type decoder traverses the position corner table from the same start: the order, the maps and
    // the entry -> point list are those of the first one, so they are computed once and shared.
    const uint32_t *map_src = nullptr;
    Shared &sh = shared[dec[i].traversal_method];       // one set per traversal method (MeshEdgeBreakerDecoder.cs:681-690)
    const bool first_of_kind = corner_att || !sh.d2c;
    if (!first_of_kind) { d2c = sh.d2c; v2d = sh.v2d; pids = sh.pids; entries = sh.entries; map_src = sh.map; }
    else {
      if (stage != ATT_SEQUENCE) entries = D->gen_dec_entries[i];
      else if (corner_att) {
        if (!traverse(D, act[dd], nverts, c2p, fvis, vvis, dfs, F + 1, d2c, v2d, pids, cap_entries, &entries,
                      G + g.data + (uint64_t)dd * g.data_stride + g.orient, act[dd].nv)) return false;
      }
      else if (dec[i].traversal_method == 1) {
        if (!traverse_prediction_degree(D, ct, nverts, c2p, fvis, vvis, (uint32_t *)(G + g.pd_next), (uint32_t *)(G + g.pd_degree), d2c, v2d, pids, cap_entries, &entries)) return false;
      } else if (!traverse(D, ct, nverts, c2p, fvis, vvis, dfs, F + 1, d2c, v2d, pids, cap_entries, &entries, m.is_hole, ct.nv)) return false;
      if (!corner_att) { sh.d2c = d2c; sh.v2d = v2d; sh.pids = pids; sh.entries = entries; }
    }
    dec[i].num_entries = entries;
    if (stage == ATT_SEQUENCE) { D->gen_dec_entries[i] = entries; if (dd < 0) D->num_entries = entries; }
    const uint64_t tb = gclk();
    // point -> entry map of every attribute of the decoder, MeshTraversalSequencer.cs:33-50.  All corners of a point
    // carry the same attribute vertex (that is how phase 2 numbered the points), so the stores of different lanes to
    // one point agree.
    for (uint32_t ai = dec[i].first_att; ai < dec[i].first_att + dec[i].num_atts; ++ai) {
      uint32_t *map = (uint32_t *)(arena + L.map[ai]);
      if (stage == ATT_MAPS) {
        if (map_src) { for (uint32_t p = lane; p < num_points; p += nl) map[p] = map_src[p]; }
        else {
          for (uint32_t p = lane; p < num_points; p += nl) map[p] = 0;
#if defined(__HIPCC__)
          __threadfence_block();
#endif
          for (uint32_t c = lane; c < C; c += nl) {
            const uint32_t point = (uint32_t)c2p[c];
            const uint32_t v = corner_att ? act[dd].vertex(c) : ct.vertex(c);
            GREQ(v < nverts && point < num_points, 670);
            const int32_t e = v2d[v];
            GREQ(e >= 0 && (uint32_t)e < num_points, 671);
            map[point] = (uint32_t)e;
          }
#if defined(__HIPCC__)
          __threadfence_block();
#endif
        }
      }
      if (!map_src) {
        map_src = map;                    // further attributes of this decoder share the map
        if (!corner_att && !sh.map) sh.map = map;
      }
    }
    if (stage == ATT_MAPS) {
      if (use_act) parallelogram_operands(act[dd], d2c, v2d, nverts, entries, para, lane, nl);
      else parallelogram_operands(ct, d2c, v2d, nverts, entries, para, lane, nl);
    }
    const uint64_t tc = gclk();
    if (stage == ATT_SEQUENCE || stage == ATT_MAPS) { acc_trav += tb - ta; acc_map += tc - tb; continue; }
    // values, then transform parameters, of every attribute of the decoder (SequentialAttributeDecodersController.cs:29-38,
    // AttributesDecoder.cs:65-70)
    ValueCtx vc;
    vc.ct = &ct; vc.act = use_act ? &act[dd] : nullptr;
    vc.para_ct = use_act ? nullptr : para; vc.para_act = use_act ? para : nullptr;
    vc.d2c = d2c; vc.v2d = v2d; vc.pids = pids;
    // orientation scratch: the attribute data block's, or (attributes of the position decoder) the Edgebreaker
    // machine's corner stack, which is free by now
    vc.orient = dd >= 0 ? G + g.data + (uint64_t)dd * g.data_stride + g.orient : G + g.stack;
    vc.orient_cap = dd >= 0 ? NVMAX : 4 * F; vc.num_points = num_points;
    vc.num_verts = nverts; vc.num_corners = C;
    if (stage == ATT_NORMALS) { if (!normals_stage(arena, L, D, dec[i].first_att, dec[i].num_atts, entries, vc, lane, nl)) return false; continue; }
    for (uint32_t ai = dec[i].first_att; ai < dec[i].first_att + dec[i].num_atts; ++ai) if (!decode_values(arena, L, D, r, ai, entries, rs, vc)) return false;
    for (uint32_t ai = dec[i].first_att; ai < dec[i].first_att + dec[i].num_atts; ++ai) if (!decode_transform_params(D, r, ai)) return false;
    acc_trav += tb - ta; acc_map += tc - tb; acc_val += gclk() - tc;
  }
  // diagnostics: traversals, point maps, values
  if (stage == ATT_SEQUENCE) D->dbg[7] = (uint32_t)acc_trav;
  else if (stage == ATT_MAPS) { if (lane == 0) D->dbg[8] = (uint32_t)acc_map; }
  else if (stage == ATT_VALUES) { D->end_pos = r.pos; D->dbg[9] = (uint32_t)acc_val; }
  return true;
}


// The whole mesh in one go (host check; the device runs the three phases as separate kernels so that the second
// one can use the whole wave).
__device__ __forceinline__ bool decode_mesh(uint8_t *arena, const MeshLayout &L, MeshDesc *D, Rd &r, RansScratch rs) {
  return mesh_connectivity(arena, L, D, r, rs) && mesh_tables(arena, L, D) && mesh_attributes(arena, L, D, rs, ATT_SEQUENCE) &&
         mesh_attributes(arena, L, D, rs, ATT_MAPS) && mesh_attributes(arena, L, D, rs, ATT_VALUES) && mesh_attributes(arena, L, D, rs, ATT_NORMALS);
}

#undef GFAIL
#undef GREQ
#undef GNOTIMPL
}  // namespace gen

#if defined(__HIPCC__)
// The general path as three kernels on the third stream, beside the fast kernels (which skip general meshes):
//   k_general             one lane per mesh: connectivity + seam bits (sequential meshes: the whole decode)
//   k_general_tables      one wave per mesh, all lanes: attribute corner tables, points per corner
//   k_general_attributes  three launches: traversal orders (one lane per mesh), point maps (whole wave), attribute values (one lane)
__global__ __launch_bounds__(WAVE) void k_general(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  __shared__ uint16_t sh_lut[GEN_LUT_SLOTS / 2];          // half-resolution tables (8 KB): the entropy decode here is the small part
  __shared__ uint16_t sh_cum[GEN_LUT_SYMS + 1];
  const uint32_t mesh = blockIdx.x;
  if (mesh >= n || threadIdx.x != 0) return;
  MeshDesc *D = &descs[mesh];
  if (!D->general || status_of(D) != ST_OK) return;
  const MeshLayout &L = layouts[mesh];
  Rd r(arena + L.stream, L.stream_len, D->end_pos);       // k_locate parked the reader behind the header
  gen::RansScratch rs = {sh_lut, 1, sh_cum, nullptr, nullptr, 0};
  if (D->encoder_method == 0) (void)gen::decode_sequential_mesh(arena, L, D, r, rs);
  else (void)gen::mesh_connectivity(arena, L, D, r, rs);
}
__global__ __launch_bounds__(WAVE) void k_general_tables(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  const uint32_t mesh = blockIdx.x;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (!D->general || D->encoder_method == 0 || status_of(D) != ST_OK) return;
  (void)gen::mesh_tables(arena, layouts[mesh], D);
}
// One instantiation per stage: each gets the registers its own code needs (the values stage carries every predictor
// and wants ~180 VGPRs = 8 blocks per CU; its CROWDED variant is held to 128 so that 16 blocks per CU = a batch of 4096
// meshes are resident at once, which is worth the spills only when the batch is that large).
template <int STAGE, bool CROWDED>
__device__ __forceinline__ void general_attributes_body(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n);
template <int STAGE>
__global__ __launch_bounds__(WAVE) void k_general_attributes(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  general_attributes_body<STAGE, false>(arena, layouts, descs, n);
}
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_general_values_crowded(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  general_attributes_body<gen::ATT_VALUES, true>(arena, layouts, descs, n);
}
template <int STAGE, bool CROWDED>
__device__ __forceinline__ void general_attributes_body(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  __shared__ uint16_t sh_lut[STAGE == gen::ATT_VALUES ? (CROWDED ? GEN_LUT_SLOTS / 2 : GEN_LUT_SLOTS) : 1];
  __shared__ uint16_t sh_cum16[STAGE == gen::ATT_VALUES && CROWDED ? GEN_LUT_SYMS + 1 : 1];
  __shared__ uint32_t sh_cum32[STAGE == gen::ATT_VALUES && !CROWDED ? GEN_LUT_SYMS + 1 : 1];
  const uint32_t mesh = blockIdx.x;
  if (mesh >= n || (STAGE != gen::ATT_MAPS && STAGE != gen::ATT_NORMALS && threadIdx.x != 0)) return;
  MeshDesc *D = &descs[mesh];
  if (!D->general || D->encoder_method == 0 || status_of(D) != ST_OK) return;
  gen::RansScratch rs = {sh_lut, CROWDED ? 1u : 0u, sh_cum16, sh_cum32, nullptr, 0};
  (void)gen::mesh_attributes(arena, layouts[mesh], D, rs, STAGE);
}
#endif

}  // namespace dsa
