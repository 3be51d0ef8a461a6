// draco-sharp_amd/csrc/dsa_encode_host.h
// -----------------------------------------------------------------------------
// Host half of the encode direction (BASELINE.json configs[4]) and of the synthetic
// input generator: corner table, Edgebreaker connectivity encoder, DFS sequencing, the stream layout of a
// Draco v2.2 mesh, symbol-scheme selection and rANS table construction, plus a complete CPU attribute coder.
//   * draco-sharp_amd/synth (tests / bench inputs) uses all of it, CPU only;
//   * the product encoder (dsa_encode.hip) uses the connectivity + layout + table parts and runs
//     quantisation, prediction, symbol statistics and the rANS coding on the GPU; its output is byte-identical
//     to the CPU coder's, which is what tests/test_gpu_encode.py checks.
// Follows the rules of SURVEY.md Appendix D:
//   IO/Mesh/MeshEdgeBreakerEncoder.cs:38-124,176-303  (connectivity traversal)
//   IO/Mesh/MeshEdgeBreakerTraversalEncoder.cs:40-71  (symbol/start-face/seam sections)
//   IO/Entropy/SymbolEncoding.cs:8-193, RAnsSymbolEncoder.cs:15-184,
//   RAnsEncoder.cs:22-30, AnsEncoder.cs:19-64, BitCoders/RAnsBitEncoder.cs
//   IO/Attributes/SequentialIntegerAttributeEncoder.cs:55-128,
//   AttributeQuantizationTransform.cs:66-177, OctahedronToolBox.cs:28-119,
//   PredictionSchemes/*Encoder.cs, *EncodingTransform.cs
// (each with the defects listed there, E-1..E-8, corrected to the bitstream's semantics).
// -----------------------------------------------------------------------------
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dsa_symbol_plan.h"

namespace synth {

static const uint32_t kInvalid = 0xFFFFFFFFu;

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };
static inline void check(bool ok, const char *msg) { if (!ok) throw Fail(msg); }

// ------------------------------------------------------------------ writers
struct ByteWriter {
  std::vector<uint8_t> d;
  void u8(uint8_t v) { d.push_back(v); }
  void i8(int8_t v) { d.push_back((uint8_t)v); }
  void u16(uint16_t v) { u8(v & 0xFF); u8(v >> 8); }
  void u32(uint32_t v) { for (int i = 0; i < 4; ++i) u8((v >> (8 * i)) & 0xFF); }
  void i32(int32_t v) { u32((uint32_t)v); }
  void f32(float f) { uint32_t v; memcpy(&v, &f, 4); u32(v); }
  void varint(uint64_t v) { while (v >= 0x80) { u8((uint8_t)(v | 0x80)); v >>= 7; } u8((uint8_t)v); }
  void bytes(const std::vector<uint8_t> &b) { d.insert(d.end(), b.begin(), b.end()); }
};
struct BitWriter {   // LSB-first, EncoderBuffer.cs bit mode
  std::vector<uint8_t> d;
  uint64_t nbits = 0;
  void put(int count, uint32_t v) {       // the low `count` bits of v, least significant first; d always ends in the byte being filled
    if (count <= 0) return;
    uint64_t x = count >= 32 ? (uint64_t)v : (uint64_t)(v & ((1u << count) - 1u));
    const int off = (int)(nbits & 7);
    if (off) {
      d.back() |= (uint8_t)(x << off);
      const int took = 8 - off;
      if (count <= took) { nbits += (uint64_t)count; return; }
      x >>= took; count -= took; nbits += (uint64_t)took;
    }
    while (count > 0) { d.push_back((uint8_t)x); x >>= 8; const int t = count < 8 ? count : 8; nbits += (uint64_t)t; count -= t; }
  }
};

static inline int msb(uint32_t v) { int r = 0; while (v >>= 1) ++r; return r; }
static inline uint32_t zigzag(int32_t v) { return v >= 0 ? (uint32_t)v << 1 : (((uint32_t)(-(v + 1))) << 1) | 1; }

// ------------------------------------------------------------------ entropy
// rABS bit block: prob_zero u8, size varint, bytes (RAnsBitEncoder.cs:25-44,86-126)
static void write_rabs(ByteWriter &w, const std::vector<uint8_t> &bits) {
  uint64_t zeros = 0;
  for (uint8_t b : bits) zeros += b ? 0 : 1;
  uint64_t total = bits.size() ? bits.size() : 1;
  uint32_t raw = (uint32_t)(((double)zeros / (double)total) * 256.0 + 0.5);
  uint8_t p0 = 255;
  if (raw < 255) p0 = (uint8_t)raw;
  if (p0 == 0) p0 = 1;
  std::vector<uint8_t> buf;
  buf.reserve(bits.size() / 8 + 16);
  uint32_t state = 4096;
  const uint32_t p = 256u - p0;            // 1 .. 255, like p0
  // state / ls by a multiply-high with ceil(2^32 / ls): the quotient or one more (state < 2^32), one correction
  const uint32_t ls2[2] = {p0, p}, lim2[2] = {16u * 256u * p0, 16u * 256u * p}, add2[2] = {p, 0u};
  const uint64_t magic2[2] = {0xFFFFFFFFull / p0 + 1ull, 0xFFFFFFFFull / p + 1ull};
  for (size_t k = bits.size(); k-- > 0;) {
    const int val = bits[k] != 0;
    const uint32_t ls = ls2[val];
    if (state >= lim2[val]) { buf.push_back(state & 0xFF); state >>= 8; }
    uint32_t quot = (uint32_t)(((uint64_t)state * magic2[val]) >> 32);
    uint32_t rem = state - quot * ls;
    if ((int32_t)rem < 0) { --quot; rem += ls; }
    state = quot * 256 + rem + add2[val];
  }
  uint32_t s = state - 4096;   // AnsEncoder.cs:34-64
  if (s < (1u << 6)) buf.push_back((uint8_t)s);
  else if (s < (1u << 14)) { uint32_t v = 0x4000 + s; buf.push_back(v & 0xFF); buf.push_back(v >> 8); }
  else if (s < (1u << 22)) { uint32_t v = 0x800000 + s; buf.push_back(v & 0xFF); buf.push_back((v >> 8) & 0xFF); buf.push_back(v >> 16); }
  else check(false, "rABS state too large");
  w.u8(p0);
  w.varint(buf.size());
  w.bytes(buf);
}

struct RansEncoder {
  int precision_bits = 12;
  uint32_t precision = 4096, l_base = 16384;
  std::vector<uint32_t> prob, cum;
  uint32_t num_symbols = 0;
  // RAnsSymbolEncoder.cs:15-123 (normalisation) + :125-164 (table bytes, E-3 corrected)
  void create(ByteWriter &w, int max_bit_length, const std::vector<uint64_t> &freq) {
    prob.assign(freq.size() ? freq.size() : 1, 0);
    cum.assign(prob.size(), 0);
    std::vector<uint32_t> order(prob.size()), tmp(prob.size());
    const int rc = dsa::plan::rans_tables(max_bit_length, freq.data(), freq.size(), prob.data(), cum.data(), order.data(), tmp.data(), &precision_bits, &num_symbols);
    check(rc == dsa::plan::PLAN_OK, dsa::plan::plan_message(rc));
    prob.resize(num_symbols); cum.resize(num_symbols);
    precision = 1u << precision_bits;
    l_base = precision * 4;
    write_table(w);
  }
  // RAnsSymbolEncoder.cs:125-164 (E-3 corrected): the probability table as the stream carries it
  void write_table(ByteWriter &w) const {
    w.varint(num_symbols);
    for (uint32_t i = 0; i < num_symbols; ++i) {
      uint32_t pr = prob[i];
      int extra = 0;
      if (pr >= (1u << 6)) { extra++; if (pr >= (1u << 14)) { extra++; check(pr < (1u << 22), "probability too large"); } }
      if (pr == 0) {
        uint32_t offset = 0;
        for (; offset < 63; ++offset) if (prob[i + offset + 1] > 0) break;
        w.u8((uint8_t)((offset << 2) | 3));
        i += offset;
      } else {
        w.u8((uint8_t)((pr << 2) | (uint32_t)extra));
        for (int b = 0; b < extra; ++b) w.u8((uint8_t)(pr >> (8 * (b + 1) - 2)));
      }
    }
  }
  // symbols fed last->first (SymbolEncoding.cs:177-183); RAnsEncoder.cs:22-30; AnsEncoder.cs:34-64
  void encode(ByteWriter &w, const uint32_t *syms, size_t n, size_t stride = 1) {
    std::vector<uint8_t> buf;
    buf.reserve(n);
    uint32_t state = l_base;
    for (size_t k = n; k-- > 0;) {
      uint32_t s = syms[k * stride];
      uint32_t p = prob[s];
      uint64_t lim = (uint64_t)(l_base / precision) * 256u * p;
      while ((uint64_t)state >= lim) { buf.push_back(state & 0xFF); state >>= 8; }
      state = (state / p) * precision + state % p + cum[s];
    }
    uint32_t s = state - l_base;
    if (s < (1u << 6)) buf.push_back((uint8_t)s);
    else if (s < (1u << 14)) { uint32_t v = 0x4000 + s; buf.push_back(v & 0xFF); buf.push_back(v >> 8); }
    else if (s < (1u << 22)) { uint32_t v = 0x800000 + s; buf.push_back(v & 0xFF); buf.push_back((v >> 8) & 0xFF); buf.push_back(v >> 16); }
    else if (s < (1u << 30)) { uint32_t v = 0xC0000000u + s; for (int i = 0; i < 4; ++i) buf.push_back((v >> (8 * i)) & 0xFF); }
    else check(false, "rANS state too large");
    w.varint(buf.size());
    w.bytes(buf);
  }
};

// Statistics of one symbol stream: everything the scheme selection and the table construction need.  The CPU coder
// fills it from the symbol list; the GPU encoder fills it with histograms computed on the device.
struct SymbolStats {
  size_t n = 0;                       // symbols (entries * nc)
  int nc = 1;
  uint32_t max_value = 0;
  uint64_t total_bl = 0;              // sum of the per-entry bit lengths
  std::vector<uint64_t> tag_freq;     // [33] histogram of per-entry bit lengths
  std::vector<uint64_t> raw_freq;     // [max_value + 1] histogram of symbol values
};
static void symbol_stats(const std::vector<uint32_t> &v, int nc, SymbolStats &st, std::vector<uint32_t> *bit_lengths_out = nullptr) {
  st.n = v.size(); st.nc = nc; st.max_value = 0; st.total_bl = 0;
  st.tag_freq.assign(33, 0);
  std::vector<uint32_t> bl;
  bl.reserve(v.size() / nc);
  for (size_t i = 0; i < v.size(); i += nc) {
    uint32_t mc = v[i];
    for (int j = 1; j < nc; ++j) mc = std::max(mc, v[i + j]);
    const uint32_t b = (uint32_t)(mc > 0 ? msb(mc) : 0) + 1;
    st.max_value = std::max(st.max_value, mc);
    bl.push_back(b);
    st.total_bl += b;
    ++st.tag_freq[b];
  }
  st.raw_freq.assign((size_t)st.max_value + 1, 0);
  for (uint32_t x : v) ++st.raw_freq[x];
  if (bit_lengths_out) bit_lengths_out->swap(bl);
}
// SymbolEncoding.cs:8-40 (E-2 corrected): scheme choice, then the coder's tables.  `head` receives the bytes that
// precede the rANS payload: scheme byte, (raw: unique-symbols bit length), probability table.
struct SymbolPlan {
  int method = 1;                     // 0 tagged, 1 raw
  RansEncoder coder;                  // tagged: over bit lengths; raw: over symbol values
  ByteWriter head;
};
static void plan_symbols(const SymbolStats &st, int force_scheme, int compression_level, SymbolPlan &pl) {
  int usbl = 0;
  const int rc = dsa::plan::choose_scheme(st.tag_freq.data(), st.raw_freq.data(), st.max_value, (uint64_t)st.n, (uint32_t)st.nc, st.total_bl,
                                          force_scheme, compression_level, &pl.method, &usbl);
  check(rc == dsa::plan::PLAN_OK, dsa::plan::plan_message(rc));
  pl.head.u8((uint8_t)pl.method);
  if (pl.method == 0) {
    pl.coder.create(pl.head, 5, st.tag_freq);
  } else {
    pl.head.u8((uint8_t)usbl);
    pl.coder.create(pl.head, usbl, st.raw_freq);
  }
}

// SymbolEncoding.cs:92-137 tagged, :139-193 raw -- the CPU coder
static void encode_symbols(ByteWriter &w, const std::vector<uint32_t> &v, int nc, int force_scheme, int compression_level) {
  if (v.empty()) return;
  SymbolStats st;
  std::vector<uint32_t> bit_lengths;
  symbol_stats(v, nc, st, &bit_lengths);
  SymbolPlan pl;
  plan_symbols(st, force_scheme, compression_level, pl);
  w.bytes(pl.head.d);
  if (pl.method == 0) {
    pl.coder.encode(w, bit_lengths.data(), bit_lengths.size());
    BitWriter bw;
    for (size_t e = 0; e < bit_lengths.size(); ++e)
      for (int c = 0; c < nc; ++c) bw.put((int)bit_lengths[e], v[e * nc + c]);
    w.bytes(bw.d);
  } else {
    pl.coder.encode(w, v.data(), v.size());
  }
}

// -------------------------------------------------------------- corner table
struct CornerTable {
  std::vector<uint32_t> opp, c2v, vcorner;
  uint32_t nf() const { return (uint32_t)(c2v.size() / 3); }
  uint32_t nc() const { return (uint32_t)c2v.size(); }
  uint32_t nv() const { return (uint32_t)vcorner.size(); }
  static uint32_t next(uint32_t c) { return c == kInvalid ? c : ((c + 1) % 3 ? c + 1 : c - 2); }
  static uint32_t prev(uint32_t c) { return c == kInvalid ? c : (c % 3 ? c - 1 : c + 2); }
  uint32_t opposite(uint32_t c) const { return c == kInvalid ? c : opp[c]; }
  uint32_t vertex(uint32_t c) const { return c == kInvalid ? kInvalid : c2v[c]; }
  uint32_t swing_right(uint32_t c) const { return prev(opposite(prev(c))); }
  uint32_t swing_left(uint32_t c) const { return next(opposite(next(c))); }
  uint32_t right_corner(uint32_t c) const { return opposite(next(c)); }
  uint32_t left_corner(uint32_t c) const { return opposite(prev(c)); }
  bool on_boundary(uint32_t v) const { return swing_left(vcorner[v]) == kInvalid; }

  void build(const uint32_t *faces, uint32_t num_faces, uint32_t num_vertices) {
    c2v.assign(faces, faces + (size_t)num_faces * 3);
    opp.assign((size_t)num_faces * 3, kInvalid);
    vcorner.assign(num_vertices, kInvalid);
    // half-edge matching: corner c is opposite to edge (next(c) -> prev(c))
    std::vector<std::pair<uint64_t, uint32_t>> edges;
    edges.reserve(c2v.size());
    for (uint32_t c = 0; c < nc(); ++c) {
      uint64_t a = c2v[next(c)], b = c2v[prev(c)];
      check(a != b, "degenerate face in input mesh");
      edges.push_back({(a << 32) | b, c});
    }
    std::sort(edges.begin(), edges.end());
    for (size_t i = 1; i < edges.size(); ++i) check(edges[i].first != edges[i - 1].first, "non-manifold edge (duplicate half-edge)");
    for (auto &e : edges) {
      uint64_t a = e.first >> 32, b = e.first & 0xFFFFFFFFu;
      uint64_t rev = (b << 32) | a;
      auto it = std::lower_bound(edges.begin(), edges.end(), std::make_pair(rev, (uint32_t)0));
      if (it != edges.end() && it->first == rev) opp[e.second] = it->second;
    }
    for (uint32_t c = 0; c < nc(); ++c) if (vcorner[c2v[c]] == kInvalid) vcorner[c2v[c]] = c;
    // left-most corner for boundary vertices (CornerTable.cs UpdateVertexToCornerMap)
    for (uint32_t v = 0; v < num_vertices; ++v) {
      uint32_t first = vcorner[v];
      if (first == kInvalid) continue;
      uint32_t act = swing_left(first), c = first;
      size_t guard = 0;
      while (act != kInvalid && act != first) { c = act; act = swing_left(act); check(++guard < c2v.size(), "vertex ring does not close"); }
      if (act != first) vcorner[v] = c;
    }
    // manifold vertex check: every corner of v must be reachable from its left-most corner
    std::vector<uint32_t> count(num_vertices, 0), reach(num_vertices, 0);
    for (uint32_t c = 0; c < nc(); ++c) ++count[c2v[c]];
    for (uint32_t v = 0; v < num_vertices; ++v) {
      uint32_t s = vcorner[v];
      if (s == kInvalid) continue;
      uint32_t c = s;
      do { ++reach[v]; c = swing_right(c); } while (c != kInvalid && c != s && reach[v] <= count[v]);
      check(reach[v] == count[v], "non-manifold vertex in input mesh");
    }
  }
};

// The connectivity of one attribute with seams (MeshAttributeCornerTable.cs, encoder side): the position corner
// table with every seam edge cut.  Built from a value id per corner (:32-78: an interior edge is a seam when either of
// its end points carries different value ids on its two faces); vertices are the fans between cuts (:95-155).  Same
// interface as CornerTable, so that the traversal and the prediction schemes are templates over either.
struct AttrConn {
  const CornerTable *ct = nullptr;
  std::vector<uint8_t> edge_seam, vert_seam;
  std::vector<uint32_t> c2v, v2lm;
  bool no_interior_seams = true;
  uint32_t nf() const { return ct->nf(); }
  uint32_t nc() const { return ct->nc(); }
  uint32_t nv() const { return (uint32_t)v2lm.size(); }
  static uint32_t next(uint32_t c) { return CornerTable::next(c); }
  static uint32_t prev(uint32_t c) { return CornerTable::prev(c); }
  uint32_t opposite(uint32_t c) const { return (c == kInvalid || edge_seam[c]) ? kInvalid : ct->opposite(c); }
  uint32_t vertex(uint32_t c) const { return c == kInvalid ? kInvalid : c2v[c]; }
  uint32_t swing_right(uint32_t c) const { return prev(opposite(prev(c))); }
  uint32_t swing_left(uint32_t c) const { return next(opposite(next(c))); }
  uint32_t right_corner(uint32_t c) const { return opposite(next(c)); }
  uint32_t left_corner(uint32_t c) const { return opposite(prev(c)); }
  bool on_boundary(uint32_t v) const { return swing_left(v2lm[v]) == kInvalid; }
  void mark(uint32_t c) {
    edge_seam[c] = 1;
    vert_seam[ct->vertex(next(c))] = 1;
    vert_seam[ct->vertex(prev(c))] = 1;
  }
  void build(const CornerTable &t, const uint32_t *corner_value) {
    ct = &t;
    edge_seam.assign(t.nc(), 0);
    vert_seam.assign(t.nv(), 0);
    c2v.assign(t.nc(), kInvalid);
    no_interior_seams = true;
    for (uint32_t c = 0; c < t.nc(); ++c) {
      const uint32_t o = t.opposite(c);
      if (o == kInvalid) { mark(c); continue; }
      if (o < c) continue;
      // the edge's end points: next(c) lies on prev(o), prev(c) on next(o)
      if (corner_value[next(c)] != corner_value[prev(o)] || corner_value[prev(c)] != corner_value[next(o)]) {
        no_interior_seams = false;
        mark(c); mark(o);
      }
    }
    // :95-155 vertices: around every position vertex a new one behind every cut
    v2lm.clear();
    for (uint32_t v = 0; v < t.nv(); ++v) {
      uint32_t first_c = t.vcorner[v];
      if (first_c == kInvalid) continue;
      if (vert_seam[v]) {
        uint32_t act = swing_left(first_c);
        size_t guard = 0;
        while (act != kInvalid) { first_c = act; act = swing_left(act); check(++guard <= t.nc(), "attribute seam loop"); }
      }
      uint32_t id = (uint32_t)v2lm.size();
      c2v[first_c] = id;
      v2lm.push_back(first_c);
      uint32_t act = t.swing_right(first_c);
      while (act != kInvalid && act != first_c) {
        if (edge_seam[next(act)]) { id = (uint32_t)v2lm.size(); v2lm.push_back(act); }
        c2v[act] = id;
        act = t.swing_right(act);
      }
    }
  }
};

// DFS traversal shared by attribute sequencing (Traverser/DepthFirstTraverser.cs:9-99)
struct Sequence {
  std::vector<uint32_t> data_to_corner;   // entry -> corner (source corner table)
  std::vector<int32_t> vertex_to_data;
};
template <class CT>
static void dfs_sequence(const CT &ct, const std::vector<uint32_t> &corner_order, Sequence &seq) {
  std::vector<uint8_t> fvis(ct.nf(), 0), vvis(ct.nv(), 0);
  seq.vertex_to_data.assign(ct.nv(), -1);
  seq.data_to_corner.clear();
  std::vector<uint32_t> stack;
  auto visit = [&](uint32_t v, uint32_t c) { vvis[v] = 1; seq.vertex_to_data[v] = (int32_t)seq.data_to_corner.size(); seq.data_to_corner.push_back(c); };
  auto fdone = [&](uint32_t f) { return f == kInvalid || fvis[f]; };
  for (uint32_t start : corner_order) {
    if (fdone(start / 3)) continue;
    stack.clear();
    stack.push_back(start);
    uint32_t nv = ct.vertex(CT::next(start)), pv = ct.vertex(CT::prev(start));
    if (!vvis[nv]) visit(nv, CT::next(start));
    if (!vvis[pv]) visit(pv, CT::prev(start));
    while (!stack.empty()) {
      uint32_t corner = stack.back();
      uint32_t face = corner == kInvalid ? kInvalid : corner / 3;
      if (corner == kInvalid || fdone(face)) { stack.pop_back(); continue; }
      for (;;) {
        fvis[face] = 1;
        uint32_t v = ct.vertex(corner);
        if (!vvis[v]) {
          bool ob = ct.on_boundary(v);
          visit(v, corner);
          if (!ob) { corner = ct.right_corner(corner); face = corner / 3; continue; }
        }
        uint32_t rc = ct.right_corner(corner), lc = ct.left_corner(corner);
        uint32_t rf = rc == kInvalid ? kInvalid : rc / 3, lf = lc == kInvalid ? kInvalid : lc / 3;
        if (fdone(rf)) {
          if (fdone(lf)) { stack.pop_back(); break; }
          corner = lc; face = lf;
        } else {
          if (fdone(lf)) { corner = rc; face = rf; }
          else { stack.back() = lc; stack.push_back(rc); break; }
        }
      }
    }
  }
}

// MaxPredictionDegreeTraverser.cs:22-152 (with the degree list sized, as in the bitstream): faces whose tip vertex is
// already coded first, then tips seen from two faces, then the rest.
static void prediction_degree_sequence(const CornerTable &ct, const std::vector<uint32_t> &corner_order, Sequence &seq) {
  std::vector<uint8_t> fvis(ct.nf(), 0), vvis(ct.nv(), 0);
  std::vector<uint32_t> degree(ct.nv(), 0), stacks[3];
  seq.vertex_to_data.assign(ct.nv(), -1);
  seq.data_to_corner.clear();
  int best = 0;
  auto visit = [&](uint32_t v, uint32_t c) { vvis[v] = 1; seq.vertex_to_data[v] = (int32_t)seq.data_to_corner.size(); seq.data_to_corner.push_back(c); };
  auto fdone = [&](uint32_t c) { return c == kInvalid || fvis[c / 3]; };
  auto pop = [&]() {
    for (int i = best; i < 3; ++i) if (!stacks[i].empty()) { uint32_t c = stacks[i].back(); stacks[i].pop_back(); best = i; return c; }
    return kInvalid;
  };
  auto push = [&](uint32_t c, int pr) { stacks[pr].push_back(c); if (pr < best) best = pr; };
  auto priority = [&](uint32_t c) { uint32_t v = ct.vertex(c); if (vvis[v]) return 0; return ++degree[v] > 1 ? 1 : 2; };
  for (uint32_t start : corner_order) {
    stacks[0].push_back(start);
    best = 0;
    uint32_t nv = ct.vertex(CornerTable::next(start)), pv = ct.vertex(CornerTable::prev(start)), tv = ct.vertex(start);
    if (!vvis[nv]) visit(nv, CornerTable::next(start));
    if (!vvis[pv]) visit(pv, CornerTable::prev(start));
    if (!vvis[tv]) visit(tv, start);
    uint32_t corner;
    while ((corner = pop()) != kInvalid) {
      if (fdone(corner)) continue;
      for (;;) {
        fvis[corner / 3] = 1;
        uint32_t v = ct.vertex(corner);
        if (!vvis[v]) visit(v, corner);
        uint32_t rc = ct.right_corner(corner), lc = ct.left_corner(corner);
        const bool rdone = fdone(rc), ldone = fdone(lc);
        if (!ldone) { int pr = priority(lc); if (rdone && pr <= best) { corner = lc; continue; } push(lc, pr); }
        if (!rdone) { int pr = priority(rc); if (pr <= best) { corner = rc; continue; } push(rc, pr); }
        break;
      }
    }
  }
}

// ------------------------------------------------- Edgebreaker connectivity
struct EbResult {
  std::vector<uint8_t> symbols;               // encoder order, bit patterns (0,1,3,5,7)
  std::vector<uint8_t> start_face_bits;
  std::vector<uint32_t> processed_corners;    // decoder face order
  struct Split { uint32_t source, split, edge; };
  std::vector<Split> splits;
  uint32_t num_split_symbols = 0;
  std::vector<uint32_t> face_time;            // face -> index of the first symbol coded with the face already visited
};

struct EbEncoder {
  const CornerTable &ct;
  EbResult &r;
  std::vector<uint8_t> visited_faces, visited_verts, visited_holes;
  std::vector<int32_t> vertex_hole_id;
  std::map<int, int> face_to_split_symbol;
  int last_symbol_id = -1;
  EbEncoder(const CornerTable &t, EbResult &res) : ct(t), r(res) {}

  void find_holes() {     // MeshEdgeBreakerEncoder.cs:331-361
    for (uint32_t i = 0; i < ct.nc(); ++i) {
      if (ct.opposite(i) != kInvalid) continue;
      uint32_t bv = ct.vertex(CornerTable::next(i));
      if (vertex_hole_id[bv] != -1) continue;
      int id = (int)visited_holes.size();
      visited_holes.push_back(0);
      uint32_t c = i;
      while (vertex_hole_id[bv] == -1) {
        vertex_hole_id[bv] = id;
        c = CornerTable::next(c);
        while (ct.opposite(c) != kInvalid) c = CornerTable::next(ct.opposite(c));
        bv = ct.vertex(CornerTable::next(c));
      }
    }
  }
  bool find_init_face(uint32_t face, uint32_t *out) {   // :158-183
    uint32_t corner = 3 * face;
    for (int i = 0; i < 3; ++i) {
      if (ct.opposite(corner) == kInvalid) { *out = corner; return false; }
      if (vertex_hole_id[ct.vertex(corner)] != -1) {
        uint32_t rc = corner;
        while (rc != kInvalid) { corner = rc; rc = ct.swing_right(rc); }
        *out = CornerTable::prev(corner);
        return false;
      }
      corner = CornerTable::next(corner);
    }
    *out = corner;
    return true;
  }
  void encode_hole(uint32_t start_corner, bool encode_first) {   // :276-303
    uint32_t c = CornerTable::prev(start_corner);
    while (ct.opposite(c) != kInvalid) c = CornerTable::next(ct.opposite(c));
    uint32_t start_v = ct.vertex(start_corner);
    if (encode_first) visited_verts[start_v] = 1;
    visited_holes[vertex_hole_id[start_v]] = 1;
    uint32_t act = ct.vertex(CornerTable::prev(c));
    while (act != start_v) {
      visited_verts[act] = 1;
      c = CornerTable::next(c);
      while (ct.opposite(c) != kInvalid) c = CornerTable::next(ct.opposite(c));
      act = ct.vertex(CornerTable::prev(c));
    }
  }
  bool right_visited(uint32_t c) const { uint32_t o = ct.opposite(CornerTable::next(c)); return o == kInvalid || visited_faces[o / 3]; }
  bool left_visited(uint32_t c) const { uint32_t o = ct.opposite(CornerTable::prev(c)); return o == kInvalid || visited_faces[o / 3]; }
  void check_split(int src_symbol, uint32_t edge, uint32_t neighbor_face) {   // :373-390
    auto it = face_to_split_symbol.find((int)neighbor_face);
    if (it == face_to_split_symbol.end()) return;
    r.splits.push_back({(uint32_t)src_symbol, (uint32_t)it->second, edge});
  }
  void encode_from_corner(uint32_t corner) {   // :185-274
    std::vector<uint32_t> stack{corner};
    while (!stack.empty()) {
      corner = stack.back();
      if (corner == kInvalid || visited_faces[corner / 3]) { stack.pop_back(); continue; }
      for (;;) {
        ++last_symbol_id;
        uint32_t face = corner / 3;
        visited_faces[face] = 1;
        r.face_time[face] = (uint32_t)last_symbol_id;
        r.processed_corners.push_back(corner);
        uint32_t v = ct.vertex(corner);
        bool on_boundary = vertex_hole_id[v] != -1;
        if (!visited_verts[v]) {
          visited_verts[v] = 1;
          if (!on_boundary) { r.symbols.push_back(0); corner = ct.right_corner(corner); continue; }
        }
        uint32_t rc = ct.right_corner(corner), lc = ct.left_corner(corner);
        uint32_t rf = rc == kInvalid ? kInvalid : rc / 3, lf = lc == kInvalid ? kInvalid : lc / 3;
        if (right_visited(corner)) {
          if (rf != kInvalid) check_split(last_symbol_id, 1, rf);
          if (left_visited(corner)) {
            if (lf != kInvalid) check_split(last_symbol_id, 0, lf);
            r.symbols.push_back(7);
            stack.pop_back();
            break;
          }
          r.symbols.push_back(5);
          corner = lc;
        } else {
          if (left_visited(corner)) {
            if (lf != kInvalid) check_split(last_symbol_id, 0, lf);
            r.symbols.push_back(3);
            corner = rc;
          } else {
            r.symbols.push_back(1);
            ++r.num_split_symbols;
            if (on_boundary) {
              int hid = vertex_hole_id[v];
              if (!visited_holes[hid]) encode_hole(corner, false);
            }
            face_to_split_symbol[(int)face] = last_symbol_id;
            stack.back() = lc;
            stack.push_back(rc);
            break;
          }
        }
      }
    }
  }
  void run() {   // MeshEdgeBreakerEncoder.cs:38-124
    visited_faces.assign(ct.nf(), 0);
    r.face_time.assign(ct.nf(), kInvalid);
    visited_verts.assign(ct.nv(), 0);
    vertex_hole_id.assign(ct.nv(), -1);
    find_holes();
    std::vector<uint32_t> init_corners;
    for (uint32_t c = 0; c < ct.nc(); ++c) {
      uint32_t face = c / 3;
      if (visited_faces[face]) continue;
      uint32_t start;
      bool interior = find_init_face(face, &start);
      r.start_face_bits.push_back(interior ? 1 : 0);
      if (interior) {
        visited_verts[ct.vertex(start)] = 1;
        visited_verts[ct.vertex(CornerTable::next(start))] = 1;
        visited_verts[ct.vertex(CornerTable::prev(start))] = 1;
        visited_faces[face] = 1;
        r.face_time[face] = (uint32_t)(last_symbol_id + 1);
        init_corners.push_back(CornerTable::next(start));
        uint32_t o = ct.opposite(CornerTable::next(start));
        if (o != kInvalid && !visited_faces[o / 3]) encode_from_corner(o);
      } else {
        encode_hole(CornerTable::next(start), true);
        encode_from_corner(start);
      }
    }
    std::reverse(r.processed_corners.begin(), r.processed_corners.end());
    r.processed_corners.insert(r.processed_corners.end(), init_corners.begin(), init_corners.end());
  }
};

// Predictive Edgebreaker traversal, encoder side (the reference has only the decoder,
// MeshEdgeBreakerTraversalPredictiveDecoder.cs:19-93; this follows the format's encoder): valences of the not yet
// coded part of the mesh are what the decoder will have built when it gets there.  Before a C or R symbol the
// previous symbol (the decoder's next) is predicted from the valence of the pivot: a hit costs one bit, a miss one
// bit and the symbol.  The tip of a split face is poisoned (two decoder vertices until the S merges them).
static void predictive_symbols(const CornerTable &ct, const EbResult &eb, std::vector<uint8_t> &explicit_symbols, std::vector<uint8_t> &predictions) {
  const size_t n = eb.symbols.size();
  std::vector<int32_t> valence(ct.nv(), 0);
  for (uint32_t c = 0; c < ct.nc(); ++c) {                 // edges around a vertex: faces, +1 on a boundary
    valence[ct.vertex(c)] += 1;
    if (ct.opposite(CornerTable::prev(c)) == kInvalid) valence[ct.vertex(c)] += 1;   // the boundary edge leaving the fan on this side
  }
  explicit_symbols.clear();
  predictions.clear();
  int prev_symbol = -1;
  for (size_t i = 0; i < n; ++i) {
    const uint8_t symbol = eb.symbols[i];
    const uint32_t corner = eb.processed_corners[n - 1 - i], next = CornerTable::next(corner), prev = CornerTable::prev(corner);
    const uint32_t a = ct.vertex(corner), b = ct.vertex(next), c = ct.vertex(prev);
    auto predict = [&](uint32_t pivot) { const int32_t v = valence[pivot]; return v < 0 ? -2 : (v < 6 ? 5 : 0); };
    int predicted = -1;
    switch (symbol) {
      case 0: predicted = predict(b); valence[b] -= 1; valence[c] -= 1; break;
      case 1: valence[b] -= 1; valence[c] -= 1; valence[a] = -1; break;
      case 5: predicted = predict(b); valence[a] -= 1; valence[b] -= 1; valence[c] -= 2; break;
      case 3: valence[a] -= 1; valence[b] -= 2; valence[c] -= 1; break;
      default: valence[a] -= 2; valence[b] -= 2; valence[c] -= 2; break;
    }
    bool store_prev = true;
    if (predicted != -1) {
      if (predicted == prev_symbol) { predictions.push_back(1); store_prev = false; }
      else if (prev_symbol != -1) predictions.push_back(0);
    }
    if (store_prev && prev_symbol != -1) explicit_symbols.push_back((uint8_t)prev_symbol);
    prev_symbol = symbol;
  }
  if (prev_symbol != -1) explicit_symbols.push_back((uint8_t)prev_symbol);
}

// Valence Edgebreaker traversal, encoder side (decoder: MeshEdgeBreakerTraversalValenceDecoder.cs:22-154): every
// symbol but the last coded one goes into one of six lists chosen by the valence, in the not yet coded part of the
// mesh, of the vertex the decoder will stand on when it reads it.  A split face's tip is two decoder vertices until
// the S merges them: the encoder splits it the same way (faces left of the split keep the vertex, faces right of it
// get a new one).  Symbol ids: C 0, S 1, L 2, R 3, E 4 (Constants.cs:88-95).
static void valence_context_symbols(const CornerTable &ct, const EbResult &eb, std::vector<uint32_t> ctx[6]) {
  const size_t n = eb.symbols.size();
  std::vector<int32_t> valence(ct.nv(), 0);
  for (uint32_t c = 0; c < ct.nc(); ++c) {
    valence[ct.vertex(c)] += 1;
    if (ct.opposite(CornerTable::prev(c)) == kInvalid) valence[ct.vertex(c)] += 1;
  }
  std::vector<uint32_t> c2v(ct.nc());
  for (uint32_t c = 0; c < ct.nc(); ++c) c2v[c] = ct.vertex(c);
  for (int i = 0; i < 6; ++i) ctx[i].clear();
  int prev_symbol = -1;
  for (size_t i = 0; i < n; ++i) {
    const uint8_t symbol = eb.symbols[i];
    const uint32_t corner = eb.processed_corners[n - 1 - i], next = CornerTable::next(corner), prev = CornerTable::prev(corner);
    auto coded = [&](uint32_t c) { return eb.face_time[c / 3] <= i; };
    const int32_t active_valence = valence[c2v[next]];
    switch (symbol) {
      case 0: valence[c2v[next]] -= 1; valence[c2v[prev]] -= 1; break;
      case 1: {
        valence[c2v[next]] -= 1; valence[c2v[prev]] -= 1;
        int left = 0, right = 0;
        uint32_t a = ct.opposite(prev);
        while (a != kInvalid && !coded(a)) { ++left; a = ct.opposite(CornerTable::next(a)); }
        valence[c2v[corner]] = left + 1;
        const uint32_t nv = (uint32_t)valence.size();
        a = ct.opposite(next);
        while (a != kInvalid && !coded(a)) { ++right; c2v[CornerTable::next(a)] = nv; a = ct.opposite(CornerTable::prev(a)); }
        valence.push_back(right + 1);
        break;
      }
      case 5: valence[c2v[corner]] -= 1; valence[c2v[next]] -= 1; valence[c2v[prev]] -= 2; break;
      case 3: valence[c2v[corner]] -= 1; valence[c2v[next]] -= 2; valence[c2v[prev]] -= 1; break;
      default: valence[c2v[corner]] -= 2; valence[c2v[next]] -= 2; valence[c2v[prev]] -= 2; break;
    }
    if (prev_symbol != -1) {
      const int clamped = active_valence < 2 ? 2 : (active_valence > 7 ? 7 : active_valence);
      static const uint32_t id_of[8] = {0, 1, 0, 2, 0, 3, 0, 4};
      ctx[clamped - 2].push_back(id_of[prev_symbol]);
    }
    prev_symbol = symbol;
  }
}

// ----------------------------------------------------------------- options
struct Options {
  int32_t pos_bits = 11, uv_bits = 10, normal_bits = 8;
  int32_t single_connectivity = 0;   // split_mesh_on_seams=false at speed 5 -> per-attribute connectivity
  int32_t force_scheme = -1;         // -1 auto, 0 tagged, 1 raw
  int32_t compression_level = 5;     // 10 - speed
  int32_t pos_prediction = 1;        // 1 parallelogram, 0 difference
  int32_t uv_prediction = 1;
  int32_t generic_u8 = 0;            // add a per-vertex uint8 generic attribute (Integer decoder) when generic data is given
  int32_t normal_prediction = 0;     // 0 difference, 6 geometric normal (the CPU coder only)
  int32_t predictive_connectivity = 0;   // 1: predictive Edgebreaker traversal (deprecated in the format); 2: valence traversal, what stock
                                         // encoders write at their default level (CPU coder only)
  int32_t traversal_method = 0;      // attribute sequencing: 0 depth first; 1 prediction degree for the decoder of the positions
                                     // (what stock encoders do at their highest level); 2 prediction degree for every decoder (CPU coder only)
  // Decoder branches no stock encoder setting reaches (CPU coder only; tests and tools/soak.py):
  int32_t normal_transform = 3;      // 3 NormalOctahedronCanonicalized, 2 NormalOctahedron (difference prediction)
  int32_t raw_integers = 0;          // 1 / 2 / 4: values of the difference / parallelogram attributes stored uncompressed at that many
                                     // bytes (SequentialIntegerAttributeDecoder.cs:68-84)
  int32_t no_prediction = 0;         // bit 0 positions, bit 1 texture coordinates, bit 2 normals: prediction method -2 (none)
  int32_t generic_components = 1;    // components of the generic attribute (uint8 each: 4 = the RGBA colours of a scan); CPU coder only
};

// Octahedral quantisation (OctahedronToolBox.cs:28-119)
struct Octa {
  int q, max_q, max_value, center;
  explicit Octa(int bits) { q = bits; max_q = (1 << bits) - 1; max_value = max_q - 1; center = max_value / 2; }
  void canonicalize(int &s, int &t) const {
    if ((s == 0 && t == 0) || (s == 0 && t == max_value) || (s == max_value && t == 0)) { s = max_value; t = max_value; }
    else if (s == 0 && t > center) t = center - (t - center);
    else if (s == max_value && t < center) t = center + (center - t);
    else if (t == max_value && s < center) s = center + (center - s);
    else if (t == 0 && s > center) s = center - (s - center);
  }
  void from_int_vector(const int v[3], int &s, int &t) const {
    if (v[0] >= 0) { s = v[1] + center; t = v[2] + center; }
    else {
      s = v[1] < 0 ? std::abs(v[2]) : max_value - std::abs(v[2]);
      t = v[2] < 0 ? std::abs(v[1]) : max_value - std::abs(v[1]);
    }
    canonicalize(s, t);
  }
  void from_float_vector(const float in[3], int &s, int &t) const {
    double v[3] = {in[0], in[1], in[2]};
    double abs_sum = std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]);
    double sv[3];
    if (abs_sum > 1e-6) { double sc = 1.0 / abs_sum; sv[0] = v[0] * sc; sv[1] = v[1] * sc; sv[2] = v[2] * sc; }
    else { sv[0] = 1; sv[1] = 0; sv[2] = 0; }
    int iv[3];
    iv[0] = (int)std::floor(sv[0] * center + 0.5);
    iv[1] = (int)std::floor(sv[1] * center + 0.5);
    iv[2] = center - std::abs(iv[0]) - std::abs(iv[1]);
    if (iv[2] < 0) { if (iv[1] > 0) iv[1] += iv[2]; else iv[1] -= iv[2]; iv[2] = 0; }
    if (sv[2] < 0) iv[2] *= -1;
    from_int_vector(iv, s, t);
  }
  bool in_diamond(int s, int t) const { return (uint32_t)std::abs(s) + (uint32_t)std::abs(t) <= (uint32_t)center; }
  void invert_diamond(int &s, int &t) const {
    int ss, st;
    if (s >= 0 && t >= 0) { ss = 1; st = 1; }
    else if (s <= 0 && t <= 0) { ss = -1; st = -1; }
    else { ss = s > 0 ? 1 : -1; st = t > 0 ? 1 : -1; }
    int cs = ss * center, ctt = st * center;
    int us = s + s - cs, ut = t + t - ctt, tmp = us;
    if (ss * st >= 0) { us = -ut; ut = -tmp; } else { us = ut; ut = tmp; }
    us += cs; ut += ctt;
    s = us / 2; t = ut / 2;
  }
  int make_positive(int x) const { return x < 0 ? x + max_q : x; }
};

struct PortableAttr {
  int att_type, nc_out, nc;            // nc = portable components
  int seq_type;                        // 1 integer, 2 quantisation, 3 normals
  int data_type;
  std::vector<int32_t> vals;           // per value id, AoS (value id = vertex unless corner_value is set)
  const uint32_t *corner_value = nullptr;   // value id per corner of the source mesh (attributes given per corner: seams)
  std::vector<float> qmin; float qrange = 1; int bits = 0;
  int prediction = 1;
};

// AttributeQuantizationTransform.cs:66-108,136-177 + Core/Quantizer.cs (E-1 corrected)
static void quantize(const float *src, uint32_t n, int nc, int bits, PortableAttr &a) {
  a.qmin.assign(nc, 0);
  std::vector<float> mx(nc, 0);
  for (int c = 0; c < nc; ++c) { a.qmin[c] = src[c]; mx[c] = src[c]; }
  for (uint32_t i = 1; i < n; ++i)
    for (int c = 0; c < nc; ++c) { float v = src[(size_t)i * nc + c]; if (v < a.qmin[c]) a.qmin[c] = v; if (v > mx[c]) mx[c] = v; }
  a.qrange = 0;
  for (int c = 0; c < nc; ++c) { float d = mx[c] - a.qmin[c]; if (d > a.qrange) a.qrange = d; }
  if (a.qrange == 0.0f) a.qrange = 1.0f;
  a.bits = bits;
  int32_t max_q = (1 << bits) - 1;
  volatile float inv_delta = (float)max_q / a.qrange;
  a.vals.resize((size_t)n * nc);
  for (uint32_t i = 0; i < n; ++i)
    for (int c = 0; c < nc; ++c) {
      volatile float v = src[(size_t)i * nc + c] - a.qmin[c];
      volatile float s = v * inv_delta;
      a.vals[(size_t)i * nc + c] = (int32_t)std::floor(s + 0.5f);
    }
}

// Corrections -----------------------------------------------------------------
struct WrapEnc {   // PredictionSchemeWrapEncodingTransform.cs:45-90 (E-5 corrected) + WrapTransform.cs:88-100
  int32_t mn = 0, mx = 0, max_dif = 0, max_corr = 0, min_corr = 0;
  void init(const std::vector<int32_t> &d) {
    if (d.empty()) return;
    mn = mx = d[0];
    for (int32_t v : d) { if (v < mn) mn = v; if (v > mx) mx = v; }
    max_dif = 1 + mx - mn;
    max_corr = max_dif / 2;
    min_corr = -max_corr;
    if ((max_dif & 1) == 0) max_corr -= 1;
  }
  int32_t corr(int32_t orig, int32_t pred) const {
    int32_t p = pred > mx ? mx : (pred < mn ? mn : pred);
    int32_t c = orig - p;
    if (c < min_corr) c += max_dif; else if (c > max_corr) c -= max_dif;
    return c;
  }
};

static void rotate(int &x, int &y, int rot) {
  int a = x, b = y;
  switch (rot) { case 1: x = b; y = -a; break; case 2: x = -a; y = -b; break; case 3: x = -b; y = a; break; default: break; }
}
// PredictionSchemeNormalOctahedronCanonicalizedEncodingTransform.cs:47-83
static void oct_canon_corr(const Octa &o, const int32_t orig_in[2], const int32_t pred_in[2], int32_t out[2]) {
  int os = orig_in[0] - o.center, ot = orig_in[1] - o.center;
  int ps = pred_in[0] - o.center, pt = pred_in[1] - o.center;
  if (!o.in_diamond(ps, pt)) { o.invert_diamond(os, ot); o.invert_diamond(ps, pt); }
  bool bottom_left = (ps == 0 && pt == 0) || (ps < 0 && pt <= 0);
  if (!bottom_left) {
    int rot;
    if (ps == 0) rot = pt == 0 ? 0 : (pt > 0 ? 3 : 1);
    else if (ps > 0) rot = pt >= 0 ? 2 : 1;
    else rot = pt <= 0 ? 0 : 3;
    rotate(os, ot, rot); rotate(ps, pt, rot);
  }
  out[0] = o.make_positive(os - ps);
  out[1] = o.make_positive(ot - pt);
}

struct MeshIn {
  const float *pos; uint32_t nv; const uint32_t *faces; uint32_t nf;
  const float *normals; const float *uvs; const uint8_t *generic;
  // Attributes given per corner (the CPU coder only): value ids per corner of `faces` (3 * nf) into `normals` (nn rows) /
  // `uvs` (nu rows); an interior edge whose end points carry different ids on its two faces is an attribute seam
  // (MeshAttributeCornerTable.cs:32-78).  Null: one value per vertex.
  const uint32_t *normal_corners = nullptr; uint32_t nn = 0;
  const uint32_t *uv_corners = nullptr; uint32_t nu = 0;
};

// One attribute's value section: method, transform, compressed flag, symbols, prediction data
// (SequentialIntegerAttributeEncoder.cs:55-128)
// Area-weighted normal of the faces around a corner's vertex from the quantised positions, canonicalised to the
// octahedron (MeshPredictionSchemeGeometricNormalPredictorArea.cs:16-63 + OctahedronToolBox.cs:121-137, with the
// bitstream's 64-bit arithmetic).  Per-vertex attributes only: a data id's position is its vertex's.
template <class CT>
static void geometric_normal_prediction(const Octa &o, const CT &ct, const CornerTable &pos_ct, const std::vector<int32_t> &pos, uint32_t ci, int32_t v3[3]) {
  auto P = [&](uint32_t c, int k) { return (int64_t)pos[(size_t)pos_ct.vertex(c) * 3 + k]; };
  uint64_t n[3] = {0, 0, 0};
  uint32_t c = ci;
  bool left = true;
  while (c != kInvalid) {
    uint32_t cn = CornerTable::next(c), cp = CornerTable::prev(c);
    uint64_t a[3], b[3];
    for (int k = 0; k < 3; ++k) { a[k] = (uint64_t)(P(cn, k) - P(ci, k)); b[k] = (uint64_t)(P(cp, k) - P(ci, k)); }
    n[0] += a[1] * b[2] - a[2] * b[1];
    n[1] += a[2] * b[0] - a[0] * b[2];
    n[2] += a[0] * b[1] - a[1] * b[0];
    if (left) {
      c = ct.swing_left(c);
      if (c == kInvalid) { c = ct.swing_right(ci); left = false; }
      else if (c == ci) break;
    } else c = ct.swing_right(c);
  }
  int64_t nv[3] = {(int64_t)n[0], (int64_t)n[1], (int64_t)n[2]};
  uint64_t as = 0;
  bool sat = false;
  for (int k = 0; k < 3; ++k) {
    uint64_t x = nv[k] < 0 ? (uint64_t)0 - (uint64_t)nv[k] : (uint64_t)nv[k];
    if (x > (uint64_t)INT64_MAX || as > (uint64_t)INT64_MAX - x) sat = true; else as += x;
  }
  int64_t abs_sum = sat ? INT64_MAX : (int64_t)as;
  const int64_t upper = (int64_t)1 << 29;
  if (abs_sum > upper) { int64_t q = abs_sum / upper; for (int k = 0; k < 3; ++k) nv[k] /= q; }
  for (int k = 0; k < 3; ++k) v3[k] = (int32_t)nv[k];
  int64_t s3 = std::llabs((int64_t)v3[0]) + std::llabs((int64_t)v3[1]) + std::llabs((int64_t)v3[2]);
  if (s3 == 0) v3[0] = o.center;
  else {
    v3[0] = (int32_t)(((int64_t)v3[0] * o.center) / s3);
    v3[1] = (int32_t)(((int64_t)v3[1] * o.center) / s3);
    int32_t rest = o.center - std::abs(v3[0]) - std::abs(v3[1]);
    v3[2] = v3[2] >= 0 ? rest : -rest;
  }
}

// PredictionSchemeNormalOctahedronEncodingTransform.cs (the non-canonicalised transform: no rotation step)
static void oct_plain_corr(const Octa &o, const int32_t orig_in[2], const int32_t pred_in[2], int32_t out[2]) {
  int os = orig_in[0] - o.center, ot = orig_in[1] - o.center;
  int ps = pred_in[0] - o.center, pt = pred_in[1] - o.center;
  if (!o.in_diamond(ps, pt)) { o.invert_diamond(os, ot); o.invert_diamond(ps, pt); }
  out[0] = o.make_positive(os - ps);
  out[1] = o.make_positive(ot - pt);
}

template <class CT>
static void write_attribute_values(ByteWriter &w, const PortableAttr &a, const CT &ct, const CornerTable &pos_ct, const Sequence &seq, const Options &opt,
                                   const PortableAttr *positions = nullptr) {
  int nc = a.nc;
  size_t entries = seq.data_to_corner.size();
  // compressed flag + symbols: through the entropy coder, or as they are at 1 / 2 / 4 bytes each
  auto put_symbols = [&](const std::vector<uint32_t> &sy) {
    if (opt.raw_integers == 0) { w.u8(1); encode_symbols(w, sy, nc, opt.force_scheme, opt.compression_level); return; }
    const int nb = opt.raw_integers;
    check(nb == 1 || nb == 2 || nb == 4, "raw_integers must be 1, 2 or 4");
    w.u8(0);
    w.u8((uint8_t)nb);
    for (uint32_t v : sy) {
      check(nb == 4 || v < (1u << (8 * nb)), "value does not fit the raw integer width");
      for (int k = 0; k < nb; ++k) w.u8((uint8_t)(v >> (8 * k)));
    }
  };
  // values in entry order
  std::vector<int32_t> d(entries * nc);
  for (size_t e = 0; e < entries; ++e) {
    const uint32_t corner = seq.data_to_corner[e];
    const uint32_t v = a.corner_value ? a.corner_value[corner] : pos_ct.vertex(corner);
    for (int c = 0; c < nc; ++c) d[e * nc + c] = a.vals[(size_t)v * nc + c];
  }
  std::vector<uint32_t> symbols(entries * nc);
  {
    const int none_bit = a.seq_type == 3 ? 4 : (a.att_type == 0 ? 1 : (a.att_type == 3 ? 2 : 0));
    if (opt.no_prediction & none_bit) {        // PredictionSchemeMethod.None (-2): no transform byte, the values themselves, signed
      w.i8(-2);
      for (size_t i = 0; i < symbols.size(); ++i) symbols[i] = zigzag(d[i]);
      put_symbols(symbols);
      return;
    }
  }
  if (a.seq_type == 3 && a.prediction == 6) {
    // MeshPredictionSchemeGeometricNormalEncoder.cs:47-108: the prediction or its negation, whichever leaves the
    // smaller correction; one flip bit per entry behind the transform data
    check(positions != nullptr && positions->nc == 3, "geometric normal prediction needs quantised positions");
    w.i8(6);
    w.i8(3);
    Octa o(a.bits);
    std::vector<uint8_t> flips(entries);
    auto mod_max = [&](int x) { return x > o.center ? x - o.max_q : (x < -o.center ? x + o.max_q : x); };
    for (size_t e = 0; e < entries; ++e) {
      int32_t v3[3];
      geometric_normal_prediction(o, ct, pos_ct, positions->vals, seq.data_to_corner[e], v3);
      int32_t pp[2], pn[2], cp[2], cn[2];
      int s, t;
      o.from_int_vector(v3, s, t); pp[0] = s; pp[1] = t;
      int32_t neg[3] = {-v3[0], -v3[1], -v3[2]};
      o.from_int_vector(neg, s, t); pn[0] = s; pn[1] = t;
      oct_canon_corr(o, &d[e * 2], pp, cp);
      oct_canon_corr(o, &d[e * 2], pn, cn);
      int wp = std::abs(mod_max(cp[0])) + std::abs(mod_max(cp[1]));
      int wn = std::abs(mod_max(cn[0])) + std::abs(mod_max(cn[1]));
      const bool flip = !(wp < wn);
      flips[e] = flip;
      symbols[e * 2] = (uint32_t)(flip ? cn[0] : cp[0]); symbols[e * 2 + 1] = (uint32_t)(flip ? cn[1] : cp[1]);
    }
    w.u8(1);
    encode_symbols(w, symbols, nc, opt.force_scheme, opt.compression_level);
    w.i32(o.max_q);
    w.i32(o.center);
    write_rabs(w, flips);
    return;
  }
  if (a.seq_type == 3) {
    const bool canonical = opt.normal_transform != 2;
    w.i8(0);   // Difference
    w.i8(canonical ? 3 : 2);   // NormalOctahedronCanonicalized / NormalOctahedron
    Octa o(a.bits);
    int32_t zero[2] = {0, 0};
    for (size_t e = entries; e-- > 0;) {
      int32_t out[2];
      if (canonical) oct_canon_corr(o, &d[e * 2], e ? &d[(e - 1) * 2] : zero, out);
      else oct_plain_corr(o, &d[e * 2], e ? &d[(e - 1) * 2] : zero, out);
      symbols[e * 2] = (uint32_t)out[0]; symbols[e * 2 + 1] = (uint32_t)out[1];   // positive: no zig-zag
    }
    put_symbols(symbols);
    w.i32(o.max_q);
    if (canonical) w.i32(o.center);      // PredictionSchemeNormalOctahedronCanonicalizedDecodingTransform reads a centre it does not use
    return;
  }
  if (a.prediction == 5) {
    // MeshPredictionSchemeTexCoordsPortableEncoder.cs + ...PortablePredictor.cs:46-150 (encoder side: both candidate
    // predictions, the closer one wins and its orientation is recorded), last entry first
    check(nc == 2 && positions != nullptr && positions->nc == 3, "texture coordinate prediction needs 2 components and quantised positions");
    WrapEnc wr;
    wr.init(d);
    w.i8(5);
    w.i8(1);
    std::vector<uint8_t> orientations;
    auto isqrt = [](uint64_t number) {   // Core/MathUtilities.cs:5-25
      if (number == 0) return (uint64_t)0;
      uint64_t act = number, root = 1;
      while (act >= 2) { root *= 2; act /= 4; }
      do { root = (root + number / root) / 2; } while (root * root > number);
      return root;
    };
    auto P = [&](int32_t entry, int k) { return (int64_t)positions->vals[(size_t)pos_ct.vertex(seq.data_to_corner[entry]) * 3 + k]; };
    for (size_t p = entries; p-- > 0;) {
      const int32_t data_id = (int32_t)p;
      const uint32_t ci = seq.data_to_corner[p];
      const int32_t next_id = seq.vertex_to_data[ct.vertex(CornerTable::next(ci))], prev_id = seq.vertex_to_data[ct.vertex(CornerTable::prev(ci))];
      int32_t pred[2] = {0, 0};
      bool done = false;
      if (prev_id >= 0 && next_id >= 0 && prev_id < data_id && next_id < data_id) {
        const int64_t n_uv[2] = {d[next_id * 2], d[next_id * 2 + 1]}, p_uv[2] = {d[prev_id * 2], d[prev_id * 2 + 1]};
        if (p_uv[0] == n_uv[0] && p_uv[1] == n_uv[1]) { pred[0] = (int32_t)p_uv[0]; pred[1] = (int32_t)p_uv[1]; done = true; }
        else {
          int64_t pn[3], cn[3];
          for (int k = 0; k < 3; ++k) { pn[k] = P(prev_id, k) - P(next_id, k); cn[k] = P(data_id, k) - P(next_id, k); }
          const int64_t pn_norm2 = pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2];
          if (pn_norm2 != 0) {
            const int64_t cn_dot_pn = pn[0] * cn[0] + pn[1] * cn[1] + pn[2] * cn[2];
            const int64_t pn_uv[2] = {p_uv[0] - n_uv[0], p_uv[1] - n_uv[1]};
            const int64_t x_uv[2] = {n_uv[0] * pn_norm2 + cn_dot_pn * pn_uv[0], n_uv[1] * pn_norm2 + cn_dot_pn * pn_uv[1]};
            int64_t cx[3];
            for (int k = 0; k < 3; ++k) cx[k] = P(data_id, k) - (P(next_id, k) + (cn_dot_pn * pn[k]) / pn_norm2);
            const uint64_t cx_norm2 = (uint64_t)(cx[0] * cx[0] + cx[1] * cx[1] + cx[2] * cx[2]);
            const int64_t norm = (int64_t)isqrt(cx_norm2 * (uint64_t)pn_norm2);
            const int64_t cx_uv[2] = {pn_uv[1] * norm, -pn_uv[0] * norm};
            const int64_t c0[2] = {(x_uv[0] + cx_uv[0]) / pn_norm2, (x_uv[1] + cx_uv[1]) / pn_norm2};
            const int64_t c1[2] = {(x_uv[0] - cx_uv[0]) / pn_norm2, (x_uv[1] - cx_uv[1]) / pn_norm2};
            const int64_t u = d[p * 2], v = d[p * 2 + 1];
            const uint64_t e0 = (uint64_t)((u - c0[0]) * (u - c0[0]) + (v - c0[1]) * (v - c0[1]));
            const uint64_t e1 = (uint64_t)((u - c1[0]) * (u - c1[0]) + (v - c1[1]) * (v - c1[1]));
            if (e0 < e1) { pred[0] = (int32_t)c0[0]; pred[1] = (int32_t)c0[1]; orientations.push_back(1); }
            else { pred[0] = (int32_t)c1[0]; pred[1] = (int32_t)c1[1]; orientations.push_back(0); }
            done = true;
          }
        }
      }
      if (!done) {
        int32_t data_offset = 0;
        bool zero = false;
        if (prev_id >= 0 && prev_id < data_id) data_offset = prev_id * 2;
        if (next_id >= 0 && next_id < data_id) data_offset = next_id * 2;
        else { if (data_id > 0) data_offset = (data_id - 1) * 2; else zero = true; }
        if (!zero) { pred[0] = d[data_offset]; pred[1] = d[data_offset + 1]; }
      }
      symbols[p * 2] = zigzag(wr.corr(d[p * 2], pred[0]));
      symbols[p * 2 + 1] = zigzag(wr.corr(d[p * 2 + 1], pred[1]));
    }
    w.u8(1);
    encode_symbols(w, symbols, nc, opt.force_scheme, opt.compression_level);
    // orientations in the order they were found (last entry first), delta-coded against `true`
    w.i32((int32_t)orientations.size());
    std::vector<uint8_t> bits(orientations.size());
    bool last = true;
    for (size_t i = 0; i < orientations.size(); ++i) { const bool o = orientations[i] != 0; bits[i] = o == last; last = o; }
    write_rabs(w, bits);
    w.i32(wr.mn);
    w.i32(wr.mx);
    return;
  }
  WrapEnc wr;
  wr.init(d);
  w.i8((int8_t)a.prediction);
  w.i8(1);     // Wrap
  std::vector<int32_t> pred(nc);
  if (a.prediction == 2 || a.prediction == 4) {
    // MeshPredictionSchemeMultiParallelogramEncoder.cs / ...ConstrainedMultiParallelogramEncoder.cs: averages of the
    // parallelograms around the entry's vertex.  The constrained scheme may drop any of (up to four) parallelograms
    // by a crease flag; this coder takes the subset with the smallest correction (the reference's entropy-driven
    // choice is an encoder heuristic, the stream is valid for any choice).
    auto para = [&](size_t p, uint32_t ci, int32_t *out) {
      const uint32_t oci = ct.opposite(ci);
      if (oci == kInvalid) return false;
      const int32_t vo = seq.vertex_to_data[ct.vertex(oci)], vn = seq.vertex_to_data[ct.vertex(CornerTable::next(oci))], vp = seq.vertex_to_data[ct.vertex(CornerTable::prev(oci))];
      if (!(vo < (int32_t)p && vn < (int32_t)p && vp < (int32_t)p)) return false;
      for (int c = 0; c < nc; ++c) out[c] = (int32_t)((uint32_t)d[vn * nc + c] + (uint32_t)d[vp * nc + c] - (uint32_t)d[vo * nc + c]);
      return true;
    };
    std::vector<uint8_t> crease[4];
    std::vector<int32_t> cand(4 * nc), sum(nc);
    for (int c = 0; c < nc; ++c) symbols[c] = zigzag(wr.corr(d[c], 0));
    for (size_t p = 1; p < entries; ++p) {
      const uint32_t start = seq.data_to_corner[p];
      uint32_t c = start;
      int found = 0;
      bool have = false;
      if (a.prediction == 2) {
        std::fill(sum.begin(), sum.end(), 0);
        while (c != kInvalid) {
          if (para(p, c, cand.data())) { for (int k = 0; k < nc; ++k) sum[k] = (int32_t)((uint32_t)sum[k] + (uint32_t)cand[k]); ++found; }
          c = ct.swing_right(c);
          if (c == start) c = kInvalid;
        }
        if (found) { for (int k = 0; k < nc; ++k) pred[k] = sum[k] / found; have = true; }
      } else {
        bool first_pass = true;
        while (c != kInvalid) {
          if (para(p, c, &cand[(size_t)found * nc])) { if (++found == 4) break; }
          c = first_pass ? ct.swing_left(c) : ct.swing_right(c);
          if (c == start) break;
          if (c == kInvalid && first_pass) { first_pass = false; c = ct.swing_right(start); }
        }
        if (found) {
          int64_t best_cost = -1;
          uint32_t best_mask = 0;
          for (uint32_t mask = 0; mask < (1u << found); ++mask) {      // bit i set: parallelogram i is used
            int used = 0;
            std::fill(sum.begin(), sum.end(), 0);
            for (int i = 0; i < found; ++i) if (mask >> i & 1) { ++used; for (int k = 0; k < nc; ++k) sum[k] = (int32_t)((uint32_t)sum[k] + (uint32_t)cand[(size_t)i * nc + k]); }
            int64_t cost = 0;
            for (int k = 0; k < nc; ++k) cost += std::abs((int64_t)wr.corr(d[p * nc + k], used ? sum[k] / used : d[(p - 1) * nc + k]));
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_mask = mask; }
          }
          int used = 0;
          std::fill(sum.begin(), sum.end(), 0);
          for (int i = 0; i < found; ++i) {
            const bool use = best_mask >> i & 1;
            crease[found - 1].push_back(use ? 0 : 1);
            if (use) { ++used; for (int k = 0; k < nc; ++k) sum[k] = (int32_t)((uint32_t)sum[k] + (uint32_t)cand[(size_t)i * nc + k]); }
          }
          if (used) { for (int k = 0; k < nc; ++k) pred[k] = sum[k] / used; have = true; }
        }
      }
      if (!have) for (int k = 0; k < nc; ++k) pred[k] = d[(p - 1) * nc + k];
      for (int k = 0; k < nc; ++k) symbols[p * nc + k] = zigzag(wr.corr(d[p * nc + k], pred[k]));
    }
    w.u8(1);
    encode_symbols(w, symbols, nc, opt.force_scheme, opt.compression_level);
    if (a.prediction == 4)
      for (int i = 0; i < 4; ++i) { w.varint(crease[i].size()); if (!crease[i].empty()) write_rabs(w, crease[i]); }
    w.i32(wr.mn);
    w.i32(wr.mx);
    return;
  }
  for (size_t p = entries; p-- > 0;) {
    bool have = false;
    if (a.prediction == 1 && p > 0) {   // MeshPredictionSchemeParallelogramEncoder.cs:35-56 (E-4 corrected)
      uint32_t ci = seq.data_to_corner[p];
      uint32_t oci = ct.opposite(ci);
      if (oci != kInvalid) {
        int32_t vo = seq.vertex_to_data[ct.vertex(oci)];
        int32_t vn = seq.vertex_to_data[ct.vertex(CornerTable::next(oci))];
        int32_t vp = seq.vertex_to_data[ct.vertex(CornerTable::prev(oci))];
        if (vo < (int32_t)p && vn < (int32_t)p && vp < (int32_t)p) {
          for (int c = 0; c < nc; ++c) pred[c] = d[vn * nc + c] + d[vp * nc + c] - d[vo * nc + c];
          have = true;
        }
      }
    }
    if (!have) for (int c = 0; c < nc; ++c) pred[c] = p > 0 ? d[(p - 1) * nc + c] : 0;
    for (int c = 0; c < nc; ++c) symbols[p * nc + c] = zigzag(wr.corr(d[p * nc + c], pred[c]));
  }
  put_symbols(symbols);
  w.i32(wr.mn);
  w.i32(wr.mx);
}

static void write_attribute_transform(ByteWriter &w, const PortableAttr &a) {
  if (a.seq_type == 2) { for (float f : a.qmin) w.f32(f); w.f32(a.qrange); w.u8((uint8_t)a.bits); }
  else if (a.seq_type == 3) w.u8((uint8_t)a.bits);
}

// Everything about a mesh that does not depend on attribute values: connectivity, traversal order, attribute list.
struct MeshPlan {
  CornerTable ct;
  EbResult eb;
  Sequence seq;                      // depth-first order
  Sequence seq_pd;                   // prediction-degree order (only when an attributes decoder asks for it)
  int traversal_method = 0;
  int force_scheme = -1, compression_level = 5;
  bool valence = false;              // valence Edgebreaker traversal: six context symbol lists
  std::vector<uint32_t> ctx_symbols[6];
  bool predictive = false;           // predictive Edgebreaker traversal: explicit symbols + prediction bits below
  std::vector<uint8_t> explicit_symbols, predictions;   // encoder order
  // Attributes given per corner: conns[att] is the attribute's own connectivity, seq_att[att] its depth-first order
  // (corner attributes are always sequenced depth first: MeshEdgeBreakerEncoder.cs:556-564); both empty for meshes
  // whose attributes are all per vertex.
  std::vector<AttrConn> conns;
  std::vector<Sequence> seq_att;
  bool seamed(size_t att) const { return att < conns.size() && conns[att].ct && !conns[att].no_interior_seams; }
  bool uses_pd(size_t att) const { return !seamed(att) && (traversal_method == 2 || (traversal_method == 1 && (single || att == 0))); }
  const Sequence &seq_of(size_t att) const { return seamed(att) ? seq_att[att] : (uses_pd(att) ? seq_pd : seq); }
  std::vector<PortableAttr> atts;    // descriptors; vals / quantisation parameters are filled by whoever codes the values
  bool single = false;
  uint32_t num_att_data = 0;
  int64_t interior_edges = -1;       // >= 0: given (connectivity coded on the device, no corner table here); else counted from ct
};
// What of a plan does not depend on the connectivity: attribute descriptors and the options that shape the stream.
static void plan_attributes(const MeshIn &in, const Options &opt, MeshPlan &pl) {
  pl.atts.clear();
  { PortableAttr a; a.att_type = 0; a.nc = a.nc_out = 3; a.seq_type = 2; a.data_type = 9; a.prediction = opt.pos_prediction; a.bits = opt.pos_bits; pl.atts.push_back(a); }
  if (in.normals) { PortableAttr a; a.att_type = 1; a.nc_out = 3; a.nc = 2; a.seq_type = 3; a.data_type = 9; a.bits = opt.normal_bits; a.prediction = opt.normal_prediction == 6 ? 6 : 0; a.corner_value = in.normal_corners; pl.atts.push_back(a); }
  if (in.uvs) { PortableAttr a; a.att_type = 3; a.nc = a.nc_out = 2; a.seq_type = 2; a.data_type = 9; a.prediction = opt.uv_prediction; a.bits = opt.uv_bits; a.corner_value = in.uv_corners; pl.atts.push_back(a); }
  // (the generic attribute takes the constrained multi-parallelogram scheme where the positions do: what an encoder at its highest levels writes)
  if (in.generic) { PortableAttr a; a.att_type = 4; a.nc = a.nc_out = opt.generic_components >= 1 && opt.generic_components <= 4 ? opt.generic_components : 1; a.seq_type = 1; a.data_type = 2; a.prediction = opt.pos_prediction == 4 ? 4 : 1; pl.atts.push_back(a); }
  pl.single = opt.single_connectivity != 0;
  pl.num_att_data = pl.single ? 0 : (uint32_t)pl.atts.size() - 1;
  pl.force_scheme = opt.force_scheme; pl.compression_level = opt.compression_level;
  pl.predictive = opt.predictive_connectivity == 1;
  pl.valence = opt.predictive_connectivity == 2;
  pl.traversal_method = opt.traversal_method;
}
static void plan_mesh(const MeshIn &in, const Options &opt, MeshPlan &pl) {
  pl.ct.build(in.faces, in.nf, in.nv);
  for (uint32_t v = 0; v < in.nv; ++v) check(pl.ct.vcorner[v] != kInvalid, "isolated vertex in input mesh");
  EbEncoder enc(pl.ct, pl.eb);
  enc.run();
  dfs_sequence(pl.ct, pl.eb.processed_corners, pl.seq);
  check(pl.seq.data_to_corner.size() == in.nv, "traversal did not reach every vertex");
  plan_attributes(in, opt, pl);
  if (pl.valence) valence_context_symbols(pl.ct, pl.eb, pl.ctx_symbols);
  if (pl.predictive) predictive_symbols(pl.ct, pl.eb, pl.explicit_symbols, pl.predictions);
  if (pl.traversal_method != 0) {
    prediction_degree_sequence(pl.ct, pl.eb.processed_corners, pl.seq_pd);
    check(pl.seq_pd.data_to_corner.size() == in.nv, "traversal did not reach every vertex");
  }
  // attributes given per corner: their seams, their own vertices and traversal order (MeshEdgeBreakerEncoder.cs:403-414,
  // :545-566: the sequencer of a seamed attribute walks the attribute's corner table in the connectivity's face order)
  bool per_corner = false;
  for (auto &a : pl.atts) per_corner = per_corner || a.corner_value != nullptr;
  if (per_corner) {
    check(!pl.single, "attributes given per corner need a connectivity of their own (single_connectivity = 0)");
    pl.conns.assign(pl.atts.size(), AttrConn());
    pl.seq_att.assign(pl.atts.size(), Sequence());
    for (size_t i = 1; i < pl.atts.size(); ++i) {
      if (!pl.atts[i].corner_value) continue;
      pl.conns[i].build(pl.ct, pl.atts[i].corner_value);
      if (pl.conns[i].no_interior_seams) continue;
      dfs_sequence(pl.conns[i], pl.eb.processed_corners, pl.seq_att[i]);
      check(pl.seq_att[i].data_to_corner.size() == pl.conns[i].nv(), "attribute traversal did not reach every attribute vertex");
    }
  }
}

// Header, connectivity sections and the head of the attribute section; then values and transform parameters of
// every attribute through the two callbacks, in the order the decoder expects (ConnectivityEncoder.cs:39-56).
template <class ValuesWriter, class TransformWriter>
static void write_stream(ByteWriter &w, const MeshIn &in, const MeshPlan &pl, ValuesWriter &&values, TransformWriter &&transform) {
  const CornerTable &ct = pl.ct;
  const EbResult &eb = pl.eb;
  w.d.insert(w.d.end(), {'D', 'R', 'A', 'C', 'O'});
  w.u8(2); w.u8(2); w.u8(1); w.u8(1); w.u16(0);
  w.u8(pl.valence ? 2 : (pl.predictive ? 1 : 0));   // Edgebreaker traversal: standard (DracoEncoder.cs:90), predictive or valence
  w.varint(in.nv);
  w.varint(in.nf);
  w.u8((uint8_t)pl.num_att_data);
  w.varint(eb.symbols.size());
  w.varint(eb.num_split_symbols);
  // split events, MeshEdgeBreakerEncoder.cs:126-148
  w.varint(eb.splits.size());
  if (!eb.splits.empty()) {
    uint32_t last = 0;
    for (auto &s : eb.splits) { w.varint(s.source - last); w.varint(s.source - s.split); last = s.source; }
    BitWriter bw;
    for (auto &s : eb.splits) bw.put(1, s.edge);
    w.bytes(bw.d);
  }
  // traversal buffer: symbols last-first (size = bytes, E-8), start faces, seams
  {
    BitWriter bw;
    static const int len[8] = {1, 3, 0, 3, 0, 3, 0, 3};
    const std::vector<uint8_t> &syms = pl.predictive ? pl.explicit_symbols : eb.symbols;
    if (!pl.valence) {
      for (size_t i = syms.size(); i-- > 0;) bw.put(len[syms[i]], syms[i]);
      w.varint(bw.d.size());
      w.bytes(bw.d);
    }
    write_rabs(w, eb.start_face_bits);
    bool any_seams = false;
    for (size_t i = 1; i < pl.atts.size(); ++i) any_seams = any_seams || pl.seamed(i);
    if (pl.num_att_data && any_seams) {
      // MeshEdgeBreakerEncoder.cs:418-440: in decoder face order, for every interior edge whose other face comes later,
      // one bit per attribute -- is the edge a seam of that attribute; a block per attribute (MeshEdgeBreakerTraversalEncoder.cs:62-70)
      std::vector<uint8_t> vis(ct.nf(), 0);
      std::vector<std::vector<uint8_t>> bits(pl.num_att_data);
      for (uint32_t c : eb.processed_corners) {
        const uint32_t corners[3] = {c, CornerTable::next(c), CornerTable::prev(c)};
        vis[c / 3] = 1;
        for (int k = 0; k < 3; ++k) {
          const uint32_t o = ct.opposite(corners[k]);
          if (o == kInvalid || vis[o / 3]) continue;
          for (uint32_t i = 0; i < pl.num_att_data; ++i) bits[i].push_back(pl.seamed(i + 1) ? pl.conns[i + 1].edge_seam[corners[k]] : 0);
        }
      }
      for (uint32_t i = 0; i < pl.num_att_data; ++i) write_rabs(w, bits[i]);
    } else if (pl.num_att_data) {
      // per-vertex attributes: no interior seams, one 0 bit per interior edge in decoder face order
      std::vector<uint8_t> vis(pl.interior_edges >= 0 ? 0 : ct.nf(), 0), bits;
      if (pl.interior_edges >= 0) bits.assign((size_t)pl.interior_edges, 0);
      else for (uint32_t c : eb.processed_corners) {
        uint32_t corners[3] = {c, CornerTable::next(c), CornerTable::prev(c)};
        vis[c / 3] = 1;
        for (int k = 0; k < 3; ++k) {
          uint32_t o = ct.opposite(corners[k]);
          if (o == kInvalid || vis[o / 3]) continue;
          bits.push_back(0);
        }
      }
      ByteWriter once;                       // the same block for every attribute: coded once
      write_rabs(once, bits);
      for (uint32_t i = 0; i < pl.num_att_data; ++i) w.bytes(once.d);
    }
    if (pl.valence)        // MeshEdgeBreakerTraversalValenceDecoder.cs:43-68: the six context lists, each through the symbol coder
      for (int i = 0; i < 6; ++i) {
        w.varint(pl.ctx_symbols[i].size());
        if (!pl.ctx_symbols[i].empty()) encode_symbols(w, pl.ctx_symbols[i], 1, pl.force_scheme, pl.compression_level);
      }
    if (pl.predictive) {   // MeshEdgeBreakerTraversalPredictiveDecoder.cs:19-27: split symbol count, then the prediction bits in decoder order
      w.i32((int32_t)eb.num_split_symbols);
      std::vector<uint8_t> bits(pl.predictions.rbegin(), pl.predictions.rend());
      write_rabs(w, bits);
    }
  }
  // attribute section (ConnectivityEncoder.cs:39-56)
  const std::vector<PortableAttr> &atts = pl.atts;
  uint32_t num_encoders = pl.single ? 1 : (uint32_t)atts.size();
  w.u8((uint8_t)num_encoders);
  // attribute data id, element type (1: corner attribute -- the attribute's own connectivity is used, :442-466), MeshTraversalMethod
  for (uint32_t i = 0; i < num_encoders; ++i) { w.i8(i == 0 ? -1 : (int8_t)(i - 1)); w.u8(pl.seamed(i) ? 1 : 0); w.u8(pl.uses_pd(i) ? 1 : 0); }
  auto write_desc = [&](const PortableAttr &a, uint32_t uid) { w.u8((uint8_t)a.att_type); w.u8((uint8_t)a.data_type); w.u8((uint8_t)a.nc_out); w.u8(0); w.varint(uid); };
  if (pl.single) {
    w.varint(atts.size());
    for (size_t i = 0; i < atts.size(); ++i) write_desc(atts[i], (uint32_t)i);
    for (auto &a : atts) w.u8((uint8_t)a.seq_type);
    for (size_t i = 0; i < atts.size(); ++i) values(w, i);
    for (size_t i = 0; i < atts.size(); ++i) transform(w, i);
  } else {
    for (size_t i = 0; i < atts.size(); ++i) { w.varint(1); write_desc(atts[i], (uint32_t)i); w.u8((uint8_t)atts[i].seq_type); }
    for (size_t i = 0; i < atts.size(); ++i) { values(w, i); transform(w, i); }
  }
}

// The CPU coder: quantisation, prediction, entropy coding all on the host.
static void encode_mesh(const MeshIn &in, const Options &opt, std::vector<uint8_t> &out) {
  MeshPlan pl;
  plan_mesh(in, opt, pl);
  for (auto &a : pl.atts) {
    if (a.att_type == 0) quantize(in.pos, in.nv, 3, opt.pos_bits, a);
    else if (a.att_type == 1) {
      const uint32_t n = in.normal_corners ? in.nn : in.nv;
      Octa o(opt.normal_bits);
      a.vals.resize((size_t)n * 2);
      for (uint32_t v = 0; v < n; ++v) { int s, t; o.from_float_vector(in.normals + (size_t)v * 3, s, t); a.vals[(size_t)v * 2] = s; a.vals[(size_t)v * 2 + 1] = t; }
    } else if (a.att_type == 3) quantize(in.uvs, in.uv_corners ? in.nu : in.nv, 2, opt.uv_bits, a);
    else { a.vals.resize((size_t)in.nv * a.nc); for (size_t k = 0; k < (size_t)in.nv * a.nc; ++k) a.vals[k] = in.generic[k]; }
  }
  ByteWriter w;
  write_stream(w, in, pl,
               [&](ByteWriter &bw, size_t i) {
                 if (pl.seamed(i)) write_attribute_values(bw, pl.atts[i], pl.conns[i], pl.ct, pl.seq_of(i), opt, &pl.atts[0]);
                 else write_attribute_values(bw, pl.atts[i], pl.ct, pl.ct, pl.seq_of(i), opt, &pl.atts[0]);
               },
               [&](ByteWriter &bw, size_t i) { write_attribute_transform(bw, pl.atts[i]); });
  out.swap(w.d);
}

// Sequential mesh (Mesh/MeshSequentialEncoder.cs:9-121 with the bitstream's index widths): faces as point indices,
// compressed (differences, sign in the LSB, through the symbol coder) or raw; one attributes encoder with a linear
// sequencer, so values are in point order and predicted by Difference + Wrap / canonicalised octahedral delta.
static void encode_mesh_sequential(const MeshIn &in, const Options &opt, bool compressed, std::vector<uint8_t> &out) {
  ByteWriter w;
  w.d.insert(w.d.end(), {'D', 'R', 'A', 'C', 'O'});
  w.u8(2); w.u8(2); w.u8(1); w.u8(0); w.u16(0);
  w.varint(in.nf);
  w.varint(in.nv);
  if (compressed) {
    w.u8(0);
    std::vector<uint32_t> sym((size_t)in.nf * 3);
    int64_t last = 0;
    for (size_t k = 0; k < sym.size(); ++k) {
      const int64_t diff = (int64_t)in.faces[k] - last;
      sym[k] = ((uint32_t)(diff < 0 ? -diff : diff) << 1) | (diff < 0 ? 1u : 0u);
      last = in.faces[k];
    }
    encode_symbols(w, sym, 1, opt.force_scheme, opt.compression_level);
  } else {
    w.u8(1);
    for (size_t k = 0; k < (size_t)in.nf * 3; ++k) {
      const uint32_t v = in.faces[k];
      if (in.nv < 256) w.u8((uint8_t)v);
      else if (in.nv < (1u << 16)) w.u16((uint16_t)v);
      else if (in.nv < (1u << 21)) w.varint(v);
      else w.u32(v);
    }
  }
  std::vector<PortableAttr> atts;
  { PortableAttr a; a.att_type = 0; a.nc = a.nc_out = 3; a.seq_type = 2; a.data_type = 9; a.prediction = 0; quantize(in.pos, in.nv, 3, opt.pos_bits, a); atts.push_back(a); }
  if (in.normals) {
    PortableAttr a; a.att_type = 1; a.nc_out = 3; a.nc = 2; a.seq_type = 3; a.data_type = 9; a.bits = opt.normal_bits; a.prediction = 0;
    Octa o(opt.normal_bits);
    a.vals.resize((size_t)in.nv * 2);
    for (uint32_t v = 0; v < in.nv; ++v) { int s, t; o.from_float_vector(in.normals + (size_t)v * 3, s, t); a.vals[(size_t)v * 2] = s; a.vals[(size_t)v * 2 + 1] = t; }
    atts.push_back(a);
  }
  if (in.uvs) { PortableAttr a; a.att_type = 3; a.nc = a.nc_out = 2; a.seq_type = 2; a.data_type = 9; a.prediction = 0; quantize(in.uvs, in.nv, 2, opt.uv_bits, a); atts.push_back(a); }
  // linear order: entry i = point i.  write_attribute_values wants a corner table and a sequence: an identity stand-in
  CornerTable ct;
  ct.c2v.resize(in.nv);
  for (uint32_t v = 0; v < in.nv; ++v) ct.c2v[v] = v;
  Sequence seq;
  seq.data_to_corner.resize(in.nv);
  for (uint32_t v = 0; v < in.nv; ++v) seq.data_to_corner[v] = v;
  w.u8(1);
  w.varint(atts.size());
  for (size_t i = 0; i < atts.size(); ++i) { w.u8((uint8_t)atts[i].att_type); w.u8((uint8_t)atts[i].data_type); w.u8((uint8_t)atts[i].nc_out); w.u8(0); w.varint(i); }
  for (auto &a : atts) w.u8((uint8_t)a.seq_type);
  for (auto &a : atts) write_attribute_values(w, a, ct, ct, seq, opt);
  for (auto &a : atts) write_attribute_transform(w, a);
  out.swap(w.d);
}

// Point cloud, sequential (BASELINE config 1): int32 num_points, one attributes
// decoder, positions quantised, Difference + Wrap in linear order.
static void encode_point_cloud(const float *pos, uint32_t n, const Options &opt, std::vector<uint8_t> &out) {
  PortableAttr a; a.att_type = 0; a.nc = a.nc_out = 3; a.seq_type = 2; a.data_type = 9; a.prediction = 0;
  quantize(pos, n, 3, opt.pos_bits, a);
  ByteWriter w;
  w.d.insert(w.d.end(), {'D', 'R', 'A', 'C', 'O'});
  w.u8(2); w.u8(2); w.u8(0); w.u8(0); w.u16(0);
  w.i32((int32_t)n);
  w.u8(1);
  w.varint(1);
  w.u8(0); w.u8(9); w.u8(3); w.u8(0); w.varint(0);
  w.u8(2);
  WrapEnc wr; wr.init(a.vals);
  w.i8(0); w.i8(1);
  std::vector<uint32_t> symbols((size_t)n * 3);
  for (size_t p = n; p-- > 0;)
    for (int c = 0; c < 3; ++c) symbols[p * 3 + c] = zigzag(wr.corr(a.vals[p * 3 + c], p ? a.vals[(p - 1) * 3 + c] : 0));
  w.u8(1);
  encode_symbols(w, symbols, 3, opt.force_scheme, opt.compression_level);
  w.i32(wr.mn); w.i32(wr.mx);
  write_attribute_transform(w, a);
  out.swap(w.d);
}

}  // namespace synth
