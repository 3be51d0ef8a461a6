// oracle/drc_oracle.cpp
// -----------------------------------------------------------------------------
// TEST INFRASTRUCTURE ONLY.  CPU restatement (scalar, single thread) of the
// Draco v2.2 mesh-decode path that B3zaleel/draco-sharp transcribes.  It is the
// checker the HIP path is compared against; nothing in the product
// (draco-sharp_amd/) links, imports or calls it.  Only tests/, smoke() and the
// cpu_baseline leg of bench.py may use it.
//
// Parity pin: tests/golden/house_04.obj.drc (the reference's own sample,
// src/Draco.Examples/Samples/house_04.obj.drc) decoded by this file reproduces
// house_04.obj (tests/test_oracle_golden.py); the L0 known answers of
// tests/Draco.UnitTests are checked in tests/test_oracle_primitives.py.
// Stages the fixture does not exercise (tagged symbols, standard traversal,
// octahedral normals, delta) are pinned only by encode->decode round trips.
//
// Every function cites the reference file:line it restates (paths relative to
// /root/reference/src/Draco/).  Where the C# as written cannot run (SURVEY.md
// Appendix B, D-1..D-21) the restatement follows the Draco bitstream semantics
// the C# is transcribing and says so.
// -----------------------------------------------------------------------------
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

static const uint32_t kInvalid = 0xFFFFFFFFu;  // IO/Constants.cs:51-54

enum { ERR_INVALID_DATA = 1, ERR_NOT_IMPLEMENTED = 2 };

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
// IO/Extensions/Assertions.cs:5-24 -> InvalidDataException
static inline void require(bool ok, const char *msg) {
  if (!ok) throw Error(ERR_INVALID_DATA, msg);
}

// ---------------------------------------------------------------- byte/bit IO
// IO/DecoderBuffer.cs:26-42 (varint), :51-120 (scalars), :138-184 (bit mode).
// Bit sections are LSB-first and end at ceil(bits/8) (spec; D-2, D-20).
struct Buffer {
  const uint8_t *d = nullptr;
  size_t n = 0, pos = 0;
  bool bit_mode = false;
  size_t bit_base = 0;
  uint64_t bit_off = 0;

  Buffer() {}
  Buffer(const uint8_t *p, size_t len) : d(p), n(len) {}
  void need(size_t k) const { require(pos + k <= n && pos + k >= pos, "unexpected end of stream"); }
  uint8_t u8() { need(1); return d[pos++]; }
  int8_t i8() { return (int8_t)u8(); }
  uint16_t u16() { need(2); uint16_t v = (uint16_t)(d[pos] | (d[pos + 1] << 8)); pos += 2; return v; }
  uint32_t u32() {
    need(4);
    uint32_t v = (uint32_t)d[pos] | ((uint32_t)d[pos + 1] << 8) | ((uint32_t)d[pos + 2] << 16) | ((uint32_t)d[pos + 3] << 24);
    pos += 4;
    return v;
  }
  int32_t i32() { return (int32_t)u32(); }
  float f32() { uint32_t v = u32(); float f; memcpy(&f, &v, 4); return f; }
  uint64_t varint() {  // DecoderBuffer.cs:26-42
    uint64_t r = 0; int shift = 0;
    for (;;) {
      uint8_t b = u8();
      require(shift < 64, "varint too long");
      r |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
    }
    return r;
  }
  const uint8_t *bytes(size_t k) { need(k); const uint8_t *p = d + pos; pos += k; return p; }
  // DecoderBuffer.cs:156-170
  void start_bits(bool decode_size, uint64_t *size) {
    if (decode_size) *size = varint();
    bit_mode = true; bit_base = pos; bit_off = 0;
  }
  // DecoderBuffer.cs:138-154 (without the (byte) truncation, D-2)
  uint32_t bits(int count) {
    uint32_t v = 0;
    for (int i = 0; i < count; ++i) {
      size_t byte = bit_base + (size_t)(bit_off >> 3);
      require(byte < n, "bit read past end of stream");
      v |= (uint32_t)((d[byte] >> (bit_off & 7)) & 1) << i;
      ++bit_off;
    }
    return v;
  }
  // DecoderBuffer.cs:172-175 + spec: consume ceil(bits/8) bytes
  void end_bits() { bit_mode = false; pos = bit_base + (size_t)((bit_off + 7) / 8); }
};

// IO/BitUtilities.cs:72-81,94-103
static inline int32_t zigzag_decode(uint32_t s) {
  return (s & 1) ? -(int32_t)(s >> 1) - 1 : (int32_t)(s >> 1);
}
static inline int msb(uint32_t v) {  // BitUtilities.MostSignificantBit
  int r = 0;
  while (v >>= 1) ++r;
  return r;
}
// IO/Core/MathUtilities.cs:5-25
static inline uint64_t int_sqrt(uint64_t number) {
  if (number == 0) return 0;
  uint64_t act = number, root = 1;
  while (act >= 2) { root *= 2; act /= 4; }
  do { root = (root + number / root) / 2; } while (root * root > number);
  return root;
}

// ------------------------------------------------------------------- entropy
// IO/Entropy/AnsDecoder.cs:12-56 + IO/BitCoders/RAnsBitDecoder.cs:12-24 (rABS).
struct RabsDecoder {
  const uint8_t *buf = nullptr;
  int off = 0;
  uint32_t state = 0;
  uint8_t prob_zero = 0;
  void start(Buffer &b) {                 // RAnsBitDecoder.cs:12-19
    prob_zero = b.u8();
    uint64_t size = b.varint();
    require(size <= b.n - b.pos, "rABS size exceeds stream");
    buf = b.bytes((size_t)size);
    read_init((int)size);
  }
  void read_init(int offset) {            // AnsDecoder.cs:12-40 (D-12: offset-1)
    require(offset >= 1, "rABS stream is empty");
    uint32_t x = buf[offset - 1] >> 6;
    if (x == 0) { off = offset - 1; state = buf[offset - 1] & 0x3F; }
    else if (x == 1) { require(offset >= 2, "rABS too short"); off = offset - 2; state = ((uint32_t)buf[offset - 2] | ((uint32_t)buf[offset - 1] << 8)) & 0x3FFF; }
    else if (x == 2) { require(offset >= 3, "rABS too short"); off = offset - 3; state = ((uint32_t)buf[offset - 3] | ((uint32_t)buf[offset - 2] << 8) | ((uint32_t)buf[offset - 1] << 16)) & 0x3FFFFF; }
    else require(false, "invalid rABS tail");
    state += 4096;                        // Constants.cs:118 DracoAnsLBase
    require(state < 4096u * 256u, "invalid rABS state");
  }
  uint32_t next() {                       // AnsDecoder.cs:42-56
    uint32_t p = 256u - prob_zero;
    if (state < 4096 && off > 0) state = state * 256 + buf[--off];
    uint32_t x = state, quot = x >> 8, rem = x & 255, xn = quot * p;
    bool val = rem < p;
    state = val ? xn + rem : x - xn - p;
    return val ? 1u : 0u;
  }
};

// IO/Entropy/RAnsSymbolDecoder.cs:12-59 + RAnsDecoder.cs:20-99 + RAnsSymbolCoding.cs:10-27
struct RansSymbolDecoder {
  int precision_bits = 0;
  uint32_t precision = 0, l_base = 0;
  uint32_t num_symbols = 0;
  std::vector<uint32_t> prob, cum, lut;
  const uint8_t *buf = nullptr;
  int off = 0;
  uint32_t state = 0;

  static int precision_for(int max_bit_length) {  // RAnsSymbolCoding.cs:10-27
    int p = (3 * max_bit_length) / 2;
    return p < 12 ? 12 : (p > 20 ? 20 : p);
  }
  void create(Buffer &b, int max_bit_length) {    // RAnsSymbolDecoder.cs:12-51
    precision_bits = precision_for(max_bit_length);
    precision = 1u << precision_bits;
    l_base = precision * 4;                       // RAnsDecoder.cs:17
    uint64_t ns = b.varint();
    require(ns <= (1u << 20), "too many rANS symbols");
    num_symbols = (uint32_t)ns;
    prob.assign(num_symbols, 0);
    for (uint32_t i = 0; i < num_symbols; ++i) {
      uint8_t pd = b.u8();
      int token = pd & 3;
      if (token == 3) {
        uint32_t offset = pd >> 2;
        require(i + offset < num_symbols, "zero run past table end");
        for (uint32_t j = 0; j < offset + 1; ++j) prob[i + j] = 0;
        i += offset;
      } else {
        uint32_t p = pd >> 2;
        for (int k = 0; k < token; ++k) p |= (uint32_t)b.u8() << (8 * (k + 1) - 2);
        prob[i] = p;
      }
    }
    build_lut();
  }
  void build_lut() {                              // RAnsDecoder.cs:69-88
    lut.assign(precision, 0);
    cum.assign(num_symbols, 0);
    uint32_t c = 0, act = 0;
    for (uint32_t i = 0; i < num_symbols; ++i) {
      cum[i] = c;
      c += prob[i];
      require(c <= precision, "invalid probability table");
      for (uint32_t j = act; j < c; ++j) lut[j] = i;
      act = c;
    }
    if (num_symbols) require(c == precision, "invalid probability table");
  }
  void start(Buffer &b) {                         // RAnsSymbolDecoder.cs:53-59
    uint64_t size = b.varint();
    require(size <= b.n - b.pos, "rANS size exceeds stream");
    buf = b.bytes((size_t)size);
    read_init((int)size);
  }
  void read_init(int offset) {                    // RAnsDecoder.cs:20-54
    require(offset >= 1, "rANS stream is empty");
    uint32_t x = buf[offset - 1] >> 6;
    if (x == 0) { off = offset - 1; state = buf[offset - 1] & 0x3F; }
    else if (x == 1) { require(offset >= 2, "rANS too short"); off = offset - 2; state = ((uint32_t)buf[offset - 2] | ((uint32_t)buf[offset - 1] << 8)) & 0x3FFF; }
    else if (x == 2) { require(offset >= 3, "rANS too short"); off = offset - 3; state = ((uint32_t)buf[offset - 3] | ((uint32_t)buf[offset - 2] << 8) | ((uint32_t)buf[offset - 1] << 16)) & 0x3FFFFF; }
    else { require(offset >= 4, "rANS too short"); off = offset - 4; state = ((uint32_t)buf[offset - 4] | ((uint32_t)buf[offset - 3] << 8) | ((uint32_t)buf[offset - 2] << 16) | ((uint32_t)buf[offset - 1] << 24)) & 0x3FFFFFFF; }
    state += l_base;
    require(state < l_base * 256u, "invalid rANS state");
  }
  uint32_t read() {                               // RAnsDecoder.cs:56-67,90-99
    while (state < l_base && off > 0) state = state * 256 + buf[--off];
    uint32_t quo = state >> precision_bits, rem = state & (precision - 1);
    uint32_t s = lut[rem];
    state = quo * prob[s] + rem - cum[s];
    return s;
  }
};

// IO/Entropy/SymbolDecoding.cs:7-67.  Tagged path follows the spec (D-1).
static void decode_symbols(Buffer &b, uint32_t num_values, int nc, std::vector<uint32_t> &out) {
  out.assign(num_values, 0);
  if (num_values == 0) return;
  uint8_t scheme = b.u8();
  if (scheme == 0) {            // Tagged, SymbolDecoding.cs:30-50
    RansSymbolDecoder tag;
    tag.create(b, 5);
    require(tag.num_symbols > 0, "wrong number of symbols");
    tag.start(b);
    uint64_t dummy;
    b.start_bits(false, &dummy);
    uint32_t vid = 0;
    for (uint32_t i = 0; i < num_values; i += nc) {
      uint32_t bit_length = tag.read();
      require(bit_length <= 32, "tag bit length too large");
      for (int j = 0; j < nc; ++j) {
        require(vid < num_values, "tagged values overflow");
        out[vid++] = b.bits((int)bit_length);
      }
    }
    b.end_bits();
  } else if (scheme == 1) {     // Raw, SymbolDecoding.cs:52-67
    uint8_t max_bit_length = b.u8();
    require(max_bit_length >= 1 && max_bit_length <= 18, "invalid raw max bit length");
    RansSymbolDecoder dec;
    dec.create(b, max_bit_length);
    require(dec.num_symbols > 0, "wrong number of symbols");
    dec.start(b);
    for (uint32_t i = 0; i < num_values; ++i) out[i] = dec.read();
  } else {
    require(false, "unsupported symbol scheme");
  }
}

// -------------------------------------------------------------- corner table
// IO/Mesh/CornerTable.cs:59-82,127-130,174-192,213-245,275-279
struct CornerTable {
  std::vector<uint32_t> opp, c2v, vcorner;
  uint32_t num_faces() const { return (uint32_t)(c2v.size() / 3); }
  uint32_t num_corners() const { return (uint32_t)c2v.size(); }
  uint32_t num_vertices() const { return (uint32_t)vcorner.size(); }
  static uint32_t next(uint32_t c) { return c == kInvalid ? c : ((c + 1) % 3 ? c + 1 : c - 2); }
  static uint32_t prev(uint32_t c) { return c == kInvalid ? c : (c % 3 ? c - 1 : c + 2); }
  uint32_t opposite(uint32_t c) const { return c == kInvalid ? c : opp[c]; }
  uint32_t vertex(uint32_t c) const { return (c == kInvalid || c >= c2v.size()) ? kInvalid : c2v[c]; }
  uint32_t left_most(uint32_t v) const { return vcorner[v]; }
  uint32_t swing_right(uint32_t c) const { return prev(opposite(prev(c))); }
  uint32_t swing_left(uint32_t c) const { return next(opposite(next(c))); }
  uint32_t right_corner(uint32_t c) const { return c == kInvalid ? c : opposite(next(c)); }
  uint32_t left_corner(uint32_t c) const { return c == kInvalid ? c : opposite(prev(c)); }
  bool is_on_boundary(uint32_t v) const { return swing_left(left_most(v)) == kInvalid; }
  uint32_t add_vertex() { vcorner.push_back(kInvalid); return (uint32_t)vcorner.size() - 1; }
  void set_opp(uint32_t a, uint32_t b) { if (a != kInvalid) opp[a] = b; if (b != kInvalid) opp[b] = a; }
};

// IO/Mesh/MeshAttributeCornerTable.cs:19-30 (ctor), :80-93 (AddSeamEdge),
// :95-155 (RecomputeVertices), :157-215 (accessors).
struct AttrCornerTable {
  const CornerTable *ct = nullptr;
  std::vector<uint8_t> edge_seam, vert_seam;
  std::vector<uint32_t> c2v, v2lm;
  bool no_interior_seams = true;
  void init(const CornerTable *t) {
    ct = t;
    edge_seam.assign(t->num_corners(), 0);
    vert_seam.assign(t->num_vertices(), 0);
    c2v.assign(t->num_corners(), kInvalid);
    v2lm.clear();
    no_interior_seams = true;
  }
  void add_seam_edge(uint32_t c) {
    edge_seam[c] = 1;
    vert_seam[ct->vertex(CornerTable::next(c))] = 1;
    vert_seam[ct->vertex(CornerTable::prev(c))] = 1;
    uint32_t o = ct->opposite(c);
    if (o != kInvalid) {
      no_interior_seams = false;
      edge_seam[o] = 1;
      vert_seam[ct->vertex(CornerTable::next(o))] = 1;
      vert_seam[ct->vertex(CornerTable::prev(o))] = 1;
    }
  }
  static uint32_t next(uint32_t c) { return CornerTable::next(c); }
  static uint32_t prev(uint32_t c) { return CornerTable::prev(c); }
  uint32_t opposite(uint32_t c) const { return (c == kInvalid || edge_seam[c]) ? kInvalid : ct->opposite(c); }
  uint32_t vertex(uint32_t c) const { return c == kInvalid ? kInvalid : c2v[c]; }
  uint32_t left_most(uint32_t v) const { return v2lm[v]; }
  uint32_t swing_right(uint32_t c) const { return prev(opposite(prev(c))); }
  uint32_t swing_left(uint32_t c) const { return next(opposite(next(c))); }
  uint32_t right_corner(uint32_t c) const { return opposite(next(c)); }
  uint32_t left_corner(uint32_t c) const { return opposite(prev(c)); }
  bool is_on_boundary(uint32_t v) const {
    uint32_t c = left_most(v);
    return c == kInvalid || swing_left(c) == kInvalid;
  }
  bool is_corner_on_seam(uint32_t c) const { return vert_seam[ct->vertex(c)] != 0; }
  uint32_t num_vertices() const { return (uint32_t)v2lm.size(); }
  uint32_t num_faces() const { return ct->num_faces(); }
  // MeshAttributeCornerTable.cs:95-155 with mesh==null.  The left-most corner
  // recorded for a vertex opened behind a seam is the corner where it opens
  // (Draco semantics; the C# line 147 stores firstC, which changes nothing
  // observable on the decode path: both satisfy IsOnBoundary).
  void recompute_vertices() {
    v2lm.clear();
    uint32_t num_new = 0;
    for (uint32_t v = 0; v < ct->num_vertices(); ++v) {
      uint32_t c = ct->left_most(v);
      if (c == kInvalid) continue;
      uint32_t first_vert = num_new++;
      uint32_t first_c = c;
      if (vert_seam[v]) {
        uint32_t act = swing_left(first_c);
        while (act != kInvalid) {
          first_c = act;
          act = swing_left(act);
          require(act != c, "attribute seam loop");
        }
      }
      c2v[first_c] = first_vert;
      v2lm.push_back(first_c);
      uint32_t act = ct->swing_right(first_c);
      while (act != kInvalid && act != first_c) {
        if (edge_seam[CornerTable::next(act)]) {
          first_vert = num_new++;
          v2lm.push_back(act);
        }
        c2v[act] = first_vert;
        act = ct->swing_right(act);
      }
    }
  }
};

// IO/Attributes/MeshAttributeIndicesEncodingData.cs:5-19
struct EncodingData {
  std::vector<uint32_t> data_to_corner;
  std::vector<int32_t> vertex_to_data;
  int num_values = 0;
  void init(size_t nv) { vertex_to_data.assign(nv, -1); data_to_corner.clear(); num_values = 0; }
};

// IO/Mesh/Traverser/DepthFirstTraverser.cs:9-99 with the observer of
// MeshAttributeIndicesEncodingObserver.cs:14-21 folded in.
template <class CT>
struct DepthFirst {
  const CT &ct;
  const std::vector<int32_t> &corner_to_point;
  EncodingData &ed;
  std::vector<uint32_t> &point_ids;
  std::vector<uint8_t> face_visited, vert_visited;
  std::vector<uint32_t> stack;
  DepthFirst(const CT &t, uint32_t num_verts, const std::vector<int32_t> &c2p, EncodingData &e, std::vector<uint32_t> &pids)
      : ct(t), corner_to_point(c2p), ed(e), point_ids(pids) {
    face_visited.assign(t.num_faces(), 0);
    vert_visited.assign(num_verts, 0);
  }
  bool face_done(uint32_t f) const { return f == kInvalid || face_visited[f]; }
  void visit_vertex(uint32_t v, uint32_t c) {
    vert_visited[v] = 1;
    point_ids.push_back((uint32_t)corner_to_point[c]);
    ed.data_to_corner.push_back(c);
    ed.vertex_to_data[v] = ed.num_values++;
  }
  void traverse_from(uint32_t corner) {
    if (face_done(corner / 3)) return;
    stack.clear();
    stack.push_back(corner);
    uint32_t nv = ct.vertex(CT::next(corner)), pv = ct.vertex(CT::prev(corner));
    require(nv != kInvalid && pv != kInvalid, "invalid vertex in traversal");
    if (!vert_visited[nv]) visit_vertex(nv, CT::next(corner));
    if (!vert_visited[pv]) visit_vertex(pv, CT::prev(corner));
    while (!stack.empty()) {
      corner = stack.back();
      uint32_t face = corner == kInvalid ? kInvalid : corner / 3;
      if (corner == kInvalid || face_done(face)) { stack.pop_back(); continue; }
      for (;;) {
        face_visited[face] = 1;
        uint32_t v = ct.vertex(corner);
        require(v != kInvalid, "invalid vertex in traversal");
        if (!vert_visited[v]) {
          bool on_boundary = ct.is_on_boundary(v);
          visit_vertex(v, corner);
          if (!on_boundary) {
            corner = ct.right_corner(corner);
            face = corner / 3;
            continue;
          }
        }
        uint32_t rc = ct.right_corner(corner), lc = ct.left_corner(corner);
        uint32_t rf = rc == kInvalid ? kInvalid : rc / 3, lf = lc == kInvalid ? kInvalid : lc / 3;
        if (face_done(rf)) {
          if (face_done(lf)) { stack.pop_back(); break; }
          corner = lc; face = lf;
        } else {
          if (face_done(lf)) { corner = rc; face = rf; }
          else { stack.back() = lc; stack.push_back(rc); break; }
        }
      }
    }
  }
  // IO/Mesh/Traverser/MeshTraversalSequencer.cs:13-31
  void run() {
    for (uint32_t f = 0; f < ct.num_faces(); ++f) traverse_from(3 * f);
  }
};

// IO/Mesh/Traverser/MaxPredictionDegreeTraverser.cs:22-152 with the same observer.  Three stacks by priority: 0 = the
// tip of the face behind the edge is already visited, 1 = it has been seen from two or more faces, 2 = first time
// seen.  D-28: the C# never sizes its prediction-degree list, so its TraverseFromCorner returns at once (:24-27);
// the bitstream sizes it to the vertex count when the traversal starts.
template <class CT>
struct PredictionDegree {
  const CT &ct;
  const std::vector<int32_t> &corner_to_point;
  EncodingData &ed;
  std::vector<uint32_t> &point_ids;
  std::vector<uint8_t> face_visited, vert_visited;
  std::vector<uint32_t> degree;
  std::vector<uint32_t> stacks[3];
  int best = 0;
  PredictionDegree(const CT &t, uint32_t num_verts, const std::vector<int32_t> &c2p, EncodingData &e, std::vector<uint32_t> &pids)
      : ct(t), corner_to_point(c2p), ed(e), point_ids(pids) {
    face_visited.assign(t.num_faces(), 0);
    vert_visited.assign(num_verts, 0);
    degree.assign(num_verts, 0);
  }
  bool face_done(uint32_t f) const { return f == kInvalid || face_visited[f]; }
  void visit_vertex(uint32_t v, uint32_t c) {
    vert_visited[v] = 1;
    point_ids.push_back((uint32_t)corner_to_point[c]);
    ed.data_to_corner.push_back(c);
    ed.vertex_to_data[v] = ed.num_values++;
  }
  uint32_t pop() {                                     // :114-127
    for (int i = best; i < 3; ++i)
      if (!stacks[i].empty()) { uint32_t c = stacks[i].back(); stacks[i].pop_back(); best = i; return c; }
    return kInvalid;
  }
  void push(uint32_t c, int priority) {                // :129-137
    stacks[priority].push_back(c);
    if (priority < best) best = priority;
  }
  int priority_of(uint32_t c) {                        // :139-152
    uint32_t v = ct.vertex(c);
    require(v != kInvalid, "invalid vertex in traversal");
    if (vert_visited[v]) return 0;
    return ++degree[v] > 1 ? 1 : 2;
  }
  void traverse_from(uint32_t corner) {                // :22-112
    if (degree.empty()) return;
    stacks[0].push_back(corner);
    best = 0;
    uint32_t nv = ct.vertex(CT::next(corner)), pv = ct.vertex(CT::prev(corner)), tv = ct.vertex(corner);
    require(nv != kInvalid && pv != kInvalid && tv != kInvalid, "invalid vertex in traversal");
    if (!vert_visited[nv]) visit_vertex(nv, CT::next(corner));
    if (!vert_visited[pv]) visit_vertex(pv, CT::prev(corner));
    if (!vert_visited[tv]) visit_vertex(tv, corner);
    while ((corner = pop()) != kInvalid) {
      if (face_done(corner / 3)) continue;
      for (;;) {
        face_visited[corner / 3] = 1;
        uint32_t v = ct.vertex(corner);
        require(v != kInvalid, "invalid vertex in traversal");
        if (!vert_visited[v]) visit_vertex(v, corner);
        uint32_t rc = ct.right_corner(corner), lc = ct.left_corner(corner);
        bool right_done = face_done(rc == kInvalid ? kInvalid : rc / 3), left_done = face_done(lc == kInvalid ? kInvalid : lc / 3);
        if (!left_done) {
          int pr = priority_of(lc);
          if (right_done && pr <= best) { corner = lc; continue; }
          push(lc, pr);
        }
        if (!right_done) {
          int pr = priority_of(rc);
          if (pr <= best) { corner = rc; continue; }
          push(rc, pr);
        }
        break;
      }
    }
  }
  void run() {
    for (uint32_t f = 0; f < ct.num_faces(); ++f) traverse_from(3 * f);
  }
};

// ---------------------------------------------------------------- attributes
// IO/Attributes/GeometryAttribute.cs:8-67 + PointAttribute.cs:5-63 (subset)
struct Attribute {
  int att_type = 0, data_type = 0, nc = 0, normalized = 0;
  uint32_t unique_id = 0;
  int seq_type = 0;                       // SequentialAttributeEncoderType
  uint32_t num_entries = 0;
  std::vector<uint8_t> values;            // final format, AoS
  std::vector<uint32_t> point_map;        // point -> entry (explicit)
  std::vector<int32_t> portable;          // int32 AoS, num_entries*nc_portable
  int nc_portable = 0;
  std::vector<uint32_t> symbols;          // raw entropy-decoded symbols (diagnostics)
  int pred_method = -2, pred_transform = -1;
  // transform params
  std::vector<float> q_min; float q_range = 0; int q_bits = 0;   // quantization
  int oct_bits = 0;                                              // normals
  int decoder_id = -1;
};

static int data_type_length(int dt) {  // IO/Constants.cs:134-150
  switch (dt) {
    case 1: case 2: case 11: return 1;
    case 3: case 4: return 2;
    case 5: case 6: case 9: return 4;
    case 7: case 8: case 10: return 8;
    default: return 0;
  }
}

// IO/Attributes/OctahedronToolBox.cs
struct OctaToolBox {
  int q = -1, max_q = -1, max_value = -1, center = -1;
  float dequant_scale = 1.0f;
  void set_bits(int bits) {               // :13-21
    require(bits >= 2 && bits <= 30, "invalid octahedron quantization bits");
    q = bits; max_q = (1 << bits) - 1; max_value = max_q - 1;
    dequant_scale = 2.0f / (float)max_value; center = max_value / 2;
  }
  bool in_diamond(int s, int t) const {   // :144-150
    return (uint32_t)std::abs(s) + (uint32_t)std::abs(t) <= (uint32_t)center;
  }
  void invert_diamond(int &s, int &t) const {   // :152-196
    int sign_s, sign_t;
    if (s >= 0 && t >= 0) { sign_s = 1; sign_t = 1; }
    else if (s <= 0 && t <= 0) { sign_s = -1; sign_t = -1; }
    else { sign_s = s > 0 ? 1 : -1; sign_t = t > 0 ? 1 : -1; }
    int cs = sign_s * center, ctt = sign_t * center;
    int us = s + s - cs, ut = t + t - ctt;
    int tmp = us;
    if (sign_s * sign_t >= 0) { us = -ut; ut = -tmp; } else { us = ut; ut = tmp; }
    us += cs; ut += ctt;
    s = us / 2; t = ut / 2;
  }
  int mod_max(int x) const {              // :206-213
    if (x > center) return x - max_q;
    return x < -center ? x + max_q : x;
  }
  // :139-142,220-239 with D-8 fixed (z*z).  float32 arithmetic with one double
  // step, exactly as the C# expression types dictate.
  void to_unit_vector(int s, int t, float out[3]) const {
    float y = (float)s * dequant_scale - 1.0f;
    float z = (float)t * dequant_scale - 1.0f;
    float x = 1.0f - std::fabs(y) - std::fabs(z);
    float x_off = -x < 0 ? 0 : -x;
    y += y < 0 ? x_off : -x_off;
    z += z < 0 ? x_off : -x_off;
    float norm2 = x * x + y * y + z * z;
    if ((double)norm2 < 1e-6) { out[0] = out[1] = out[2] = 0; return; }
    double d = 1.0 / std::sqrt((double)norm2);   // 1.0f / Math.Sqrt(...) is double
    out[0] = (float)((double)x * d); out[1] = (float)((double)y * d); out[2] = (float)((double)z * d);
  }
};

// Prediction transforms -------------------------------------------------------
struct Transform {
  int type = 1;                 // 1 wrap, 2 normal-oct, 3 normal-oct canonicalized
  int nc = 0;
  int32_t wmin = 0, wmax = 0, max_dif = 0;   // wrap
  OctaToolBox oct;
  bool corrections_positive() const { return type == 2 || type == 3; }
  // PredictionSchemeWrapDecodingTransform.cs:69-75 + WrapTransform.cs:88-100;
  // NormalOctahedron*DecodingTransform (D-19: 0 is legal; v2.2: non-canonical
  // reads only max_q, canonical reads max_q + centre).
  void decode_data(Buffer &b) {
    if (type == 1) {
      wmin = b.i32(); wmax = b.i32();
      require(wmin <= wmax, "wrap min > max");
      int64_t dif = (int64_t)wmax - (int64_t)wmin;
      require(dif >= 0 && dif < 0x7FFFFFFF, "wrap range overflow");
      max_dif = (int32_t)(1 + dif);
    } else {
      int32_t max_q = b.i32();
      if (type == 3) (void)b.i32();
      require(max_q > 0 && (max_q % 2) == 1, "invalid max quantized value");
      oct.set_bits(msb((uint32_t)max_q) + 1);
    }
  }
  // PredictionSchemeWrapDecodingTransform.cs:46-67 / WrapTransform.cs:67-86
  void original(const int32_t *pred, const int32_t *corr, int32_t *out) const {
    if (type == 1) {
      for (int i = 0; i < nc; ++i) {
        int32_t p = pred[i] > wmax ? wmax : (pred[i] < wmin ? wmin : pred[i]);
        int32_t o = (int32_t)((uint32_t)p + (uint32_t)corr[i]);
        if (o > wmax) o -= max_dif; else if (o < wmin) o += max_dif;
        out[i] = o;
      }
    } else if (type == 2) {   // PredictionSchemeNormalOctahedronDecodingTransform.cs:47-76
      int c = oct.center;
      int ps = pred[0] - c, pt = pred[1] - c;
      bool in_d = oct.in_diamond(ps, pt);
      if (!in_d) oct.invert_diamond(ps, pt);
      int os = oct.mod_max((int32_t)((uint32_t)ps + (uint32_t)corr[0]));
      int ot = oct.mod_max((int32_t)((uint32_t)pt + (uint32_t)corr[1]));
      if (!in_d) oct.invert_diamond(os, ot);
      out[0] = os + c; out[1] = ot + c;
    } else {                  // ...CanonicalizedDecodingTransform.cs:48-84, ...CanonicalizedTransform.cs:43-89
      int c = oct.center;
      int ps = pred[0] - c, pt = pred[1] - c;
      bool in_d = oct.in_diamond(ps, pt);
      if (!in_d) oct.invert_diamond(ps, pt);
      bool bottom_left = (ps == 0 && pt == 0) || (ps < 0 && pt <= 0);
      int rot;
      if (ps == 0) rot = pt == 0 ? 0 : (pt > 0 ? 3 : 1);
      else if (ps > 0) rot = pt >= 0 ? 2 : 1;
      else rot = pt <= 0 ? 0 : 3;
      if (!bottom_left) rotate(ps, pt, rot);
      int os = oct.mod_max((int32_t)((uint32_t)ps + (uint32_t)corr[0]));
      int ot = oct.mod_max((int32_t)((uint32_t)pt + (uint32_t)corr[1]));
      if (!bottom_left) rotate(os, ot, (4 - rot) % 4);
      if (!in_d) oct.invert_diamond(os, ot);
      out[0] = os + c; out[1] = ot + c;
    }
  }
  static void rotate(int &x, int &y, int rot) {
    int a = x, b = y;
    switch (rot) {
      case 1: x = b; y = -a; break;
      case 2: x = -a; y = -b; break;
      case 3: x = -b; y = a; break;
      default: break;
    }
  }
};

// PredictionSchemeDeltaDecoder.cs:23-37
static void delta_original(const Transform &tr, const std::vector<int32_t> &corr, int nc, std::vector<int32_t> &out) {
  size_t n = corr.size();
  out.assign(n, 0);
  std::vector<int32_t> zero(nc, 0);
  if (n == 0) return;
  tr.original(zero.data(), corr.data(), out.data());
  for (size_t i = nc; i < n; i += nc) tr.original(&out[i - nc], &corr[i], &out[i]);
}

// MeshPredictionSchemeParallelogramDecoder.cs:29-54,56-89
template <class CT>
static void parallelogram_original(const Transform &tr, const CT &ct, const EncodingData &ed,
                                   const std::vector<int32_t> &corr, int nc, std::vector<int32_t> &out) {
  size_t n = corr.size();
  out.assign(n, 0);
  if (n == 0) return;
  std::vector<int32_t> pred(nc, 0);
  tr.original(pred.data(), corr.data(), out.data());
  size_t entries = ed.data_to_corner.size();
  require(entries * nc <= n, "parallelogram: fewer values than entries");
  for (size_t p = 1; p < entries; ++p) {
    uint32_t ci = ed.data_to_corner[p];
    uint32_t oci = ct.opposite(ci);
    bool ok = false;
    if (oci != kInvalid) {
      int32_t vo = ed.vertex_to_data[ct.vertex(oci)];
      int32_t vn = ed.vertex_to_data[ct.vertex(CT::next(oci))];
      int32_t vp = ed.vertex_to_data[ct.vertex(CT::prev(oci))];
      if (vo >= 0 && vn >= 0 && vp >= 0 && vo < (int32_t)p && vn < (int32_t)p && vp < (int32_t)p) {
        for (int c = 0; c < nc; ++c)
          pred[c] = (int32_t)((uint32_t)out[vn * nc + c] + (uint32_t)out[vp * nc + c] - (uint32_t)out[vo * nc + c]);
        ok = true;
      }
    }
    if (ok) tr.original(pred.data(), &corr[p * nc], &out[p * nc]);
    else tr.original(&out[(p - 1) * nc], &corr[p * nc], &out[p * nc]);
  }
}

// MeshPredictionSchemeParallelogramDecoder.cs:56-89 (TryComputeParallelogramPrediction)
template <class CT>
static bool parallelogram_prediction(const CT &ct, const EncodingData &ed, size_t p, uint32_t ci,
                                     const std::vector<int32_t> &out, int nc, int32_t *pred) {
  uint32_t oci = ct.opposite(ci);
  if (oci == kInvalid) return false;
  uint32_t a = ct.vertex(oci), b = ct.vertex(CT::next(oci)), c = ct.vertex(CT::prev(oci));
  if (a == kInvalid || b == kInvalid || c == kInvalid) return false;
  int32_t vo = ed.vertex_to_data[a], vn = ed.vertex_to_data[b], vp = ed.vertex_to_data[c];
  if (!(vo >= 0 && vn >= 0 && vp >= 0 && vo < (int32_t)p && vn < (int32_t)p && vp < (int32_t)p)) return false;
  for (int k = 0; k < nc; ++k)
    pred[k] = (int32_t)((uint32_t)out[vn * nc + k] + (uint32_t)out[vp * nc + k] - (uint32_t)out[vo * nc + k]);
  return true;
}

// MeshPredictionSchemeMultiParallelogramDecoder.cs:24-73: the average of the parallelograms met while swinging
// right from the entry's corner.  D-27: the C# never clears the running sum between entries; the bitstream does.
template <class CT>
static void multi_parallelogram_original(const Transform &tr, const CT &ct, const EncodingData &ed,
                                         const std::vector<int32_t> &corr, int nc, std::vector<int32_t> &out) {
  size_t n = corr.size();
  out.assign(n, 0);
  if (n == 0) return;
  std::vector<int32_t> pred(nc, 0), one(nc, 0);
  tr.original(pred.data(), corr.data(), out.data());
  size_t entries = ed.data_to_corner.size();
  require(entries * nc <= n, "multi-parallelogram: fewer values than entries");
  const size_t max_steps = (size_t)ct.num_faces() * 3 + 1;
  for (size_t p = 1; p < entries; ++p) {
    uint32_t start = ed.data_to_corner[p], c = start;
    int count = 0;
    size_t steps = 0;
    std::fill(pred.begin(), pred.end(), 0);
    while (c != kInvalid) {
      require(++steps <= max_steps, "multi-parallelogram: corner fan does not close");
      if (parallelogram_prediction(ct, ed, p, c, out, nc, one.data())) {
        for (int k = 0; k < nc; ++k) pred[k] = (int32_t)((uint32_t)pred[k] + (uint32_t)one[k]);
        ++count;
      }
      c = ct.swing_right(c);
      if (c == start) c = kInvalid;
    }
    if (count == 0) tr.original(&out[(p - 1) * nc], &corr[p * nc], &out[p * nc]);
    else {
      for (int k = 0; k < nc; ++k) pred[k] /= count;
      tr.original(pred.data(), &corr[p * nc], &out[p * nc]);
    }
  }
}

// MeshPredictionSchemeConstrainedMultiParallelogramDecoder.cs:28-108 with D-14 resolved to the bitstream: up to four
// parallelograms (left swing first, then right from the start), each kept or dropped by its crease flag; the flags
// live in one list per context (= number of parallelograms found - 1).
template <class CT>
static void constrained_multi_parallelogram_original(const Transform &tr, const CT &ct, const EncodingData &ed,
                                                     const std::vector<int32_t> &corr, int nc,
                                                     const std::vector<uint8_t> crease[4], std::vector<int32_t> &out) {
  size_t n = corr.size();
  out.assign(n, 0);
  if (n == 0) return;
  std::vector<int32_t> preds[4], multi(nc, 0);
  for (auto &v : preds) v.assign(nc, 0);
  tr.original(preds[0].data(), corr.data(), out.data());
  size_t entries = ed.data_to_corner.size();
  require(entries * nc <= n, "constrained multi-parallelogram: fewer values than entries");
  size_t crease_pos[4] = {0, 0, 0, 0};
  const size_t max_steps = (size_t)ct.num_faces() * 3 + 1;
  for (size_t p = 1; p < entries; ++p) {
    uint32_t start = ed.data_to_corner[p], c = start;
    int found = 0;
    bool first_pass = true;
    size_t steps = 0;
    while (c != kInvalid) {
      require(++steps <= max_steps, "constrained multi-parallelogram: corner fan does not close");
      if (parallelogram_prediction(ct, ed, p, c, out, nc, preds[found].data())) {
        if (++found == 4) break;
      }
      c = first_pass ? ct.swing_left(c) : ct.swing_right(c);
      if (c == start) break;
      if (c == kInvalid && first_pass) { first_pass = false; c = ct.swing_right(start); }
    }
    int used = 0;
    if (found > 0) {
      std::fill(multi.begin(), multi.end(), 0);
      for (int i = 0; i < found; ++i) {
        int context = found - 1;
        size_t pos = crease_pos[context]++;
        require(pos < crease[context].size(), "constrained multi-parallelogram: ran out of crease flags");
        if (!crease[context][pos]) {
          ++used;
          for (int k = 0; k < nc; ++k) multi[k] = (int32_t)((uint32_t)multi[k] + (uint32_t)preds[i][k]);
        }
      }
    }
    if (used == 0) tr.original(&out[(p - 1) * nc], &corr[p * nc], &out[p * nc]);
    else {
      for (int k = 0; k < nc; ++k) multi[k] /= used;
      tr.original(multi.data(), &corr[p * nc], &out[p * nc]);
    }
  }
}

// MeshPredictionSchemeTexCoordsPortableDecoder.cs:50-85 +
// MeshPredictionSchemeTexCoordsPortablePredictor.cs:46-150 (fallback chain kept
// exactly as written there, which is also what upstream Draco does).
template <class CT>
static void texcoords_portable_original(const Transform &tr, const CT &ct, const EncodingData &ed,
                                        const std::vector<int32_t> &corr, int nc,
                                        const std::vector<uint32_t> &entry_to_point, const Attribute &pos,
                                        std::vector<uint8_t> &orientations, std::vector<int32_t> &out) {
  require(nc == 2, "texcoord prediction needs 2 components");
  require(pos.nc_portable == 3, "texcoord prediction needs 3-component positions");
  size_t entries = ed.data_to_corner.size();
  out.assign(entries * 2, 0);
  require(corr.size() >= entries * 2, "texcoords: fewer values than entries");
  auto get_pos = [&](int entry, int64_t p[3]) {
    uint32_t point = entry_to_point[entry];
    uint32_t e = pos.point_map.empty() ? point : pos.point_map[point];
    for (int k = 0; k < 3; ++k) p[k] = pos.portable[(size_t)e * 3 + k];
  };
  for (size_t p = 0; p < entries; ++p) {
    int data_id = (int)p;
    uint32_t ci = ed.data_to_corner[p];
    int32_t next_id = ed.vertex_to_data[ct.vertex(CT::next(ci))];
    int32_t prev_id = ed.vertex_to_data[ct.vertex(CT::prev(ci))];
    int32_t pred[2] = {0, 0};
    bool done = false;
    if (prev_id >= 0 && next_id >= 0 && prev_id < data_id && next_id < data_id) {
      int64_t n_uv[2] = {out[next_id * 2], out[next_id * 2 + 1]};
      int64_t p_uv[2] = {out[prev_id * 2], out[prev_id * 2 + 1]};
      if (p_uv[0] == n_uv[0] && p_uv[1] == n_uv[1]) {
        pred[0] = (int32_t)p_uv[0]; pred[1] = (int32_t)p_uv[1];
        done = true;
      } else {
        int64_t tip[3], np[3], pp[3];
        get_pos(data_id, tip); get_pos(next_id, np); get_pos(prev_id, pp);
        int64_t pn[3] = {pp[0] - np[0], pp[1] - np[1], pp[2] - np[2]};
        int64_t pn_norm2 = pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2];
        if (pn_norm2 != 0) {
          int64_t cn[3] = {tip[0] - np[0], tip[1] - np[1], tip[2] - np[2]};
          int64_t cn_dot_pn = pn[0] * cn[0] + pn[1] * cn[1] + pn[2] * cn[2];
          int64_t pn_uv[2] = {p_uv[0] - n_uv[0], p_uv[1] - n_uv[1]};
          int64_t x_uv[2] = {n_uv[0] * pn_norm2 + cn_dot_pn * pn_uv[0], n_uv[1] * pn_norm2 + cn_dot_pn * pn_uv[1]};
          int64_t x_pos[3];
          for (int k = 0; k < 3; ++k) x_pos[k] = np[k] + (cn_dot_pn * pn[k]) / pn_norm2;
          int64_t cx[3] = {tip[0] - x_pos[0], tip[1] - x_pos[1], tip[2] - x_pos[2]};
          uint64_t cx_norm2 = (uint64_t)(cx[0] * cx[0] + cx[1] * cx[1] + cx[2] * cx[2]);
          int64_t cx_uv[2] = {pn_uv[1], -pn_uv[0]};
          int64_t norm = (int64_t)int_sqrt(cx_norm2 * (uint64_t)pn_norm2);
          cx_uv[0] *= norm; cx_uv[1] *= norm;
          require(!orientations.empty(), "texcoords: ran out of orientations");
          bool orientation = orientations.back() != 0;
          orientations.pop_back();
          int64_t pu, pv;
          if (orientation) { pu = (x_uv[0] + cx_uv[0]) / pn_norm2; pv = (x_uv[1] + cx_uv[1]) / pn_norm2; }
          else { pu = (x_uv[0] - cx_uv[0]) / pn_norm2; pv = (x_uv[1] - cx_uv[1]) / pn_norm2; }
          pred[0] = (int32_t)pu; pred[1] = (int32_t)pv;
          done = true;
        }
      }
    }
    if (!done) {
      int data_offset = 0;
      bool zero = false;
      if (prev_id >= 0 && prev_id < data_id) data_offset = prev_id * 2;
      if (next_id >= 0 && next_id < data_id) data_offset = next_id * 2;
      else {
        if (data_id > 0) data_offset = (data_id - 1) * 2;
        else zero = true;
      }
      if (!zero) { pred[0] = out[data_offset]; pred[1] = out[data_offset + 1]; }
    }
    tr.original(pred, &corr[p * 2], &out[p * 2]);
  }
}

// MeshPredictionSchemeGeometricNormalDecoder.cs:44-82 +
// MeshPredictionSchemeGeometricNormalPredictorArea.cs:16-63 +
// MeshPredictionSchemeGeometricNormalPredictor.cs:19-33 +
// OctahedronToolBox.cs:28-77,121-137.  The bitstream's arithmetic is kept where
// the C# narrows it (D-9, D-10, D-23..D-25 in DESIGN.md): the area-weighted normal is summed
// and scaled in 64 bits, its third component is normal[2], and the
// canonicalisation multiplies in 64 bits.
template <class CT>
static void geometric_normal_original(const Transform &tr, const CT &ct, const EncodingData &ed,
                                      const std::vector<int32_t> &corr,
                                      const std::vector<uint32_t> &entry_to_point, const Attribute &pos,
                                      const std::vector<uint8_t> &flips, std::vector<int32_t> &out) {
  require(pos.nc_portable == 3, "normal prediction needs 3-component positions");
  size_t entries = ed.data_to_corner.size();
  out.assign(entries * 2, 0);
  require(corr.size() >= entries * 2, "normals: fewer values than entries");
  require(flips.size() >= entries, "normals: fewer flip bits than entries");
  const OctaToolBox &oct = tr.oct;
  const size_t pos_entries = pos.portable.size() / 3;
  auto pos_of_corner = [&](uint32_t c, int64_t p[3]) {
    uint32_t v = ct.vertex(c);
    require(v != kInvalid && v < ed.vertex_to_data.size(), "normals: corner without vertex");
    int32_t d = ed.vertex_to_data[v];
    require(d >= 0 && (size_t)d < entry_to_point.size(), "normals: vertex without data id");
    uint32_t point = entry_to_point[d];
    require(pos.point_map.empty() || point < pos.point_map.size(), "normals: point outside the position map");
    uint32_t e = pos.point_map.empty() ? point : pos.point_map[point];
    require(e < pos_entries, "normals: position entry out of range");
    for (int k = 0; k < 3; ++k) p[k] = pos.portable[(size_t)e * 3 + k];
  };
  const size_t max_steps = (size_t)ct.num_faces() * 3 + 1;
  for (size_t p = 0; p < entries; ++p) {
    uint32_t ci = ed.data_to_corner[p];
    int64_t center[3];
    pos_of_corner(ci, center);
    uint64_t n[3] = {0, 0, 0};               // wrapping 64-bit sums
    // VertexCornersIterator (Mesh/VertexCornersIterator.cs): swing left from the
    // start corner; at a boundary swing right from the start corner.
    uint32_t c = ci;
    bool left = true;
    size_t steps = 0;
    while (c != kInvalid) {
      require(++steps <= max_steps, "normals: corner fan does not close");
      int64_t pn[3], pp[3];
      pos_of_corner(CT::next(c), pn);
      pos_of_corner(CT::prev(c), pp);
      uint64_t a[3], b[3];
      for (int k = 0; k < 3; ++k) { a[k] = (uint64_t)(pn[k] - center[k]); b[k] = (uint64_t)(pp[k] - center[k]); }
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
    // VectorD::AbsSum saturates instead of overflowing
    auto abs_sum64 = [](const int64_t v[3]) {
      uint64_t s = 0;
      for (int k = 0; k < 3; ++k) {
        uint64_t a = v[k] < 0 ? (uint64_t)0 - (uint64_t)v[k] : (uint64_t)v[k];
        if (a > (uint64_t)INT64_MAX || s > (uint64_t)INT64_MAX - a) return (int64_t)INT64_MAX;
        s += a;
      }
      return (int64_t)s;
    };
    const int64_t upper = (int64_t)1 << 29;
    int64_t as = abs_sum64(nv);
    if (as > upper) {
      int64_t q = as / upper;
      for (int k = 0; k < 3; ++k) nv[k] /= q;
    }
    int32_t v3[3] = {(int32_t)nv[0], (int32_t)nv[1], (int32_t)nv[2]};
    // CanonicalizeIntegerVector
    int64_t s3 = std::llabs((int64_t)v3[0]) + std::llabs((int64_t)v3[1]) + std::llabs((int64_t)v3[2]);
    if (s3 == 0) v3[0] = oct.center;
    else {
      v3[0] = (int32_t)(((int64_t)v3[0] * oct.center) / s3);
      v3[1] = (int32_t)(((int64_t)v3[1] * oct.center) / s3);
      int32_t rest = oct.center - std::abs(v3[0]) - std::abs(v3[1]);
      v3[2] = v3[2] >= 0 ? rest : -rest;
    }
    if (flips[p]) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
    // IntegerVectorToQuantizedOctahedralCoords + CanonicalizeOctahedralCoords
    int32_t s, t;
    if (v3[0] >= 0) { s = v3[1] + oct.center; t = v3[2] + oct.center; }
    else {
      s = v3[1] < 0 ? std::abs(v3[2]) : oct.max_value - std::abs(v3[2]);
      t = v3[2] < 0 ? std::abs(v3[1]) : oct.max_value - std::abs(v3[1]);
    }
    if ((s == 0 && t == 0) || (s == 0 && t == oct.max_value) || (s == oct.max_value && t == 0)) { s = oct.max_value; t = oct.max_value; }
    else if (s == 0 && t > oct.center) t = oct.center - (t - oct.center);
    else if (s == oct.max_value && t < oct.center) t = oct.center + (oct.center - t);
    else if (t == oct.max_value && s < oct.center) s = oct.center + (oct.center - s);
    else if (t == 0 && s > oct.center) s = oct.center - (s - oct.center);
    int32_t pred[2] = {s, t};
    tr.original(pred, &corr[p * 2], &out[p * 2]);
  }
}

// ---------------------------------------------------------------------- mesh
struct AttributeData {           // IO/Mesh/DecoderAttributeData.cs
  int decoder_id = -1;
  AttrCornerTable conn;
  bool is_connectivity_used = true;
  EncodingData enc;
  std::vector<uint32_t> seam_corners;
};

struct AttDecoder {              // one attributes decoder (SequentialAttributeDecodersController)
  int att_data_id = -1, element_type = 0, traversal_method = 0;
  std::vector<int> att_ids;
  std::vector<uint32_t> point_ids;   // entry -> point
};

struct Mesh {
  // header (DracoHeader.cs:5-23)
  int major = 0, minor = 0, encoder_type = 0, encoder_method = 0, flags = 0;
  int traversal_type = -1;
  bool is_point_cloud = false;
  bool is_sequential = false;        // sequential mesh: faces stored as point indices, linear attribute sequencing
  // connectivity
  CornerTable ct;
  std::vector<uint8_t> is_vert_hole;
  std::vector<AttributeData> att_data;
  EncodingData pos_enc;
  std::vector<int32_t> faces;        // corner -> point id  (Mesh.cs faces)
  uint32_t num_points = 0;
  uint32_t num_conn_vertices = 0;
  std::vector<uint8_t> eb_symbols;   // decoded Edgebreaker symbols (diagnostics)
  // attributes
  std::vector<Attribute> atts;
  std::vector<AttDecoder> decoders;
  size_t end_pos = 0;                // bytes consumed
};

struct EdgebreakerDecoder {
  Buffer &b;
  Mesh &m;
  int traversal_type;
  // traversal decoder state
  Buffer symbol_buf;                       // standard
  RabsDecoder start_face, seams[64];
  uint32_t num_att_data = 0;
  // valence
  std::vector<uint32_t> vertex_valences;
  std::vector<std::vector<uint32_t>> ctx_symbols;
  std::vector<int> ctx_counters;
  int last_symbol = -1, active_context = -1;
  int predicted_symbol = -1;              // predictive traversal
  RabsDecoder prediction;
  // topology splits
  struct Split { uint32_t source, split, edge; };
  std::vector<Split> splits;

  EdgebreakerDecoder(Buffer &buf, Mesh &mesh, int tt) : b(buf), m(mesh), traversal_type(tt) {}

  // MeshEdgeBreakerDecoder.cs:136-230 (v2.2 branch only)
  void decode_split_events(uint32_t num_faces) {
    uint32_t num = (uint32_t)b.varint();
    if (num > 0) {
      require(num <= num_faces, "too many topology splits");
      int last = 0;
      for (uint32_t i = 0; i < num; ++i) {
        Split s;
        uint32_t delta = (uint32_t)b.varint();
        s.source = delta + (uint32_t)last;
        delta = (uint32_t)b.varint();
        require(delta <= s.source, "split delta larger than source");
        s.split = s.source - delta;
        last = (int)s.source;
        s.edge = 0;
        splits.push_back(s);
      }
      uint64_t dummy;
      b.start_bits(false, &dummy);
      for (uint32_t i = 0; i < num; ++i) splits[i].edge = b.bits(1) & 1;
      b.end_bits();
    }
  }

  // MeshEdgeBreakerTraversalDecoder.cs:27-61 / ...ValenceDecoder.cs:22-69
  void traversal_start(uint32_t num_encoded_vertices_total) {
    if (traversal_type == 0 || traversal_type == 1) {
      // D-3: the symbol section is a size-prefixed byte block read LSB-first.
      uint64_t size = b.varint();
      require(size <= b.n - b.pos, "traversal symbol section exceeds stream");
      symbol_buf = Buffer(b.bytes((size_t)size), (size_t)size);
      uint64_t dummy;
      symbol_buf.start_bits(false, &dummy);
    }
    start_face.start(b);
    require(num_att_data <= 64, "too many attribute data");
    for (uint32_t i = 0; i < num_att_data; ++i) seams[i].start(b);
    if (traversal_type == 1) {             // MeshEdgeBreakerTraversalPredictiveDecoder.cs:19-27
      int32_t num_split_symbols = b.i32();
      require(num_split_symbols >= 0 && (uint32_t)num_split_symbols < num_encoded_vertices_total, "invalid predictive split symbol count");
      vertex_valences.assign(num_encoded_vertices_total, 0);
      prediction.start(b);
    }
    if (traversal_type == 2) {
      vertex_valences.assign(num_encoded_vertices_total, 0);
      ctx_symbols.assign(6, {});
      ctx_counters.assign(6, 0);
      for (int i = 0; i < 6; ++i) {
        uint32_t num = (uint32_t)b.varint();
        require(num <= m.ct.num_faces(), "too many valence context symbols");
        if (num > 0) {
          decode_symbols(b, num, 1, ctx_symbols[i]);
          ctx_counters[i] = (int)num;
        }
      }
    }
  }
  // MeshEdgeBreakerTraversalDecoder.cs:89-99 / ...ValenceDecoder.cs:77-98
  uint32_t decode_symbol() {
    if (traversal_type == 1) {             // MeshEdgeBreakerTraversalPredictiveDecoder.cs:34-46
      if (predicted_symbol != -1 && prediction.next() != 0) { last_symbol = predicted_symbol; return (uint32_t)last_symbol; }
      uint32_t s = symbol_buf.bits(1);
      if (s != 0) s |= symbol_buf.bits(2) << 1;
      last_symbol = (int)s;
      return s;
    }
    if (traversal_type == 0) {
      uint32_t s = symbol_buf.bits(1);
      if (s == 0) return 0;
      return s | (symbol_buf.bits(2) << 1);
    }
    static const uint8_t sym_to_topo[5] = {0, 1, 3, 5, 7};   // Constants.cs:88-95
    if (active_context != -1) {
      int cnt = --ctx_counters[active_context];
      require(cnt >= 0, "valence context exhausted");
      uint32_t sid = ctx_symbols[active_context][cnt];
      require(sid <= 4, "invalid valence symbol");
      last_symbol = sym_to_topo[sid];
    } else {
      last_symbol = 7;   // first symbol is implicitly E for v2.2
    }
    return (uint32_t)last_symbol;
  }
  // ...ValenceDecoder.cs:100-149
  void new_active_corner(uint32_t corner) {
    if (traversal_type == 0) return;
    uint32_t nx = CornerTable::next(corner), pv = CornerTable::prev(corner);
    switch (last_symbol) {
      case 0: case 1:
        vertex_valences[m.ct.vertex(nx)] += 1; vertex_valences[m.ct.vertex(pv)] += 1; break;
      case 5:
        vertex_valences[m.ct.vertex(corner)] += 1; vertex_valences[m.ct.vertex(nx)] += 1; vertex_valences[m.ct.vertex(pv)] += 2; break;
      case 3:
        vertex_valences[m.ct.vertex(corner)] += 1; vertex_valences[m.ct.vertex(nx)] += 2; vertex_valences[m.ct.vertex(pv)] += 1; break;
      case 7:
        vertex_valences[m.ct.vertex(corner)] += 2; vertex_valences[m.ct.vertex(nx)] += 2; vertex_valences[m.ct.vertex(pv)] += 2; break;
      default: break;
    }
    int v = (int)vertex_valences[m.ct.vertex(nx)];
    if (traversal_type == 1) {             // ...PredictiveDecoder.cs:79-92: after C or R, R while the pivot has fewer than six edges
      predicted_symbol = (last_symbol == 0 || last_symbol == 5) ? (v < 6 ? 5 : 0) : -1;
      return;
    }
    int clamped = v < 2 ? 2 : (v > 7 ? 7 : v);
    active_context = clamped - 2;
  }
  void merge_vertices(uint32_t dest, uint32_t src) {   // ...ValenceDecoder.cs:151-154
    if (traversal_type != 0) vertex_valences[dest] += vertex_valences[src];
  }
  // MeshEdgeBreakerDecoder.cs:450-471
  bool is_topology_split(int encoder_symbol_id, int *edge, int *split_id) {
    *edge = -1; *split_id = -1;
    if (splits.empty()) return false;
    if ((int64_t)splits.back().source > (int64_t)encoder_symbol_id) { *split_id = -1; return true; }
    if ((int64_t)splits.back().source != (int64_t)encoder_symbol_id) return false;
    *edge = (int)splits.back().edge;
    *split_id = (int)splits.back().split;
    splits.pop_back();
    return true;
  }

  // MeshEdgeBreakerDecoder.cs:232-442
  int decode_connectivity_symbols(int num_symbols) {
    CornerTable &ct = m.ct;
    std::vector<uint32_t> stack;
    std::map<int, uint32_t> split_active;
    std::vector<uint32_t> invalid_vertices;
    bool remove_invalid = m.att_data.empty();
    size_t max_vertices = m.is_vert_hole.size();
    uint32_t num_faces = 0;
    m.eb_symbols.reserve(num_symbols);
    for (int sid = 0; sid < num_symbols; ++sid) {
      uint32_t face = num_faces++;
      bool check_split = false;
      uint32_t sym = decode_symbol();
      m.eb_symbols.push_back((uint8_t)sym);
      uint32_t corner = 3 * face;
      if (sym == 0) {            // C
        require(!stack.empty(), "C with empty stack");
        uint32_t ca = stack.back();
        uint32_t vx = ct.vertex(CornerTable::next(ca));
        require(vx != kInvalid && vx < ct.num_vertices(), "C: invalid vertex");
        uint32_t lm = ct.left_most(vx);
        require(lm != kInvalid, "C: vertex without corner");
        uint32_t cb = CornerTable::next(lm);
        require(ca != cb, "matched corners must differ");
        require(ct.opposite(ca) == kInvalid && ct.opposite(cb) == kInvalid, "corner already has an opposite");
        ct.set_opp(ca, corner + 1);
        ct.set_opp(cb, corner + 2);
        uint32_t va_prev = ct.vertex(CornerTable::prev(ca));
        uint32_t vb_next = ct.vertex(CornerTable::next(cb));
        require(vx != va_prev && vx != vb_next, "degenerate face");
        ct.c2v[corner] = vx; ct.c2v[corner + 1] = vb_next; ct.c2v[corner + 2] = va_prev;
        ct.vcorner[va_prev] = corner + 2;
        m.is_vert_hole[vx] = 0;
        stack.back() = corner;
      } else if (sym == 5 || sym == 3) {   // R / L
        require(!stack.empty(), "R/L with empty stack");
        uint32_t ca = stack.back();
        require(ct.opposite(ca) == kInvalid, "corner already has an opposite");
        uint32_t oc, cl, cr;
        if (sym == 5) { oc = corner + 2; cl = corner + 1; cr = corner; }
        else { oc = corner + 1; cl = corner; cr = corner + 2; }
        ct.set_opp(oc, ca);
        uint32_t nv = ct.add_vertex();
        require(ct.num_vertices() <= max_vertices, "unexpected number of decoded vertices");
        ct.c2v[oc] = nv;
        ct.vcorner[nv] = oc;
        uint32_t vr = ct.vertex(CornerTable::prev(ca));
        ct.c2v[cr] = vr;
        ct.vcorner[vr] = cr;
        ct.c2v[cl] = ct.vertex(CornerTable::next(ca));
        stack.back() = corner;
        check_split = true;
      } else if (sym == 1) {     // S
        require(!stack.empty(), "S with empty stack");
        uint32_t cb = stack.back();
        stack.pop_back();
        auto it = split_active.find(sid);
        if (it != split_active.end()) stack.push_back(it->second);
        require(!stack.empty(), "S with empty stack");
        uint32_t ca = stack.back();
        require(ca != cb, "matched corners must differ");
        require(ct.opposite(ca) == kInvalid && ct.opposite(cb) == kInvalid, "corner already has an opposite");
        ct.set_opp(ca, corner + 2);
        ct.set_opp(cb, corner + 1);
        uint32_t vp = ct.vertex(CornerTable::prev(ca));
        ct.c2v[corner] = vp;
        ct.c2v[corner + 1] = ct.vertex(CornerTable::next(ca));
        uint32_t vb_prev = ct.vertex(CornerTable::prev(cb));
        ct.c2v[corner + 2] = vb_prev;
        ct.vcorner[vb_prev] = corner + 2;
        uint32_t cn = CornerTable::next(cb);
        uint32_t vn = ct.vertex(cn);
        merge_vertices(vp, vn);
        ct.vcorner[vp] = ct.left_most(vn);
        uint32_t first = cn;
        while (cn != kInvalid) {
          ct.c2v[cn] = vp;
          cn = ct.swing_left(cn);
          require(cn != first, "split loop reached start");
        }
        ct.vcorner[vn] = kInvalid;
        if (remove_invalid) invalid_vertices.push_back(vn);
        stack.back() = corner;
      } else if (sym == 7) {     // E
        uint32_t v0 = ct.add_vertex();
        ct.c2v[corner] = v0;
        ct.c2v[corner + 1] = ct.add_vertex();
        ct.c2v[corner + 2] = ct.add_vertex();
        require(ct.num_vertices() <= max_vertices, "unexpected number of decoded vertices");
        ct.vcorner[v0] = corner; ct.vcorner[v0 + 1] = corner + 1; ct.vcorner[v0 + 2] = corner + 2;
        stack.push_back(corner);
        check_split = true;
      } else {
        require(false, "unknown Edgebreaker symbol");
      }
      new_active_corner(stack.back());
      if (check_split) {
        int enc_id = num_symbols - sid - 1;
        int edge, enc_split;
        while (is_topology_split(enc_id, &edge, &enc_split)) {
          require(enc_split >= 0, "wrong split symbol id");
          uint32_t top = stack.back();
          uint32_t nc = edge == 1 ? CornerTable::next(top) : CornerTable::prev(top);   // 1 = right face edge
          int dec_split = num_symbols - enc_split - 1;
          split_active[dec_split] = nc;
        }
      }
    }
    require(ct.num_vertices() <= max_vertices, "unexpected number of decoded vertices");
    // start faces, MeshEdgeBreakerDecoder.cs:378-415
    while (!stack.empty()) {
      uint32_t corner = stack.back();
      stack.pop_back();
      bool interior = start_face.next() != 0;
      if (interior) {
        require(num_faces < ct.num_faces(), "more faces than expected");
        uint32_t ca = corner;
        uint32_t vn = ct.vertex(CornerTable::next(ca));
        uint32_t cb = CornerTable::next(ct.left_most(vn));
        uint32_t vx = ct.vertex(CornerTable::next(cb));
        uint32_t cc = CornerTable::next(ct.left_most(vx));
        require(corner != cb && corner != cc && cb != cc, "matched corners must differ");
        require(ct.opposite(corner) == kInvalid && ct.opposite(cb) == kInvalid && ct.opposite(cc) == kInvalid, "corner already has an opposite");
        uint32_t vp = ct.vertex(CornerTable::next(cc));
        uint32_t face = num_faces++;
        uint32_t nc = 3 * face;
        ct.set_opp(nc, corner); ct.set_opp(nc + 1, cb); ct.set_opp(nc + 2, cc);
        ct.c2v[nc] = vx; ct.c2v[nc + 1] = vp; ct.c2v[nc + 2] = vn;
        for (int k = 0; k < 3; ++k) m.is_vert_hole[ct.c2v[nc + k]] = 0;
      }
    }
    require(num_faces == ct.num_faces(), "unexpected number of decoded faces");
    // isolated-vertex compaction, MeshEdgeBreakerDecoder.cs:417-441 (D-10: the
    // iterator must start at the vertex's own left-most corner)
    int num_vertices = (int)ct.num_vertices();
    for (uint32_t inv : invalid_vertices) {
      uint32_t src = (uint32_t)num_vertices - 1;
      while (ct.left_most(src) == kInvalid) src = (uint32_t)(--num_vertices) - 1;
      if (src < inv) continue;
      // VertexCornersIterator: swing left from the left-most corner, then (on a
      // boundary) swing right from the start.
      uint32_t start = ct.left_most(src), c = start;
      bool left = true;
      while (c != kInvalid) {
        require(ct.vertex(c) == src, "vertex corner mismatch");
        ct.c2v[c] = inv;
        if (left) {
          c = ct.swing_left(c);
          if (c == kInvalid) { c = ct.swing_right(start); left = false; }
          else if (c == start) c = kInvalid;
        } else {
          c = ct.swing_right(c);
        }
      }
      ct.vcorner[inv] = ct.left_most(src);
      ct.vcorner[src] = kInvalid;
      m.is_vert_hole[inv] = m.is_vert_hole[src];
      m.is_vert_hole[src] = 0;
      num_vertices--;
    }
    return num_vertices;
  }

  // MeshEdgeBreakerDecoder.cs:502-535
  void decode_attribute_seams() {
    CornerTable &ct = m.ct;
    for (uint32_t ci = 0; ci < ct.num_corners(); ci += 3) {
      uint32_t corners[3] = {ci, ci + 1, ci + 2};
      uint32_t src_face = ci / 3;
      for (int k = 0; k < 3; ++k) {
        uint32_t oc = ct.opposite(corners[k]);
        if (oc == kInvalid) {
          for (auto &ad : m.att_data) ad.seam_corners.push_back(corners[k]);
          continue;
        }
        if (oc / 3 < src_face) continue;
        for (size_t i = 0; i < m.att_data.size(); ++i)
          if (seams[i].next()) m.att_data[i].seam_corners.push_back(corners[k]);
      }
    }
  }

  // MeshEdgeBreakerDecoder.cs:537-638
  void assign_points_to_corners(int num_conn_vertices) {
    CornerTable &ct = m.ct;
    m.faces.assign(ct.num_corners(), 0);
    if (m.att_data.empty()) {
      for (uint32_t c = 0; c < ct.num_corners(); ++c) m.faces[c] = (int32_t)ct.c2v[c];
      m.num_points = (uint32_t)num_conn_vertices;
      return;
    }
    std::vector<int32_t> point_to_corner;
    std::vector<int32_t> &c2p = m.faces;
    for (uint32_t v = 0; v < ct.num_vertices(); ++v) {
      uint32_t c = ct.left_most(v);
      if (c == kInvalid) continue;
      uint32_t dedup_first = c;
      if (!m.is_vert_hole[v]) {
        for (auto &ad : m.att_data) {
          if (!ad.conn.is_corner_on_seam(c)) continue;
          uint32_t vid = ad.conn.vertex(c);
          uint32_t act = ct.swing_right(c);
          bool seam_found = false;
          while (act != c) {
            require(act != kInvalid, "open ring on interior vertex");
            if (ad.conn.vertex(act) != vid) { dedup_first = act; seam_found = true; break; }
            act = ct.swing_right(act);
          }
          if (seam_found) break;
        }
      }
      c = dedup_first;
      c2p[c] = (int32_t)point_to_corner.size();
      point_to_corner.push_back((int32_t)c);
      uint32_t prev_c = c;
      c = ct.swing_right(c);
      while (c != kInvalid && c != dedup_first) {
        bool seam = false;
        for (auto &ad : m.att_data)
          if (ad.conn.vertex(c) != ad.conn.vertex(prev_c)) { seam = true; break; }
        if (seam) { c2p[c] = (int32_t)point_to_corner.size(); point_to_corner.push_back((int32_t)c); }
        else c2p[c] = c2p[prev_c];
        prev_c = c;
        c = ct.swing_right(c);
      }
    }
    m.num_points = (uint32_t)point_to_corner.size();
  }

  // MeshEdgeBreakerDecoder.cs:25-134
  void decode_connectivity() {
    uint32_t num_encoded_vertices = (uint32_t)b.varint();
    uint32_t num_faces = (uint32_t)b.varint();
    require(num_faces <= 0x7FFFFFFFu / 3, "too many faces");
    require(num_encoded_vertices <= num_faces * 3, "more vertices than 3*faces");
    uint32_t min_face_edges = 3 * num_faces / 2;
    uint64_t nv64 = num_encoded_vertices;
    uint64_t max_vertex_edges = nv64 * (nv64 - 1) / 2;
    require(max_vertex_edges >= min_face_edges, "cannot build a manifold mesh");
    num_att_data = b.u8();
    uint32_t num_symbols = (uint32_t)b.varint();
    require(num_faces >= num_symbols, "fewer faces than symbols");
    uint32_t max_enc_faces = num_symbols + num_symbols / 3;
    require(num_faces <= max_enc_faces, "too many faces for the symbol count");
    uint32_t num_split_symbols = (uint32_t)b.varint();
    require(num_split_symbols <= num_symbols, "split symbols exceed symbols");
    m.att_data.assign(num_att_data, AttributeData());
    m.ct.c2v.assign((size_t)num_faces * 3, kInvalid);
    m.ct.opp.assign((size_t)num_faces * 3, kInvalid);
    m.ct.vcorner.clear();
    m.is_vert_hole.assign((size_t)num_encoded_vertices + num_split_symbols, 1);
    decode_split_events(num_faces);
    traversal_start(num_encoded_vertices + num_split_symbols);
    int num_conn_vertices = decode_connectivity_symbols((int)num_symbols);
    if (!m.att_data.empty()) decode_attribute_seams();
    for (auto &ad : m.att_data) {
      ad.conn.init(&m.ct);
      for (uint32_t c : ad.seam_corners) ad.conn.add_seam_edge(c);
      ad.conn.recompute_vertices();
    }
    m.pos_enc.init(m.ct.num_vertices());
    for (auto &ad : m.att_data) {
      size_t nv = std::max<size_t>(ad.conn.num_vertices(), m.ct.num_vertices());
      ad.enc.init(nv);
    }
    m.num_conn_vertices = (uint32_t)num_conn_vertices;
    assign_points_to_corners(num_conn_vertices);
  }
};

// ------------------------------------------------------- attribute decoding
struct AttributeSectionDecoder {
  Buffer &b;
  Mesh &m;
  AttributeSectionDecoder(Buffer &buf, Mesh &mesh) : b(buf), m(mesh) {}

  // MeshEdgeBreakerDecoder.cs:733-758 without the early return (D-11)
  EncodingData *encoding_data_for(int att_id) {
    for (auto &ad : m.att_data) {
      if (ad.decoder_id < 0 || ad.decoder_id >= (int)m.decoders.size()) continue;
      for (int a : m.decoders[ad.decoder_id].att_ids) if (a == att_id) return &ad.enc;
    }
    return &m.pos_enc;
  }
  // MeshEdgeBreakerDecoder.cs:710-731
  const AttrCornerTable *att_corner_table_for(int att_id) {
    for (auto &ad : m.att_data) {
      if (ad.decoder_id < 0 || ad.decoder_id >= (int)m.decoders.size()) continue;
      for (int a : m.decoders[ad.decoder_id].att_ids)
        if (a == att_id) return ad.is_connectivity_used ? &ad.conn : nullptr;
    }
    return nullptr;
  }

  // IO/ConnectivityDecoder.cs:16-44
  void decode() {
    int num_decoders = b.u8();
    m.decoders.assign(num_decoders, AttDecoder());
    if (!m.is_point_cloud && !m.is_sequential) {
      int pos_decoder = -1;
      for (int i = 0; i < num_decoders; ++i) {      // MeshEdgeBreakerDecoder.cs:640-708
        AttDecoder &d = m.decoders[i];
        d.att_data_id = b.i8();
        d.element_type = b.u8();
        if (d.att_data_id >= 0) {
          require(d.att_data_id < (int)m.att_data.size(), "unexpected attribute data");
          require(m.att_data[d.att_data_id].decoder_id < 0, "attribute data already mapped");
          m.att_data[d.att_data_id].decoder_id = i;
        } else {
          require(pos_decoder < 0, "position data already mapped");
          pos_decoder = i;
        }
        d.traversal_method = b.u8();
        require(d.traversal_method < 2, "invalid traversal method");
        if (d.element_type == 0) {
          if (d.att_data_id >= 0) m.att_data[d.att_data_id].is_connectivity_used = false;
        } else {
          require(d.traversal_method == 0, "unsupported traversal for corner attributes");
          require(d.att_data_id >= 0, "attribute data must be specified");
        }
      }
    }
    for (int i = 0; i < num_decoders; ++i) {        // AttributesDecoder.cs:19-63
      AttDecoder &d = m.decoders[i];
      uint32_t num_atts = (uint32_t)b.varint();
      require(num_atts <= 4096, "too many attributes");
      for (uint32_t k = 0; k < num_atts; ++k) {
        Attribute a;
        a.att_type = b.u8(); a.data_type = b.u8(); a.nc = b.u8(); a.normalized = b.u8() != 0;
        require(a.att_type < 5, "invalid attribute type");
        require(a.data_type != 0 && a.data_type < 12, "invalid data type");
        require(a.nc != 0, "zero components");
        a.unique_id = (uint32_t)b.varint();
        a.decoder_id = i;
        d.att_ids.push_back((int)m.atts.size());
        m.atts.push_back(a);
      }
      for (uint32_t k = 0; k < num_atts; ++k) {     // SequentialAttributeDecodersController.cs:16-27
        int t = b.u8();
        require(t <= 3, "unknown sequential decoder type");
        Attribute &a = m.atts[d.att_ids[k]];
        a.seq_type = t;
        if (t == 2) require(a.data_type == 9, "quantized attribute must be float32");
        if (t == 3) require(a.data_type == 9 && a.nc == 3, "normal attribute must be float32x3");
      }
    }
    for (int i = 0; i < num_decoders; ++i) decode_attributes(i);
  }

  // SequentialAttributeDecodersController.cs:29-38 + AttributesDecoder.cs:65-70
  void decode_attributes(int di) {
    AttDecoder &d = m.decoders[di];
    // sequence
    if (m.is_point_cloud || m.is_sequential) {      // LinearSequencer.cs:3-19
      d.point_ids.resize(m.num_points);
      for (uint32_t i = 0; i < m.num_points; ++i) d.point_ids[i] = i;
    } else {
      EncodingData *ed = d.att_data_id < 0 ? &m.pos_enc : &m.att_data[d.att_data_id].enc;
      if (d.element_type == 0 && d.traversal_method == 1) {   // MeshEdgeBreakerDecoder.cs:681-684
        PredictionDegree<CornerTable> t(m.ct, m.ct.num_vertices(), m.faces, *ed, d.point_ids);
        t.run();
      } else if (d.element_type == 0) {
        DepthFirst<CornerTable> t(m.ct, m.ct.num_vertices(), m.faces, *ed, d.point_ids);
        t.run();
      } else {
        AttrCornerTable &act = m.att_data[d.att_data_id].conn;
        DepthFirst<AttrCornerTable> t(act, (uint32_t)ed->vertex_to_data.size(), m.faces, *ed, d.point_ids);
        t.run();
      }
      // MeshTraversalSequencer.cs:33-50
      for (int aid : d.att_ids) {
        Attribute &a = m.atts[aid];
        a.point_map.assign(m.num_points, 0);
        for (uint32_t c = 0; c < m.ct.num_corners(); ++c) {
          uint32_t point = (uint32_t)m.faces[c];
          uint32_t v = d.element_type == 0 ? m.ct.vertex(c) : m.att_data[d.att_data_id].conn.vertex(c);
          require(v != kInvalid, "invalid vertex in point mapping");
          int32_t e = ed->vertex_to_data[v];
          require(point < m.num_points && e >= 0 && (uint32_t)e < m.num_points, "more attribute values than points");
          a.point_map[point] = (uint32_t)e;
        }
      }
    }
    uint32_t num_entries = (uint32_t)d.point_ids.size();
    for (int aid : d.att_ids) decode_portable(di, aid, num_entries);
    for (int aid : d.att_ids) decode_transform_data(aid);
    for (int aid : d.att_ids) to_original(aid, num_entries);
  }

  // SequentialAttributeDecoder.cs:47-52,75-86 (generic) /
  // SequentialIntegerAttributeDecoder.cs:23-44,53-101
  void decode_portable(int di, int aid, uint32_t num_entries) {
    AttDecoder &d = m.decoders[di];
    Attribute &a = m.atts[aid];
    a.num_entries = num_entries;
    if (a.seq_type == 0) {
      size_t stride = (size_t)data_type_length(a.data_type) * a.nc;
      const uint8_t *p = b.bytes(stride * num_entries);
      a.values.assign(p, p + stride * num_entries);
      return;
    }
    int nc = a.seq_type == 3 ? 2 : a.nc;   // normals are 2-component octahedral in portable form
    a.nc_portable = nc;
    int8_t method = b.i8();
    require(method >= -2 && method < 7, "invalid prediction scheme method");
    a.pred_method = method;
    Transform tr;
    bool have_scheme = false;
    if (method != -2) {
      int8_t tt = b.i8();
      require(tt >= -1 && tt < 4, "invalid prediction transform type");
      a.pred_transform = tt;
      // SequentialIntegerAttributeDecoder.cs:46-51 / SequentialNormalAttributeDecoder.cs:19-27 (D-6)
      if (a.seq_type == 3) have_scheme = (tt == 2 || tt == 3);
      else have_scheme = (tt == 1);
      tr.type = tt; tr.nc = nc;
    }
    size_t num_values = (size_t)num_entries * nc;
    std::vector<uint32_t> symbols(num_values, 0);
    uint8_t compressed = b.u8();
    if (compressed > 0) {
      decode_symbols(b, (uint32_t)num_values, nc, symbols);
    } else {                               // :68-84 with D-13 fixed
      uint8_t num_bytes = b.u8();
      require(num_bytes >= 1 && num_bytes <= 4, "invalid raw integer width");
      for (size_t i = 0; i < num_values; ++i) {
        const uint8_t *p = b.bytes(num_bytes);
        uint32_t v = 0;
        for (int k = 0; k < num_bytes; ++k) v |= (uint32_t)p[k] << (8 * k);
        symbols[i] = v;
      }
    }
    a.symbols = symbols;
    std::vector<int32_t> corr(num_values);
    // D-4: zig-zag iff the transform's corrections are not guaranteed positive
    if (num_values > 0 && (!have_scheme || !tr.corrections_positive()))
      for (size_t i = 0; i < num_values; ++i) corr[i] = zigzag_decode(symbols[i]);
    else
      for (size_t i = 0; i < num_values; ++i) corr[i] = (int32_t)symbols[i];
    if (!have_scheme) { a.portable = corr; return; }

    // scheme selection, PredictionSchemeDecoderFactory.cs:9-76
    int eff = method;
    const AttrCornerTable *act = nullptr;
    EncodingData *ed = nullptr;
    if (m.is_point_cloud || m.is_sequential) eff = 0;   // no corner table: PredictionSchemeDecoderFactory.cs:24-36,58-61
    else {
      ed = encoding_data_for(aid);
      act = att_corner_table_for(aid);
      // Which mesh schemes exist depends on the transform (the bitstream's factory): the wrap transform carries
      // the parallelogram family and the texture-coordinate schemes, the octahedral transforms carry only the
      // geometric normal scheme; every other combination is the delta scheme (D-26).
      if (tr.type == 1) {
        if (method == 1 || method == 2 || method == 4 || method == 5) eff = method;
        else if (method == 0 || method == 6) eff = 0;
        else throw Error(ERR_NOT_IMPLEMENTED, "prediction scheme not implemented in the oracle (deprecated texcoords)");
      } else eff = method == 6 ? 6 : 0;
    }
    // prediction data: scheme-specific first, then the transform's
    std::vector<uint8_t> orientations;
    if (eff == 5) {                        // MeshPredictionSchemeTexCoordsPortableDecoder.cs:66-85
      int32_t num_or = b.i32();
      require(num_or >= 0, "negative orientation count");
      bool last = true;
      RabsDecoder rd;
      rd.start(b);
      orientations.reserve(num_or);
      for (int i = 0; i < num_or; ++i) {
        if (rd.next() == 0) last = !last;
        orientations.push_back(last ? 1 : 0);
      }
    }
    std::vector<uint8_t> crease[4];
    if (eff == 4) {                        // MeshPredictionSchemeConstrainedMultiParallelogramDecoder.cs:110-134 (v2.2: no mode byte)
      for (int i = 0; i < 4; ++i) {
        uint64_t num_flags = b.varint();
        require(num_flags <= (uint64_t)m.ct.num_corners(), "more crease flags than corners");
        if (num_flags > 0) {
          RabsDecoder rd;
          rd.start(b);
          crease[i].resize((size_t)num_flags);
          for (uint64_t j = 0; j < num_flags; ++j) crease[i][j] = rd.next() ? 1 : 0;
        }
      }
    }
    tr.decode_data(b);
    std::vector<uint8_t> flips;
    if (eff == 6) {                        // MeshPredictionSchemeGeometricNormalDecoder.cs:71-82: transform data, then the flip bits
      RabsDecoder rd;
      rd.start(b);
      flips.resize(num_entries);
      for (uint32_t i = 0; i < num_entries; ++i) flips[i] = rd.next() ? 1 : 0;
    }
    if (num_values == 0) { a.portable.clear(); return; }
    if (eff == 6) {
      const Attribute *pos = nullptr;      // parent = portable position attribute, SequentialAttributeDecoder.cs:58-73
      for (auto &x : m.atts) if (x.att_type == 0) { pos = &x; break; }
      require(pos != nullptr && pos != &a && !pos->portable.empty(), "normal prediction without decoded positions");
      if (act) geometric_normal_original(tr, *act, *ed, corr, d.point_ids, *pos, flips, a.portable);
      else geometric_normal_original(tr, m.ct, *ed, corr, d.point_ids, *pos, flips, a.portable);
      return;
    }
    if (eff == 0) delta_original(tr, corr, nc, a.portable);
    else if (eff == 1) {
      if (act) parallelogram_original(tr, *act, *ed, corr, nc, a.portable);
      else parallelogram_original(tr, m.ct, *ed, corr, nc, a.portable);
    } else if (eff == 2) {
      if (act) multi_parallelogram_original(tr, *act, *ed, corr, nc, a.portable);
      else multi_parallelogram_original(tr, m.ct, *ed, corr, nc, a.portable);
    } else if (eff == 4) {
      if (act) constrained_multi_parallelogram_original(tr, *act, *ed, corr, nc, crease, a.portable);
      else constrained_multi_parallelogram_original(tr, m.ct, *ed, corr, nc, crease, a.portable);
    } else {
      // parent = portable position attribute, SequentialAttributeDecoder.cs:58-73
      const Attribute *pos = nullptr;
      for (auto &x : m.atts) if (x.att_type == 0) { pos = &x; break; }
      require(pos != nullptr && !pos->portable.empty(), "texcoord prediction without decoded positions");
      if (act) texcoords_portable_original(tr, *act, *ed, corr, nc, d.point_ids, *pos, orientations, a.portable);
      else texcoords_portable_original(tr, m.ct, *ed, corr, nc, d.point_ids, *pos, orientations, a.portable);
    }
  }

  // SequentialQuantizationAttributeDecoder.cs:26-33 + AttributeQuantizationTransform.cs:110-121;
  // SequentialNormalAttributeDecoder.cs:38-45 (D-5) + AttributeOctahedronTransform.cs:39-42
  void decode_transform_data(int aid) {
    Attribute &a = m.atts[aid];
    if (a.seq_type == 2) {
      a.q_min.resize(a.nc);
      for (int c = 0; c < a.nc; ++c) a.q_min[c] = b.f32();
      a.q_range = b.f32();
      a.q_bits = b.u8();
      require(a.q_bits >= 1 && a.q_bits <= 30, "invalid quantization bits");
    } else if (a.seq_type == 3) {
      a.oct_bits = b.u8();
      require(a.oct_bits >= 2 && a.oct_bits <= 30, "invalid octahedron bits");
    }
  }

  // SequentialIntegerAttributeDecoder.cs:103-160, AttributeQuantizationTransform.cs:179-199 +
  // Core/Dequantizer.cs:15-23, AttributeOctahedronTransform.cs:82-102 (D-7)
  void to_original(int aid, uint32_t num_entries) {
    Attribute &a = m.atts[aid];
    if (a.seq_type == 0) return;
    if (a.seq_type == 1) {
      int w = data_type_length(a.data_type);
      require(w == 1 || w == 2 || w == 4, "unsupported integer attribute type");
      a.values.assign((size_t)num_entries * a.nc * w, 0);
      for (size_t i = 0; i < (size_t)num_entries * a.nc; ++i) {
        int32_t v = a.portable[i];
        memcpy(&a.values[i * w], &v, w);   // little-endian narrowing
      }
    } else if (a.seq_type == 2) {
      uint32_t max_q = (1u << a.q_bits) - 1;
      volatile float delta = a.q_range / (float)(int32_t)max_q;
      a.values.assign((size_t)num_entries * a.nc * 4, 0);
      for (uint32_t e = 0; e < num_entries; ++e)
        for (int c = 0; c < a.nc; ++c) {
          volatile float prod = (float)a.portable[(size_t)e * a.nc + c] * delta;  // rounded to f32
          float v = prod + a.q_min[c];                                            // second rounding
          memcpy(&a.values[((size_t)e * a.nc + c) * 4], &v, 4);
        }
    } else {
      OctaToolBox tb;
      tb.set_bits(a.oct_bits);
      a.values.assign((size_t)num_entries * 12, 0);
      for (uint32_t e = 0; e < num_entries; ++e) {
        float v[3];
        tb.to_unit_vector(a.portable[(size_t)e * 2], a.portable[(size_t)e * 2 + 1], v);
        memcpy(&a.values[(size_t)e * 12], v, 12);
      }
    }
  }
};

// Metadata is outside the hot path; skipped structurally.
// IO/Metadata/MetadataDecoder.cs:5-49
static void skip_metadata_element(Buffer &b, int depth) {
  require(depth < 16, "metadata nesting too deep");   // the device parse keeps an explicit stack of 16 levels
  uint32_t n = (uint32_t)b.varint();
  for (uint32_t i = 0; i < n; ++i) {
    uint8_t ks = b.u8(); b.bytes(ks);
    uint64_t vs = b.varint(); b.bytes((size_t)vs);
  }
  uint32_t ns = (uint32_t)b.varint();
  for (uint32_t i = 0; i < ns; ++i) { uint8_t ks = b.u8(); b.bytes(ks); skip_metadata_element(b, depth + 1); }
}

// IO/DracoDecoder.cs:19-99
static void decode(const uint8_t *data, size_t len, Mesh &m) {
  Buffer b(data, len);
  require(len >= 11, "stream too short");
  require(memcmp(b.bytes(5), "DRACO", 5) == 0, "invalid Draco file");
  m.major = b.u8(); m.minor = b.u8(); m.encoder_type = b.u8(); m.encoder_method = b.u8(); m.flags = b.u16();
  require(m.major == 2 && m.minor == 2, "only bitstream 2.2 is supported");
  if (m.flags & 0x8000) {
    uint32_t n = (uint32_t)b.varint();
    for (uint32_t i = 0; i < n; ++i) { (void)b.varint(); skip_metadata_element(b, 0); }
    skip_metadata_element(b, 0);
  }
  if (m.encoder_type == 0) {
    // Point cloud, sequential: the reference stops at DracoDecoder.cs:70; layout
    // follows the upstream format (int32 num_points, LinearSequencer).
    require(m.encoder_method == 0, "only sequential point clouds are supported");
    m.is_point_cloud = true;
    int32_t np = b.i32();
    require(np >= 0, "negative point count");
    m.num_points = (uint32_t)np;
    AttributeSectionDecoder ad(b, m);
    ad.decode();
  } else if (m.encoder_type == 1) {
    if (m.encoder_method == 0) {
      // Mesh/MeshSequentialDecoder.cs:8-123.  D-22: the C# tests the sign bit inverted (:100) relative to its own
      // encoder (MeshSequentialEncoder.cs:79) and to the bitstream: LSB set = negative difference.
      m.is_sequential = true;
      uint32_t nf = (uint32_t)b.varint(), np = (uint32_t)b.varint();
      require(nf <= 0x7FFFFFFFu / 3, "too many faces");
      uint8_t method = b.u8();
      m.faces.assign((size_t)nf * 3, 0);
      if (method == 0) {
        std::vector<uint32_t> sym;
        decode_symbols(b, nf * 3, 1, sym);
        int32_t last = 0;
        for (size_t k = 0; k < (size_t)nf * 3; ++k) {
          uint32_t e = sym[k];
          int32_t diff = (int32_t)(e >> 1);
          if (e & 1) { require(diff <= last, "negative point index"); diff = -diff; }
          else require(diff <= 0x7FFFFFFF - last, "point index overflow");
          last += diff;
          m.faces[k] = last;
        }
      } else if (method == 1) {
        for (size_t k = 0; k < (size_t)nf * 3; ++k) {
          if (np < 256) m.faces[k] = b.u8();
          else if (np < (1u << 16)) m.faces[k] = b.u16();
          else if (np < (1u << 21)) m.faces[k] = (int32_t)b.varint();
          else m.faces[k] = (int32_t)b.u32();
        }
      } else require(false, "unsupported sequential connectivity method");
      m.num_points = np;
      AttributeSectionDecoder ad(b, m);
      ad.decode();
      m.end_pos = b.pos;
      return;
    }
    require(m.encoder_method == 1, "unsupported encoder method");
    m.traversal_type = b.u8();
    require(m.traversal_type <= 2, "unsupported Edgebreaker traversal type");
    EdgebreakerDecoder eb(b, m, m.traversal_type);
    eb.decode_connectivity();
    AttributeSectionDecoder ad(b, m);
    ad.decode();
  } else {
    require(false, "unsupported encoder type");
  }
  m.end_pos = b.pos;
}

}  // namespace orc

// ------------------------------------------------------------------- C ABI
extern "C" {

struct orc_mesh { orc::Mesh m; };

struct orc_attr_info {
  int32_t att_type, data_type, num_components, normalized;
  uint32_t unique_id;
  int32_t seq_type, decoder_id, pred_method, pred_transform;
  uint32_t num_entries;
  int32_t nc_portable, q_bits;
  float q_range;
  float q_min[4];
  int32_t oct_bits;
  uint64_t value_bytes;
};

orc_mesh *orc_decode(const uint8_t *data, size_t len, int *err_code, char *err, size_t errlen) {
  orc_mesh *h = new orc_mesh();
  try {
    orc::decode(data, len, h->m);
    if (err_code) *err_code = 0;
    return h;
  } catch (const orc::Error &e) {
    if (err_code) *err_code = e.code;
    if (err && errlen) snprintf(err, errlen, "%s", e.what());
  } catch (const std::exception &e) {
    if (err_code) *err_code = orc::ERR_INVALID_DATA;
    if (err && errlen) snprintf(err, errlen, "%s", e.what());
  }
  delete h;
  return nullptr;
}
void orc_free(orc_mesh *h) { delete h; }

void orc_header(const orc_mesh *h, int32_t out[8]) {
  const orc::Mesh &m = h->m;
  out[0] = m.major; out[1] = m.minor; out[2] = m.encoder_type; out[3] = m.encoder_method;
  out[4] = m.flags; out[5] = m.traversal_type; out[6] = (int32_t)m.att_data.size(); out[7] = (int32_t)m.end_pos;
}
uint32_t orc_num_faces(const orc_mesh *h) { return h->m.is_point_cloud ? 0 : (h->m.is_sequential ? (uint32_t)(h->m.faces.size() / 3) : h->m.ct.num_faces()); }
uint32_t orc_num_points(const orc_mesh *h) { return h->m.num_points; }
uint32_t orc_num_vertices(const orc_mesh *h) { return h->m.ct.num_vertices(); }
uint32_t orc_num_attributes(const orc_mesh *h) { return (uint32_t)h->m.atts.size(); }
uint32_t orc_num_decoders(const orc_mesh *h) { return (uint32_t)h->m.decoders.size(); }
void orc_faces(const orc_mesh *h, int32_t *out) { memcpy(out, h->m.faces.data(), h->m.faces.size() * 4); }
void orc_corner_table(const orc_mesh *h, uint32_t *opp, uint32_t *c2v, uint32_t *vcorner) {
  const orc::CornerTable &ct = h->m.ct;   // empty for sequential meshes (the caller's arrays stay as they are)
  if (opp) memcpy(opp, ct.opp.data(), ct.opp.size() * 4);
  if (c2v) memcpy(c2v, ct.c2v.data(), ct.c2v.size() * 4);
  if (vcorner) memcpy(vcorner, ct.vcorner.data(), ct.vcorner.size() * 4);
}
uint32_t orc_eb_symbols(const orc_mesh *h, uint8_t *out) {
  if (out) memcpy(out, h->m.eb_symbols.data(), h->m.eb_symbols.size());
  return (uint32_t)h->m.eb_symbols.size();
}
// decoder-level: entry->point sequence and encoding data
uint32_t orc_decoder_num_entries(const orc_mesh *h, uint32_t d) { return (uint32_t)h->m.decoders[d].point_ids.size(); }
void orc_decoder_info(const orc_mesh *h, uint32_t d, int32_t out[4]) {
  const orc::AttDecoder &x = h->m.decoders[d];
  out[0] = x.att_data_id; out[1] = x.element_type; out[2] = x.traversal_method; out[3] = (int32_t)x.att_ids.size();
}
void orc_decoder_sequence(const orc_mesh *h, uint32_t d, uint32_t *point_ids, uint32_t *data_to_corner) {
  const orc::Mesh &m = h->m;
  const orc::AttDecoder &x = m.decoders[d];
  if (point_ids) memcpy(point_ids, x.point_ids.data(), x.point_ids.size() * 4);
  if (data_to_corner && !m.is_point_cloud && !m.is_sequential) {
    const orc::EncodingData &ed = x.att_data_id < 0 ? m.pos_enc : m.att_data[x.att_data_id].enc;
    memcpy(data_to_corner, ed.data_to_corner.data(), ed.data_to_corner.size() * 4);
  }
}
void orc_attr_get_info(const orc_mesh *h, uint32_t a, orc_attr_info *o) {
  const orc::Attribute &x = h->m.atts[a];
  memset(o, 0, sizeof(*o));
  o->att_type = x.att_type; o->data_type = x.data_type; o->num_components = x.nc; o->normalized = x.normalized;
  o->unique_id = x.unique_id; o->seq_type = x.seq_type; o->decoder_id = x.decoder_id;
  o->pred_method = x.pred_method; o->pred_transform = x.pred_transform; o->num_entries = x.num_entries;
  o->nc_portable = x.nc_portable; o->q_bits = x.q_bits; o->q_range = x.q_range;
  for (size_t c = 0; c < x.q_min.size() && c < 4; ++c) o->q_min[c] = x.q_min[c];
  o->oct_bits = x.oct_bits; o->value_bytes = x.values.size();
}
void orc_attr_values(const orc_mesh *h, uint32_t a, void *out) { memcpy(out, h->m.atts[a].values.data(), h->m.atts[a].values.size()); }
void orc_attr_portable(const orc_mesh *h, uint32_t a, int32_t *out) { memcpy(out, h->m.atts[a].portable.data(), h->m.atts[a].portable.size() * 4); }
void orc_attr_symbols(const orc_mesh *h, uint32_t a, uint32_t *out) { memcpy(out, h->m.atts[a].symbols.data(), h->m.atts[a].symbols.size() * 4); }
uint32_t orc_attr_point_map(const orc_mesh *h, uint32_t a, uint32_t *out) {
  const auto &pm = h->m.atts[a].point_map;
  if (out) memcpy(out, pm.data(), pm.size() * 4);
  return (uint32_t)pm.size();
}

// ---- primitives, for known-answer tests ------------------------------------
uint64_t orc_varint(const uint8_t *d, size_t n, size_t *consumed) {
  orc::Buffer b(d, n);
  try { uint64_t v = b.varint(); *consumed = b.pos; return v; } catch (...) { *consumed = 0; return 0; }
}
uint32_t orc_bits(const uint8_t *d, size_t n, const int32_t *counts, int num, uint32_t *out) {
  orc::Buffer b(d, n);
  uint64_t dummy;
  try {
    b.start_bits(false, &dummy);
    for (int i = 0; i < num; ++i) out[i] = b.bits(counts[i]);
    b.end_bits();
    return (uint32_t)b.pos;
  } catch (...) { return 0xFFFFFFFFu; }
}
uint64_t orc_int_sqrt(uint64_t v) { return orc::int_sqrt(v); }
int32_t orc_zigzag(uint32_t s) { return orc::zigzag_decode(s); }
// Decode a DecodeSymbols() block (scheme byte first).  Returns bytes consumed or -1.
int64_t orc_decode_symbols(const uint8_t *d, size_t n, uint32_t num_values, int nc, uint32_t *out) {
  orc::Buffer b(d, n);
  std::vector<uint32_t> v;
  try { orc::decode_symbols(b, num_values, nc, v); } catch (...) { return -1; }
  memcpy(out, v.data(), v.size() * 4);
  return (int64_t)b.pos;
}
// Decode an rABS block {prob_zero, size varint, bytes} into num_bits bits.
int64_t orc_decode_rabs(const uint8_t *d, size_t n, uint32_t num_bits, uint8_t *out) {
  orc::Buffer b(d, n);
  try {
    orc::RabsDecoder r; r.start(b);
    for (uint32_t i = 0; i < num_bits; ++i) out[i] = (uint8_t)r.next();
  } catch (...) { return -1; }
  return (int64_t)b.pos;
}
void orc_oct_to_unit(int bits, int s, int t, float out[3]) {
  orc::OctaToolBox tb; tb.set_bits(bits); tb.to_unit_vector(s, t, out);
}
float orc_dequantize(int32_t q, float range, int bits, float minv) {
  uint32_t max_q = (1u << bits) - 1;
  volatile float delta = range / (float)(int32_t)max_q;
  volatile float prod = (float)q * delta;
  return prod + minv;
}

}  // extern "C"
