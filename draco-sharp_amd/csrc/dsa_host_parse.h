// draco-sharp_amd/csrc/dsa_host_parse.h
// Host side of batch construction that needs no HIP: the sizing parse of one stream (fixed header, section
// lengths, attribute descriptors) and the placement of one mesh's regions in the batch arena.  Included by
// dsa_api.hip; tests/hostcheck includes it too, to lay out a one-mesh arena for the sanitizer build of the
// general path.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "dsa_types.h"

namespace {

// What the host learns from the fixed part of a stream; used only to size the arena.
struct HostAttr { uint8_t att_type, data_type, nc, seq_type; bool corner = false; uint8_t dec = 0; };
struct HostMesh {
  int status = 0;   // failure of the sizing parse (the device parse decides the reported status)
  uint32_t faces = 0, enc_vertices = 0, split_symbols = 0, splits = 0, num_att_data = 0;
  bool general = false;   // valence traversal or corner attributes: decoded by k_general
  bool valence = false;   // valence-coded connectivity on the fast kernels (k_valence_lists in front of the connectivity waves)
  bool seamed = false;    // corner-attribute decoders (attribute seams) on the fast kernels (k_seam_tables, k_traverse_att)
  uint32_t meta_off = 0, meta_len = 0;   // metadata block of the stream (flag 0x8000), for dsa_batch_copy_metadata
  int first_method = -128;               // prediction method byte of the first attribute (the values of the first decoder start with it), -128: none
  std::vector<HostAttr> atts;
};

struct HRd {
  const uint8_t *p; size_t n, pos = 0; bool ok = true;
  HRd(const uint8_t *d, size_t len) : p(d), n(len) {}
  uint32_t u8() { if (pos < n) return p[pos++]; ok = false; return 0; }
  uint64_t varint() {
    uint64_t r = 0;
    for (int shift = 0; shift < 64; shift += 7) { uint32_t b = u8(); r |= (uint64_t)(b & 0x7F) << shift; if (!(b & 0x80)) return r; }
    ok = false; return r;
  }
  void skip(uint64_t k) { if (!ok || k > n - pos) { ok = false; pos = n; } else pos += (size_t)k; }
};

static void skip_metadata_element(HRd &r, int depth) {
  if (depth > 15) { r.ok = false; return; }             // same nesting limit as k_locate's explicit stack
  uint32_t ne = (uint32_t)r.varint();
  for (uint32_t i = 0; i < ne && r.ok; ++i) { uint32_t ks = r.u8(); r.skip(ks); uint64_t vs = r.varint(); r.skip(vs); }
  uint32_t ns = (uint32_t)r.varint();
  for (uint32_t i = 0; i < ns && r.ok; ++i) { uint32_t ks = r.u8(); r.skip(ks); skip_metadata_element(r, depth + 1); }
}


// Skips one SymbolDecoding.DecodeSymbols block (Entropy/SymbolDecoding.cs:7-67).  The bit section of the tagged
// scheme has no length field, so its tags are decoded here (sizing only; the device decodes everything again).
static void host_skip_symbols(HRd &r, uint64_t num_values, uint32_t nc) {
  if (num_values == 0 || !r.ok) return;
  const uint32_t scheme = r.u8();
  auto read_table = [&](std::vector<uint32_t> &prob) {
    const uint64_t ns = r.varint();
    if (!r.ok || ns > (1u << 20)) { r.ok = false; return; }
    prob.assign((size_t)ns, 0);
    for (uint64_t i = 0; i < ns && r.ok; ++i) {
      const uint32_t pd = r.u8(), token = pd & 3;
      if (token == 3) { const uint64_t off = pd >> 2; if (i + off >= ns) { r.ok = false; return; } i += off; }
      else { uint32_t p = pd >> 2; for (uint32_t k = 0; k < token; ++k) p |= r.u8() << (8 * (k + 1) - 2); prob[(size_t)i] = p; }
    }
  };
  std::vector<uint32_t> prob;
  if (scheme == 1) {
    const uint32_t mbl = r.u8();
    if (mbl < 1 || mbl > 18) { r.ok = false; return; }
    read_table(prob);
    const uint64_t size = r.varint();
    r.skip(size);
  } else if (scheme == 0) {
    read_table(prob);
    if (!r.ok || prob.empty()) { r.ok = false; return; }
    const uint64_t size = r.varint();
    if (!r.ok || size < 1 || size > r.n - r.pos) { r.ok = false; return; }
    const uint8_t *buf = r.p + r.pos;
    r.skip(size);
    // rANS, precision 12 (RAnsSymbolCoding.cs:10-27 for 5-bit tags)
    std::vector<uint32_t> cum(prob.size() + 1, 0), lut(4096, 0);
    for (size_t i = 0; i < prob.size(); ++i) {
      cum[i + 1] = cum[i] + prob[i];
      if (cum[i + 1] > 4096) { r.ok = false; return; }
      for (uint32_t j = cum[i]; j < cum[i + 1]; ++j) lut[j] = (uint32_t)i;
    }
    if (cum.back() != 4096) { r.ok = false; return; }
    size_t off = (size_t)size;
    uint32_t x = buf[off - 1] >> 6, state;
    if (x == 0) { off -= 1; state = buf[off] & 0x3F; }
    else if (x == 1) { if (off < 2) { r.ok = false; return; } off -= 2; state = (buf[off] | (buf[off + 1] << 8)) & 0x3FFF; }
    else if (x == 2) { if (off < 3) { r.ok = false; return; } off -= 3; state = (buf[off] | (buf[off + 1] << 8) | (buf[off + 2] << 16)) & 0x3FFFFF; }
    else { if (off < 4) { r.ok = false; return; } off -= 4; state = (buf[off] | (buf[off + 1] << 8) | (buf[off + 2] << 16) | ((uint32_t)buf[off + 3] << 24)) & 0x3FFFFFFF; }
    state += 16384;
    uint64_t bits = 0;
    for (uint64_t i = 0; i < num_values; i += nc) {
      while (state < 16384 && off > 0) state = state * 256 + buf[--off];
      const uint32_t rem = state & 4095u, sy = lut[rem];
      state = (state >> 12) * prob[sy] + rem - cum[sy];
      if (sy > 32) { r.ok = false; return; }
      bits += (uint64_t)sy * nc;
    }
    r.skip((bits + 7) >> 3);
  } else r.ok = false;
}

static uint32_t dt_len(uint32_t dt) {
  switch (dt) { case 1: case 2: case 11: return 1; case 3: case 4: return 2; case 5: case 6: case 9: return 4; case 7: case 8: case 10: return 8; default: return 0; }
}

// Mirrors the head of k_locate (same checks, same order) up to the attribute descriptors.
static void host_parse(const uint8_t *s, size_t len, HostMesh &m, bool want_general = false) {
  HRd r(s, len);
  auto bad = [&](int code) { m.status = code; };
  if (len < 11 || memcmp(s, "DRACO", 5) != 0) return bad(ST_INVALID);
  r.pos = 5;
  uint32_t major = r.u8(), minor = r.u8(), type = r.u8(), method = r.u8();
  uint32_t flags = r.u8(); flags |= r.u8() << 8;
  if (major != 2 || minor != 2) return bad(ST_INVALID);
  if (flags & 0x8000) {
    const size_t begin = r.pos;
    uint32_t natt = (uint32_t)r.varint();
    for (uint32_t i = 0; i < natt && r.ok; ++i) { (void)r.varint(); skip_metadata_element(r, 0); }
    skip_metadata_element(r, 0);
    if (!r.ok) return bad(ST_INVALID);
    m.meta_off = (uint32_t)begin; m.meta_len = (uint32_t)(r.pos - begin);
  }
  if (type > 1) return bad(ST_INVALID);
  const bool point_cloud = type == 0;
  const bool linear = point_cloud || method == 0;          // no decoder triples: linear sequencing
  if (point_cloud) {
    if (method > 1) return bad(ST_INVALID);
    if (method != 0) return bad(ST_NOTIMPL);                  // kd-tree point clouds
    uint32_t np = r.u8(); np |= r.u8() << 8; np |= r.u8() << 16; np |= r.u8() << 24;
    // element counts far beyond what the stream's bytes can carry are rejected here, so that one corrupt header
    // cannot claim the arena of the whole batch (a real stream spends at least a fraction of a bit per element)
    if (!r.ok || np > 0x7FFFFFFFu || np > 1024ull * len) return bad(ST_INVALID);
    m.faces = 0; m.enc_vertices = np; m.split_symbols = 0; m.splits = 0; m.num_att_data = 0;
  } else if (method == 0) {
    // sequential mesh (Mesh/MeshSequentialDecoder.cs:8-123): decoded by the general path
    const uint64_t nf = r.varint(), np = r.varint();
    if (!r.ok || nf > 0x7FFFFFFFu / 3 || np > 0x7FFFFFFFu || nf > 1024ull * len || np > 1024ull * len) return bad(ST_INVALID);
    const uint32_t cm = r.u8();
    if (cm == 0) host_skip_symbols(r, 3 * nf, 1);
    else if (cm == 1) {
      if (np < 256) r.skip(3 * nf);
      else if (np < (1u << 16)) r.skip(6 * nf);
      else if (np < (1u << 21)) { for (uint64_t k = 0; k < 3 * nf && r.ok; ++k) (void)r.varint(); }
      else r.skip(12 * nf);
    } else return bad(ST_INVALID);
    if (!r.ok) return bad(ST_INVALID);
    m.faces = (uint32_t)nf; m.enc_vertices = (uint32_t)np; m.split_symbols = 0; m.splits = 0; m.num_att_data = 0;
    m.general = true;
  } else {
    if (method > 1) return bad(ST_INVALID);
    uint32_t traversal = r.u8();
    if (!r.ok || traversal > 2) return bad(ST_INVALID);
    if (traversal == 1) m.general = true;                       // predictive symbols (deprecated by the format): general path
    // valence symbols (what stock encoders write for larger meshes): fast kernels while the ids fit their 16-byte face records
    uint64_t nv = r.varint(), nf = r.varint();
    if (!r.ok || nf > 0x7FFFFFFFu / 3 || nv > nf * 3 || nf > 1024ull * len) return bad(ST_INVALID);
    uint32_t nad = r.u8();
    uint64_t nsym = r.varint();
    if (!r.ok || nf < nsym || nf > nsym + nsym / 3) return bad(ST_INVALID);
    uint64_t nss = r.varint();
    if (!r.ok || nss > nsym) return bad(ST_INVALID);
    if (nad > DSA_MAX_ATT_DATA) return bad(ST_NOTIMPL);
    uint64_t nsplits = r.varint();
    if (!r.ok || nsplits > nf) return bad(ST_INVALID);
    m.faces = (uint32_t)nf; m.enc_vertices = (uint32_t)nv; m.split_symbols = (uint32_t)nss; m.splits = (uint32_t)nsplits; m.num_att_data = nad;
    if (traversal == 2 && (4 * nf > (1u << 20) || nv + nss >= (1u << 20))) m.general = true;
    m.valence = traversal == 2 && !m.general;
    for (uint64_t i = 0; i < 2 * nsplits && r.ok; ++i) (void)r.varint();
    r.skip((nsplits + 7) >> 3);
    uint64_t sz;
    if (traversal != 2) { sz = r.varint(); r.skip(sz); }        // symbols
    (void)r.u8(); sz = r.varint(); r.skip(sz);                  // start faces
    for (uint32_t i = 0; i < nad; ++i) { (void)r.u8(); sz = r.varint(); r.skip(sz); }
    if (traversal == 1) { r.skip(4); (void)r.u8(); sz = r.varint(); r.skip(sz); }   // MeshEdgeBreakerTraversalPredictiveDecoder.cs:19-27
    if (traversal == 2) {                                       // MeshEdgeBreakerTraversalValenceDecoder.cs:22-69
      for (int c = 0; c < 6 && r.ok; ++c) {
        const uint64_t num = r.varint();
        if (!r.ok || num > nf) return bad(ST_INVALID);
        host_skip_symbols(r, num, 1);
      }
    }
  }
  uint32_t ndec = r.u8();
  if (!r.ok) return bad(ST_INVALID);
  if (ndec > DSA_MAX_ATT) return bad(ST_NOTIMPL);               // a valid stream, more attribute decoders than the device path carries
  bool corner_dec[DSA_MAX_ATT + 1] = {};
  bool any_corner = false;
  if (!linear) for (uint32_t i = 0; i < ndec; ++i) {
    (void)r.u8();
    corner_dec[i] = r.u8() != 0;                                // MeshAttributeElementType: corner attribute
    const bool prediction_degree = r.u8() != 0;                 // MeshTraversalMethod
    any_corner = any_corner || corner_dec[i];
    if (prediction_degree) m.general = true;
  }
  // corner attributes (attribute seams): the fast kernels while the seam masks fit a byte per corner (seven attribute data and a mark)
  if (any_corner && m.num_att_data > 7) m.general = true;
  const bool force_general = getenv("DSA_FORCE_GENERAL") != nullptr;   // tests: every Edgebreaker mesh through k_general
  if ((force_general || want_general) && !point_cloud) m.general = true;
  m.seamed = any_corner && !m.general;
  for (uint32_t i = 0; i < ndec; ++i) {
    uint64_t k = r.varint();
    if (!r.ok) return bad(ST_INVALID);
    if (m.atts.size() + k > DSA_MAX_ATT) return bad(ST_NOTIMPL);
    size_t first = m.atts.size();
    for (uint64_t j = 0; j < k; ++j) {
      HostAttr a;
      a.att_type = (uint8_t)r.u8(); a.data_type = (uint8_t)r.u8(); a.nc = (uint8_t)r.u8(); (void)r.u8();
      (void)r.varint();
      a.seq_type = 0;
      a.corner = corner_dec[i];
      a.dec = (uint8_t)i;
      m.atts.push_back(a);
    }
    for (uint64_t j = 0; j < k; ++j) m.atts[first + j].seq_type = (uint8_t)r.u8();
  }
  if (!r.ok) return bad(ST_INVALID);
  for (auto &a : m.atts) if (a.nc == 0 || dt_len(a.data_type) == 0 || a.seq_type > 3) return bad(ST_INVALID);
  // SequentialIntegerAttributeDecoder.cs:70-76: the values of an integer / quantised attribute start with its prediction method
  if (!linear && !m.atts.empty() && m.atts[0].seq_type != 0 && r.pos < r.n) m.first_method = (int8_t)r.p[r.pos];
}

static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Places the regions of one mesh behind `cur` (arena offset) and returns the new end.  `slack` bytes are left
// after every region (the kernels over-read whole 16-byte words; the host check passes a larger red zone).
// `out_cur` (optional): the arrays a caller receives -- faces, attribute values, point maps -- are placed behind the cursors of
// *out_cur instead, one cursor per kind: the batch's output block (which dsa_api.hip puts behind all scratch so that one transfer
// downloads it) is three sub-blocks, every mesh's faces, then every mesh's values, then every mesh's maps -- the compact download
// takes the values as they are and packs the other two.  Offsets relative to the respective sub-block.
// MpPrep records of E entries, then the crease flags: at most 4 E bits in four arrays of whole words
static inline uint64_t mp_crease_words(uint64_t E) { return (4 * E + 31) / 32 + 4; }
static inline uint64_t mp_region_bytes(uint64_t E) { return sizeof(MpPrep) * E + 4 * mp_crease_words(E); }
struct OutCursors { uint64_t faces = 0, values = 0, maps = 0; };
static inline uint64_t layout_mesh(const HostMesh &h, uint64_t stream_len, MeshLayout &L, uint64_t cur, uint64_t slack,
                                   std::vector<std::pair<uint64_t, uint64_t>> *regions = nullptr, OutCursors *out_cur = nullptr) {
  const uint64_t F = h.faces, V = (uint64_t)h.enc_vertices + h.split_symbols;
  L.cap_faces = (uint32_t)F;
  L.cap_vertices = (uint32_t)V;
  L.cap_attributes = (uint32_t)h.atts.size();
  L.cap_splits = h.splits;
  auto take = [&](uint64_t bytes) { uint64_t at = cur; cur = align_up(cur + bytes + slack, 256); if (regions) regions->push_back({at, bytes}); return at; };
  auto take_out = [&](uint64_t bytes, uint64_t OutCursors::*which) { if (!out_cur) return take(bytes); uint64_t at = out_cur->*which; out_cur->*which = align_up(at + bytes + slack, 256); return at; };
  // the fast kernels' face records: 16 bytes while every id fits 20 bits (the general path keeps plain arrays here: 24 F)
  L.rec_compact = (!h.general && 4 * F <= (1u << 20) && V < (1u << 20)) ? 1u : 0u;
  L.frec = take(L.rec_compact ? 16 * F : 32 * F);
  L.vrec = take(8 * V);
  L.d2c = take(4 * V); L.v2d = take(4 * V);
  L.fvis = take(F); L.vvis = take(V);
  L.fstamp = take(4 * F); L.vstamp = take(4 * V);
  L.splits = take(16ull * h.splits);
  L.vrank = take(4 * V); L.para = take(12 * V);
  L.faces = take_out(12 * F, &OutCursors::faces);
  // corner attributes carry up to 3F entries, and seams up to 3F points
  const uint64_t P = ((h.general || h.seamed) && h.num_att_data > 0) ? std::max<uint64_t>(3 * F, V) : V;
  L.cap_points = (uint32_t)P;
  if (h.seamed) {
    const SeamLayout g = seam_layout(F, V, h.num_att_data, L.rec_compact != 0);
    L.seam = take(g.total);
    L.seam_bytes = g.total;
  }
  if (h.general) {
    const GenLayout g = gen_layout(F, V, h.splits, h.num_att_data, stream_len);
    L.gen = take(g.total);
    L.gen_bytes = g.total;
  }
  for (size_t a = 0; a < h.atts.size(); ++a) {
    const HostAttr &A = h.atts[a];
    uint64_t ncp = A.seq_type == 3 ? 2 : A.nc;
    const uint64_t E = A.corner ? std::max<uint64_t>(3 * F, V) : V;   // entry capacity
    uint64_t wcap = E * ncp, ocap = E * A.nc * dt_len(A.data_type);
    if (ocap < V) ocap = V;                  // tag bytes of the tagged scheme are staged here
    L.work[a] = take(4 * wcap); L.work_cap[a] = (uint32_t)wcap;
    L.out[a] = take_out(ocap, &OutCursors::values); L.out_cap[a] = (uint32_t)(ocap > 0xFFFFFFFFu ? 0xFFFFFFFFu : ocap);
    L.map[a] = take_out(4 * P, &OutCursors::maps);
    // ConstrainedMultiParallelogram records: the scheme shows at the head of the first attribute's values (positions), and an encoder
    // that gives it to them gives it to every attribute the scheme applies to (integer / quantised, on the position connectivity,
    // at most four components) -- those get the region too, whatever they turn out to use
    const bool mp = h.faces != 0 && !h.general && h.first_method == 4 && !A.corner && ncp <= 4 && (A.seq_type == 1 || A.seq_type == 2);
    uint64_t prep = (h.faces != 0 && A.nc == 2 && (A.seq_type == 1 || A.seq_type == 2)) ? sizeof(TcPrep) * E : 0;
    if (mp) prep = std::max<uint64_t>(prep, mp_region_bytes(E));
    L.tc[a] = prep ? take(prep) : 0;
    if (mp) L.mp_att |= 1u << a;
  }
  return cur;
}

}  // namespace
