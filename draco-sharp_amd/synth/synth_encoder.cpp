// draco-sharp_amd/synth/synth_encoder.cpp
// -----------------------------------------------------------------------------
// Synthetic-input generator: procedural meshes + the CPU writer of Draco v2.2
// Edgebreaker streams of ../csrc/dsa_encode_host.h.  No Draco encoder exists in the build image and the
// reference's own encode half cannot run, so
// tests and bench.py make their .drc inputs with this tool.  It is NOT part of
// the decode product path and NOT the oracle; it only has to emit streams a
// conformant decoder accepts, following the rules of SURVEY.md Appendix D:
//   IO/Mesh/MeshEdgeBreakerEncoder.cs:38-124,176-303  (connectivity traversal)
//   IO/Mesh/MeshEdgeBreakerTraversalEncoder.cs:40-71  (symbol/start-face/seam sections)
//   IO/Entropy/SymbolEncoding.cs:8-193, RAnsSymbolEncoder.cs:15-184,
//   RAnsEncoder.cs:22-30, AnsEncoder.cs:19-64, BitCoders/RAnsBitEncoder.cs
//   IO/Attributes/SequentialIntegerAttributeEncoder.cs:55-128,
//   AttributeQuantizationTransform.cs:66-177, OctahedronToolBox.cs:28-119,
//   PredictionSchemes/*Encoder.cs, *EncodingTransform.cs
// (each with the defects listed there corrected to the bitstream's semantics).
// -----------------------------------------------------------------------------
#include "../csrc/dsa_encode_host.h"

namespace synth {

// ---------------------------------------------------------------- generators
struct Rng {   // splitmix64
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double normal() { double u1 = uniform(), u2 = uniform(); if (u1 < 1e-300) u1 = 1e-300; return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
};

struct MeshBuf { std::vector<float> pos, nrm, uv; std::vector<uint32_t> faces; };

// kind 0: (nx+1)x(ny+1)-vertex displaced grid (disc, one boundary loop); SURVEY.md §8d config 2/3
// kind 1: nx x ny torus (closed, genus 1 -> topology-split events)
// kind 2: closed genus-0 "sphere" (grid rows between two pole fans -> interior start face)
// kind 3: grid with isolated missing cells (extra boundary loops -> hole handling)
// kind 4: two disjoint grids (two components)
static void make_mesh(int kind, int nx, int ny, uint64_t seed, MeshBuf &m) {
  Rng rng(seed);
  double amp = 0.25 * (0.8 + 0.4 * rng.uniform()), fx = 4.0 * (0.75 + 0.5 * rng.uniform()), fy = 6.0 * (0.75 + 0.5 * rng.uniform());
  double noise = 0.01;
  const double PI = 3.14159265358979323846;
  auto push_v = [&](double x, double y, double z, double nxn, double nyn, double nzn, double u, double v) {
    m.pos.push_back((float)x); m.pos.push_back((float)y); m.pos.push_back((float)z);
    double l = std::sqrt(nxn * nxn + nyn * nyn + nzn * nzn); if (l < 1e-12) { nxn = 0; nyn = 0; nzn = 1; l = 1; }
    m.nrm.push_back((float)(nxn / l)); m.nrm.push_back((float)(nyn / l)); m.nrm.push_back((float)(nzn / l));
    m.uv.push_back((float)u); m.uv.push_back((float)v);
  };
  auto tri = [&](uint32_t a, uint32_t b, uint32_t c) { m.faces.push_back(a); m.faces.push_back(b); m.faces.push_back(c); };
  auto grid = [&](int gx, int gy, double ox, const std::vector<uint8_t> *skip) {
    uint32_t base = (uint32_t)(m.pos.size() / 3);
    for (int j = 0; j <= gy; ++j)
      for (int i = 0; i <= gx; ++i) {
        double x = (double)i / gx, y = (double)j / gy;
        double z = amp * std::sin(fx * PI * x) * std::cos(fy * PI * y);
        double dzdx = amp * fx * PI * std::cos(fx * PI * x) * std::cos(fy * PI * y);
        double dzdy = -amp * fy * PI * std::sin(fx * PI * x) * std::sin(fy * PI * y);
        push_v(x + ox, y, z + noise * rng.normal(), -dzdx, -dzdy, 1.0, x, y);
      }
    for (int j = 0; j < gy; ++j)
      for (int i = 0; i < gx; ++i) {
        if (skip && (*skip)[(size_t)j * gx + i]) continue;
        uint32_t v00 = base + (uint32_t)(j * (gx + 1) + i), v10 = v00 + 1, v01 = v00 + (uint32_t)(gx + 1), v11 = v01 + 1;
        tri(v00, v10, v11); tri(v00, v11, v01);
      }
  };
  if (kind == 0) grid(nx, ny, 0, nullptr);
  else if (kind == 3) {
    std::vector<uint8_t> skip((size_t)nx * ny, 0);
    for (int j = 2; j + 2 < ny; j += 5) for (int i = 2; i + 2 < nx; i += 7) skip[(size_t)j * nx + i] = 1;
    grid(nx, ny, 0, &skip);
  } else if (kind == 4) { grid(nx, ny, 0, nullptr); grid(std::max(2, nx / 2), std::max(2, ny / 2), 1.5, nullptr); }
  else if (kind == 1) {
    double R = 1.0, r0 = 0.35;
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        double u = (double)i / nx, v = (double)j / ny;
        double th = 2 * PI * u, ph = 2 * PI * v;
        double r = r0 * (1.0 + 0.15 * std::sin(fx * th) * std::cos(fy * ph)) + noise * 0.3 * rng.normal();
        double cx = std::cos(th), sx = std::sin(th), cp = std::cos(ph), sp = std::sin(ph);
        push_v((R + r * cp) * cx, (R + r * cp) * sx, r * sp, cp * cx, cp * sx, sp, u, v);
      }
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        uint32_t i1 = (uint32_t)((i + 1) % nx), j1 = (uint32_t)((j + 1) % ny);
        uint32_t v00 = (uint32_t)(j * nx + i), v10 = (uint32_t)(j * nx) + i1, v01 = j1 * (uint32_t)nx + (uint32_t)i, v11 = j1 * (uint32_t)nx + i1;
        tri(v00, v10, v11); tri(v00, v11, v01);
      }
  } else if (kind == 2) {
    // rows 1..ny-1 of a lat/long sphere, periodic in i; poles are single vertices
    for (int j = 1; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        double th = 2 * PI * i / nx, ph = PI * j / ny;
        double r = 1.0 + 0.1 * std::sin(fx * th) * std::sin(fy * ph) + noise * rng.normal();
        double x = std::sin(ph) * std::cos(th), y = std::sin(ph) * std::sin(th), z = std::cos(ph);
        push_v(r * x, r * y, r * z, x, y, z, (double)i / nx, (double)j / ny);
      }
    uint32_t north = (uint32_t)(m.pos.size() / 3); push_v(0, 0, 1.0, 0, 0, 1, 0.5, 0);
    uint32_t south = north + 1; push_v(0, 0, -1.0, 0, 0, -1, 0.5, 1);
    for (int i = 0; i < nx; ++i) { uint32_t i1 = (uint32_t)((i + 1) % nx); tri(north, (uint32_t)i, i1); }
    for (int j = 0; j + 2 < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        uint32_t i1 = (uint32_t)((i + 1) % nx);
        uint32_t v00 = (uint32_t)(j * nx + i), v10 = (uint32_t)(j * nx) + i1, v01 = (uint32_t)((j + 1) * nx + i), v11 = (uint32_t)((j + 1) * nx) + i1;
        tri(v00, v01, v11); tri(v00, v11, v10);
      }
    uint32_t last = (uint32_t)((ny - 2) * nx);
    for (int i = 0; i < nx; ++i) { uint32_t i1 = (uint32_t)((i + 1) % nx); tri(south, last + i1, last + (uint32_t)i); }
  } else check(false, "unknown mesh kind");
}

}  // namespace synth

// -------------------------------------------------------------------- C ABI
extern "C" {

struct synth_options {
  int32_t pos_bits, uv_bits, normal_bits, single_connectivity, force_scheme, compression_level, pos_prediction, uv_prediction, normal_prediction, traversal_method, predictive_connectivity,
      normal_transform, raw_integers, no_prediction, generic_components;
};

static thread_local char g_err[256];
const char *synth_last_error(void) { return g_err; }

static synth::Options to_opt(const synth_options *o) {
  synth::Options r;
  if (o) {
    r.pos_bits = o->pos_bits; r.uv_bits = o->uv_bits; r.normal_bits = o->normal_bits;
    r.single_connectivity = o->single_connectivity; r.force_scheme = o->force_scheme;
    r.compression_level = o->compression_level; r.pos_prediction = o->pos_prediction; r.uv_prediction = o->uv_prediction;
    r.normal_prediction = o->normal_prediction; r.traversal_method = o->traversal_method;
    r.predictive_connectivity = o->predictive_connectivity;
    r.normal_transform = o->normal_transform; r.raw_integers = o->raw_integers; r.no_prediction = o->no_prediction; r.generic_components = o->generic_components;
  }
  return r;
}
void synth_default_options(synth_options *o) {
  synth::Options d;
  o->pos_bits = d.pos_bits; o->uv_bits = d.uv_bits; o->normal_bits = d.normal_bits; o->single_connectivity = d.single_connectivity;
  o->force_scheme = d.force_scheme; o->compression_level = d.compression_level; o->pos_prediction = d.pos_prediction; o->uv_prediction = d.uv_prediction;
  o->normal_prediction = d.normal_prediction; o->traversal_method = d.traversal_method;
  o->predictive_connectivity = d.predictive_connectivity;
  o->normal_transform = d.normal_transform; o->raw_integers = d.raw_integers; o->no_prediction = d.no_prediction; o->generic_components = d.generic_components;
}

// Encodes one mesh.  normals/uvs/generic may be NULL.  *out is malloc'ed; free with synth_free.
int synth_encode_mesh(const float *pos, uint32_t nv, const uint32_t *faces, uint32_t nf, const float *normals,
                      const float *uvs, const uint8_t *generic, const synth_options *opt, uint8_t **out, size_t *out_len) {
  try {
    synth::MeshIn in{pos, nv, faces, nf, normals, uvs, generic};
    std::vector<uint8_t> buf;
    synth::encode_mesh(in, to_opt(opt), buf);
    *out = (uint8_t *)malloc(buf.size());
    memcpy(*out, buf.data(), buf.size());
    *out_len = buf.size();
    return 0;
  } catch (const std::exception &e) { snprintf(g_err, sizeof(g_err), "%s", e.what()); return 1; }
}
// Attributes given per corner: value ids per corner of `faces` (3 * nf) into `normals` (nn rows) / `uvs` (nu rows); either
// id list may be NULL (that attribute is then per vertex, nv rows).  Interior edges whose end points carry different ids
// on their two faces become attribute seams: seam bits, attribute corner tables and corner attributes in the stream
// (MeshEdgeBreakerEncoder.cs:403-440, MeshAttributeCornerTable.cs:32-155).
int synth_encode_mesh_corners(const float *pos, uint32_t nv, const uint32_t *faces, uint32_t nf, const float *normals, uint32_t nn,
                              const uint32_t *normal_corners, const float *uvs, uint32_t nu, const uint32_t *uv_corners,
                              const synth_options *opt, uint8_t **out, size_t *out_len) {
  try {
    synth::MeshIn in{pos, nv, faces, nf, normals, uvs, nullptr, normals ? normal_corners : nullptr, nn, uvs ? uv_corners : nullptr, nu};
    for (size_t k = 0; k < (size_t)nf * 3; ++k) {
      synth::check(faces[k] < nv, "face index out of range");
      synth::check(!in.normal_corners || in.normal_corners[k] < nn, "normal id out of range");
      synth::check(!in.uv_corners || in.uv_corners[k] < nu, "texture coordinate id out of range");
    }
    std::vector<uint8_t> buf;
    synth::encode_mesh(in, to_opt(opt), buf);
    *out = (uint8_t *)malloc(buf.size());
    memcpy(*out, buf.data(), buf.size());
    *out_len = buf.size();
    return 0;
  } catch (const std::exception &e) { snprintf(g_err, sizeof(g_err), "%s", e.what()); return 1; }
}
int synth_encode_mesh_sequential(const float *pos, uint32_t nv, const uint32_t *faces, uint32_t nf, const float *normals,
                                 const float *uvs, int compressed, const synth_options *opt, uint8_t **out, size_t *out_len) {
  try {
    synth::MeshIn in{pos, nv, faces, nf, normals, uvs, nullptr};
    std::vector<uint8_t> buf;
    synth::encode_mesh_sequential(in, to_opt(opt), compressed != 0, buf);
    *out = (uint8_t *)malloc(buf.size() ? buf.size() : 1);
    memcpy(*out, buf.data(), buf.size());
    *out_len = buf.size();
    return 0;
  } catch (const std::exception &e) { snprintf(g_err, sizeof(g_err), "%s", e.what()); return 1; }
}
int synth_encode_point_cloud(const float *pos, uint32_t n, const synth_options *opt, uint8_t **out, size_t *out_len) {
  try {
    std::vector<uint8_t> buf;
    synth::encode_point_cloud(pos, n, to_opt(opt), buf);
    *out = (uint8_t *)malloc(buf.size());
    memcpy(*out, buf.data(), buf.size());
    *out_len = buf.size();
    return 0;
  } catch (const std::exception &e) { snprintf(g_err, sizeof(g_err), "%s", e.what()); return 1; }
}
void synth_free(uint8_t *p) { free(p); }

// Procedural mesh: returns counts; call once with NULL outputs to size, then again to fill.
int synth_make_mesh(int kind, int nx, int ny, uint64_t seed, uint32_t *nv, uint32_t *nf, float *pos, float *nrm, float *uv, uint32_t *faces) {
  try {
    synth::MeshBuf m;
    synth::make_mesh(kind, nx, ny, seed, m);
    *nv = (uint32_t)(m.pos.size() / 3); *nf = (uint32_t)(m.faces.size() / 3);
    if (pos) memcpy(pos, m.pos.data(), m.pos.size() * 4);
    if (nrm) memcpy(nrm, m.nrm.data(), m.nrm.size() * 4);
    if (uv) memcpy(uv, m.uv.data(), m.uv.size() * 4);
    if (faces) memcpy(faces, m.faces.data(), m.faces.size() * 4);
    return 0;
  } catch (const std::exception &e) { snprintf(g_err, sizeof(g_err), "%s", e.what()); return 1; }
}

// Batch: count meshes of one kind/size, seeds seed0..seed0+count-1, attribute mask bit0 normals, bit1 uvs.
// Streams are written back to back into one malloc'ed blob; offsets[count+1].
int synth_make_batch(int kind, int nx, int ny, uint64_t seed0, uint32_t count, int attr_mask, const synth_options *opt,
                     int num_threads, uint8_t **blob, uint64_t *offsets) {
  std::vector<std::vector<uint8_t>> streams(count);
  std::vector<std::string> errors(count);
  synth::Options o = to_opt(opt);
  if (num_threads < 1) num_threads = 1;
  auto work = [&](int t) {
    for (uint32_t i = (uint32_t)t; i < count; i += (uint32_t)num_threads) {
      try {
        synth::MeshBuf m;
        synth::make_mesh(kind, nx, ny, seed0 + i, m);
        synth::MeshIn in{m.pos.data(), (uint32_t)(m.pos.size() / 3), m.faces.data(), (uint32_t)(m.faces.size() / 3),
                         (attr_mask & 1) ? m.nrm.data() : nullptr, (attr_mask & 2) ? m.uv.data() : nullptr, nullptr};
        synth::encode_mesh(in, o, streams[i]);
      } catch (const std::exception &e) { errors[i] = e.what(); }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < num_threads; ++t) th.emplace_back(work, t);
  for (auto &x : th) x.join();
  uint64_t total = 0;
  for (uint32_t i = 0; i < count; ++i) {
    if (!errors[i].empty()) { snprintf(g_err, sizeof(g_err), "mesh %u: %s", i, errors[i].c_str()); return 1; }
    offsets[i] = total; total += streams[i].size();
  }
  offsets[count] = total;
  *blob = (uint8_t *)malloc(total ? total : 1);
  for (uint32_t i = 0; i < count; ++i) memcpy(*blob + offsets[i], streams[i].data(), streams[i].size());
  return 0;
}

// Entropy-coder primitives, exported for round-trip tests against the decoder.
int synth_encode_symbols(const uint32_t *v, size_t n, int nc, int force_scheme, int compression_level, uint8_t **out, size_t *out_len) {
  try {
    synth::ByteWriter w;
    std::vector<uint32_t> vals(v, v + n);
    synth::encode_symbols(w, vals, nc, force_scheme, compression_level);
    *out = (uint8_t *)malloc(w.d.size() ? w.d.size() : 1);
    memcpy(*out, w.d.data(), w.d.size());
    *out_len = w.d.size();
    return 0;
  } catch (const std::exception &e) { snprintf(g_err, sizeof(g_err), "%s", e.what()); return 1; }
}
int synth_encode_rabs(const uint8_t *bits, size_t n, uint8_t **out, size_t *out_len) {
  try {
    synth::ByteWriter w;
    std::vector<uint8_t> b(bits, bits + n);
    synth::write_rabs(w, b);
    *out = (uint8_t *)malloc(w.d.size());
    memcpy(*out, w.d.data(), w.d.size());
    *out_len = w.d.size();
    return 0;
  } catch (const std::exception &e) { snprintf(g_err, sizeof(g_err), "%s", e.what()); return 1; }
}

}  // extern "C"
