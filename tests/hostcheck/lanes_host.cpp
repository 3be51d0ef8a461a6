// tests/hostcheck/lanes_host.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The per-lane bodies of the lane-per-chain kernels (draco-sharp_amd/csrc/dsa_lanes.h: rANS symbol decode, prediction
// inverse) and the stream walk of k_locate (dsa_locate.h), compiled for the host with AddressSanitizer: the same
// source the GPU runs, one lane at a time, on an arena laid out by the library's own dsa_host_parse.h with every gap
// between regions poisoned.  Nothing here is linked into libdraco_mi355x.so and the product has no host decode path.
//
//   lanes_host decode <in.drc> <out.bin> [conn.bin]
//
// conn.bin (optional, written by the test from the oracle's connectivity): u32 F, NV, E, then opposite[3F],
// corner_to_vertex[3F] (reference corner numbering 3f+k), data_to_corner[E] -- what k_connectivity / k_traverse hand
// to the prediction kernels; with it the parallelogram attributes are decoded too.
#include <sanitizer/asan_interface.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../draco-sharp_amd/csrc/dsa_lanes.h"
#include "../../draco-sharp_amd/csrc/dsa_host_parse.h"

static std::vector<uint8_t> read_file(const char *path) {
  std::vector<uint8_t> v;
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  uint8_t buf[65536];
  size_t k;
  while ((k = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + k);
  fclose(f);
  return v;
}

// lanes_host octcheck <seed> <sequences>: the canonical-frame recursion of ln_predict_oct against the reference's
// ComputeOriginalValue applied entry by entry (oct_original), on random walks that keep crossing the axes, the diamond's
// edge and the square's edge, with now and then a correction from the whole int32 range.
static uint64_t rng_state = 1;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }
static int octcheck(uint64_t seed, int sequences) {
  uint64_t packed_steps = 0;
  rng_state = seed * 2654435761ull + 88172645463325252ull;
  for (int it = 0; it < sequences; ++it) {
    const int q = (it & 1) ? 2 + (int)(rnd() % 13) : 2 + (int)(rnd() % 29);     // 2..30 quantisation bits, half of the sequences within the packed step's 14
    const int32_t max_q = (int32_t)((1u << q) - 1u);
    const uint32_t entries = 1 + rnd() % 400;
    const int mode = (int)(rnd() % 5);
    const bool stored = rnd() % 3 != 0;        // corrections as a stream holds them: reduced modulo max_q
    std::vector<int32_t> w(2 * entries + 8), ref(2 * entries);
    const int64_t span = ((int64_t)1 << q);
    for (uint32_t i = 0; i < 2 * entries; ++i) {
      int64_t c;
      const uint32_t r = rnd();
      if (mode == 0) c = (int64_t)(r % 7) - 3;                              // noise around the current point
      else if (mode == 1) c = (int64_t)(r % (uint32_t)(span / 4 + 1)) - span / 8;      // quarter-range jumps
      else if (mode == 2) c = (int64_t)(r % (uint32_t)(2 * span)) - span;              // beyond the range: ModMax, edges
      else if (mode == 3) c = (r % 50 == 0) ? (int64_t)(int32_t)(rnd() * 2654435761u) : (int64_t)(r % 5) - 2;   // garbage now and then
      else c = (r % 9 == 0) ? ((int64_t)(r % (uint32_t)span) - span / 2) : (int64_t)(r % 3) - 1;
      if (stored && !(mode == 3 && r % 50 == 0)) { c %= (int64_t)max_q; if (c < 0) c += max_q; }
      w[i] = (int32_t)c;
    }
    dsa::OctParams o;
    const int32_t max_value = (1 << q) - 2;
    o.center = max_value / 2; o.max_q = (1 << q) - 1;
    int32_t ps = 0, pt = 0;
    for (uint32_t e = 0; e < entries; ++e) {
      int32_t os, ot;
      dsa::oct_original(o, true, ps, pt, w[2 * e], w[2 * e + 1], os, ot);
      ref[2 * e] = os; ref[2 * e + 1] = ot; ps = os; pt = ot;
    }
    // the packed 16-bit step of k_predict_oct_streams (octahedra of up to 14 bits), entry by entry as a lane runs it
    if (q <= OCT_PK_MAX_BITS) {
      dsa::OctPkLane st;
      dsa::oct_pk_init(st, o, (uint32_t)q);
      for (uint32_t e = 0; e < entries; ++e) {
        int32_t os, ot;
        if (dsa::oct_pk_entitled(st, w[2 * e], w[2 * e + 1])) ++packed_steps;
        dsa::oct_pk_entry(st, o, w[2 * e], w[2 * e + 1], os, ot);
        if (os != ref[2 * e] || ot != ref[2 * e + 1]) { fprintf(stderr, "octcheck: packed step, sequence %d (q %d, mode %d) differs at entry %u: (%d, %d) vs (%d, %d)\n", it, q, mode, e, os, ot, ref[2 * e], ref[2 * e + 1]); return 1; }
      }
    }
    dsa::lanes::ln_predict_oct(w.data(), entries, max_q, true);
    for (uint32_t i = 0; i < 2 * entries; ++i)
      if (w[i] != ref[i]) { fprintf(stderr, "octcheck: sequence %d (q %d, mode %d) differs at value %u: %d vs %d\n", it, q, mode, i, w[i], ref[i]); return 1; }
  }
  printf("octcheck: %d sequences equal (%llu entries on the packed step)\n", sequences, (unsigned long long)packed_steps);
  return 0;
}

// lanes_host octexhaust <bits>: the packed step against the reference's function on EVERY value of the square x every correction
// in [0, max_q], for octahedra of 2 .. <bits> bits
static int octexhaust(int bits) {
  for (int q = 2; q <= bits; ++q) {
    dsa::OctParams o;
    o.center = ((1 << q) - 2) / 2; o.max_q = (1 << q) - 1;
    long n = 0;
    for (int vs = 0; vs <= 2 * o.center; ++vs) for (int vt = 0; vt <= 2 * o.center; ++vt)
      for (int cx = 0; cx <= o.max_q; ++cx) for (int cy = 0; cy <= o.max_q; ++cy) {
        int32_t os, ot, rs, rt;
        dsa::oct_original(o, true, vs, vt, cx, cy, rs, rt);
        dsa::OctPkLane st;
        dsa::oct_pk_init(st, o, (uint32_t)q);
        st.P = ((uint32_t)(vs - o.center) & 0xFFFFu) | ((uint32_t)(vt - o.center) << 16);
        dsa::oct_pk_entry(st, o, cx, cy, os, ot);
        ++n;
        if (os != rs || ot != rt) { fprintf(stderr, "octexhaust: q %d value (%d, %d) correction (%d, %d): (%d, %d) vs (%d, %d)\n", q, vs, vt, cx, cy, os, ot, rs, rt); return 1; }
      }
    printf("octexhaust: %d bits, %ld pairs equal\n", q, n);
  }
  return 0;
}

// lanes_host divcheck <seed> <count>: div_trunc_pos (the 64-bit division of the GeometricNormal arithmetic through a double estimate
// put right by remainders) against the operator, on dividends of every magnitude and around exact multiples
static int divcheck(uint64_t seed, long count) {
  rng_state = seed * 2654435761ull + 88172645463325252ull;
  auto r64 = []() { return ((uint64_t)rnd() << 42) ^ ((uint64_t)rnd() << 21) ^ rnd(); };
  for (long it = 0; it < count; ++it) {
    const int bx = 1 + (int)(rnd() % 62), by = 1 + (int)(rnd() % 40);
    int64_t x = (int64_t)(r64() >> (64 - bx));
    if (rnd() & 1) x = -x;
    int64_t y = (int64_t)(r64() >> (64 - by));
    if (y <= 0) y = 1 + (int64_t)(rnd() % 7);
    if (it % 5 == 0) { const int64_t k = (int64_t)(rnd() % 1000000) - 500000; x = k * y + ((it % 3) - 1); }
    if (dsa::div_trunc_pos(x, y) != x / y) { fprintf(stderr, "divcheck: %lld / %lld: %lld vs %lld\n", (long long)x, (long long)y, (long long)dsa::div_trunc_pos(x, y), (long long)(x / y)); return 1; }
  }
  printf("divcheck: %ld divisions equal\n", count);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 4 && strcmp(argv[1], "divcheck") == 0) return divcheck(strtoull(argv[2], nullptr, 10), atol(argv[3]));
  if (argc >= 3 && strcmp(argv[1], "octexhaust") == 0) return octexhaust(atoi(argv[2]));
  if (argc >= 4 && strcmp(argv[1], "octcheck") == 0) return octcheck(strtoull(argv[2], nullptr, 10), atoi(argv[3]));
  if (argc < 4 || strcmp(argv[1], "decode") != 0) { fprintf(stderr, "usage: lanes_host decode <in.drc> <out.bin> [conn.bin]\n"); return 2; }
  const std::vector<uint8_t> data = read_file(argv[2]);
  const size_t len = data.size();
  FILE *out = fopen(argv[3], "wb");
  if (!out) { perror(argv[3]); return 2; }
  auto put32 = [&](uint32_t v) { fwrite(&v, 4, 1, out); };

  HostMesh h;
  unsetenv("DSA_FORCE_GENERAL");
  host_parse(data.data(), len, h);
  if (h.status != 0 || h.general) { put32(h.status ? (uint32_t)h.status : 1000u); put32(0); put32(0); fclose(out); return 0; }   // 1000: not a fast-path stream
  MeshLayout L;
  memset(&L, 0, sizeof(L));
  L.stream = 0;
  L.stream_len = (uint32_t)len;
  uint64_t cur = align_up(len + 1024, 256);
  std::vector<std::pair<uint64_t, uint64_t>> regions;
  regions.push_back({0, align_up(len, 16)});         // the byte ring and the tag decoder read 16-byte chunks
  cur = layout_mesh(h, len, L, cur, 1024, &regions);
  BatchGlobals G;
  memset(&G, 0, sizeof(G));
  G.pool = cur; G.pool_bytes = 1 << 20;
  regions.push_back({cur, G.pool_bytes});
  cur += G.pool_bytes;
  std::vector<uint8_t> arena(cur, 0);
  memcpy(arena.data(), data.data(), len);
  ASAN_POISON_MEMORY_REGION(arena.data(), arena.size());
  for (auto &rg : regions) ASAN_UNPOISON_MEMORY_REGION(arena.data() + rg.first, rg.second);

  MeshDesc D;
  memset(&D, 0, sizeof(D));
  uint32_t s_cum[LOC_MAX_TAGS + 1];
  std::vector<uint32_t> s_lut(LOC_LDS_WORDS);
  dsa::locate_all(arena.data(), L, &D, &G, s_cum, s_lut.data());
  std::vector<int> decoded(DSA_MAX_ATT, 0);
  if (D.status == ST_OK) {
    // ---- k_symbols_lanes, one lane
    for (uint32_t ai = 0; ai < D.num_attributes; ++ai) {
      const AttrDesc &a = D.att[ai];
      if (!dsa::lanes::ln_sym_eligible(a, L, ai, LN_FLAG_SYMBOLS)) continue;
      const uint32_t cap = dsa::lanes::ln_sym_tier(a.num_distinct);
      std::vector<uint16_t> lds(dsa::lanes::ln_sym_stride(cap) / 2);
      dsa::lanes::ln_symbols_stream(arena.data(), L, &D, ai, cap, lds.data());
      decoded[ai] = 1;
    }
    // ---- k_predict_lanes phase 0
    for (uint32_t ai = 0; ai < D.num_attributes && D.status == ST_OK; ++ai)
      if (decoded[ai]) dsa::lanes::ln_predict_attribute(arena.data(), L, &D, ai, 0);
    // ---- parallelogram attributes: connectivity from the oracle, operands by the product's para_operands_of
    bool have_conn = false;
    if (argc >= 5 && D.status == ST_OK && D.encoder_type == 1) {
      const std::vector<uint8_t> cb = read_file(argv[4]);
      const uint32_t *c = (const uint32_t *)cb.data();
      const uint32_t F = c[0], NV = c[1], E = c[2];
      const uint32_t *opp = c + 3, *c2v = opp + 3 * (size_t)F, *d2c_ref = c2v + 3 * (size_t)F;
      if (F == L.cap_faces && E <= L.cap_vertices && NV <= L.cap_vertices) {
        // 32-byte records in a buffer of their own (the arena's region holds the kernels' 16-byte records for a mesh of this size)
        std::vector<uint32_t> frec_wide((size_t)F * 8 + 8);
        uint32_t *frec = frec_wide.data(), *d2c = (uint32_t *)(arena.data() + L.d2c), *para = (uint32_t *)(arena.data() + L.para);
        int32_t *v2d = (int32_t *)(arena.data() + L.v2d);
        auto quad = [](uint32_t cr) { return cr == DSA_INVALID ? cr : 4u * (cr / 3u) + cr % 3u; };
        for (uint32_t f = 0; f < F; ++f) {
          for (uint32_t k = 0; k < 3; ++k) { frec[8 * f + k] = c2v[3 * f + k]; frec[8 * f + 4 + k] = quad(opp[3 * f + k]); }
          frec[8 * f + 3] = 0; frec[8 * f + 7] = 0;
        }
        for (uint32_t v = 0; v < L.cap_vertices; ++v) v2d[v] = -1;
        for (uint32_t p = 0; p < E; ++p) { d2c[p] = quad(d2c_ref[p]); v2d[c2v[d2c_ref[p]]] = (int32_t)p; }
        for (uint32_t p = 0; p < E; ++p) dsa::para_operands_of(p, frec, d2c, v2d, F, NV, para);
        have_conn = true;
      }
    }
    for (uint32_t ai = 0; ai < D.num_attributes && D.status == ST_OK; ++ai) {
      if (!decoded[ai]) continue;
      const AttrDesc &a = D.att[ai];
      if (a.have_scheme && a.pred_kind == 1) {
        if (have_conn) dsa::lanes::ln_predict_attribute(arena.data(), L, &D, ai, 1);
        else decoded[ai] = 0;
      }
    }
  }
  ASAN_UNPOISON_MEMORY_REGION(arena.data(), arena.size());
  put32((uint32_t)D.status); put32((uint32_t)D.detail);
  if (D.status != ST_OK) { put32(0); fclose(out); return 0; }
  put32(D.num_attributes);
  for (uint32_t ai = 0; ai < D.num_attributes; ++ai) {
    const AttrDesc &a = D.att[ai];
    put32((uint32_t)decoded[ai]); put32(a.num_entries); put32(a.nc_portable);
    if (decoded[ai]) fwrite(arena.data() + L.work[ai], 4, (size_t)a.num_entries * a.nc_portable, out);
  }
  fclose(out);
  return 0;
}
