// tests/hostcheck/general_host.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The serial general path of the product (draco-sharp_amd/csrc/dsa_general.h: valence traversal, attribute seams,
// corner attributes, TexCoordsPortable) compiled for the host with AddressSanitizer, so that corrupt streams can be
// thrown at exactly the code the GPU runs without risking a device fault (GPU sanitizers are not available on the
// pool).  The arena is laid out by the same dsa_host_parse.h the library uses, with every gap between regions
// poisoned.  Nothing here is linked into libdraco_mi355x.so and the product has no host decode path.
//
//   general_host decode <in.drc> <out.bin> [force]     one stream; results for comparison with the oracle
//   general_host fuzz <in.drc> <iterations> <seed> [force]   random corruptions; prints outcome counts
#include <sanitizer/asan_interface.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../draco-sharp_amd/csrc/dsa_general.h"
#include "../../draco-sharp_amd/csrc/dsa_host_parse.h"

namespace {

struct Result {
  MeshDesc D;
  MeshLayout L;
  std::vector<uint8_t> arena;
  bool general = false;
};

// what k_locate does for a general mesh: header + metadata, then hand the reader to the general path
static bool run(const uint8_t *data, size_t len, bool force, Result &R) {
  HostMesh h;
  if (force) setenv("DSA_FORCE_GENERAL", "1", 1); else unsetenv("DSA_FORCE_GENERAL");
  host_parse(data, len, h);
  memset(&R.D, 0, sizeof(R.D));
  memset(&R.L, 0, sizeof(R.L));
  R.general = h.status == 0 && h.general;
  if (h.status != 0) { R.D.status = h.status; return true; }
  if (!h.general) return false;
  R.L.stream = 0;
  R.L.stream_len = (uint32_t)len;
  uint64_t cur = align_up(len + 1024, 256);
  std::vector<std::pair<uint64_t, uint64_t>> regions;
  regions.push_back({0, len});
  cur = layout_mesh(h, len, R.L, cur, 1024, &regions);
  R.arena.assign(cur, 0);
  memcpy(R.arena.data(), data, len);
  ASAN_POISON_MEMORY_REGION(R.arena.data(), R.arena.size());
  for (auto &rg : regions) ASAN_UNPOISON_MEMORY_REGION(R.arena.data() + rg.first, rg.second);
  // header, DracoDecoder.cs:44-64 (k_locate)
  HRd hr(data, len);
  hr.pos = 5;
  R.D.major = (uint8_t)hr.u8(); R.D.minor = (uint8_t)hr.u8(); R.D.encoder_type = (uint8_t)hr.u8(); R.D.encoder_method = (uint8_t)hr.u8();
  uint32_t flags = hr.u8(); flags |= hr.u8() << 8;
  R.D.flags = (uint16_t)flags;
  if (flags & 0x8000) {
    uint32_t natt = (uint32_t)hr.varint();
    for (uint32_t i = 0; i < natt && hr.ok; ++i) { (void)hr.varint(); skip_metadata_element(hr, 0); }
    skip_metadata_element(hr, 0);
  }
  R.D.general = 1;
  dsa::Rd r(R.arena.data(), (uint32_t)len, (uint32_t)hr.pos);
  std::vector<uint16_t> lut(GEN_LUT_SLOTS);
  std::vector<uint16_t> fcum(GEN_LUT_SYMS + 1);
  std::vector<uint32_t> fcum32(GEN_LUT_SYMS + 1);
  dsa::gen::RansScratch rs = {lut.data(), getenv("DSA_HALF_LUT") ? 1u : 0u, fcum.data(), fcum32.data(), nullptr, 0};   // both LUT resolutions of the device launches
  if (R.D.encoder_method == 0) (void)dsa::gen::decode_sequential_mesh(R.arena.data(), R.L, &R.D, r, rs);
  else (void)dsa::gen::decode_mesh(R.arena.data(), R.L, &R.D, r, rs);
  ASAN_UNPOISON_MEMORY_REGION(R.arena.data(), R.arena.size());
  return true;
}

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

struct Rng { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); } };

}  // namespace

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: general_host decode <in> <out> [force] | fuzz <in> <iterations> <seed> [force]\n"); return 2; }
  const std::string mode = argv[1];
  std::vector<uint8_t> in = read_file(argv[2]);
  if (mode == "layout") {          // the sizing parse and the layout of a stream as the batch builder makes them: what the fast kernels are given
    HostMesh h;
    unsetenv("DSA_FORCE_GENERAL");
    host_parse(in.data(), in.size(), h);
    MeshLayout L;
    memset(&L, 0, sizeof(L));
    uint64_t end = h.status == 0 ? layout_mesh(h, in.size(), L, 0, 16) : 0;
    printf("status %d general %d seamed %d first_method %d mp_att %u tc0 %llu tc0_bytes %llu cap_vertices %u end %llu\n", h.status, (int)h.general, (int)h.seamed, h.first_method,
           L.mp_att, (unsigned long long)L.tc[0], (unsigned long long)(L.mp_att & 1u ? mp_region_bytes(L.cap_vertices) : 0), L.cap_vertices, (unsigned long long)end);
    return 0;
  }
  if (mode == "decode") {
    const bool force = argc > 4;
    Result R;
    if (!run(in.data(), in.size(), force, R)) { printf("not-general\n"); return 3; }
    // out.bin: status, detail, F, points, natt, then faces i32[3F], then per attribute: entries, nc_portable, map u32[points], portable i32[entries*ncp]
    FILE *f = fopen(argv[3], "wb");
    if (!f) { perror(argv[3]); return 2; }
    auto w32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); };
    w32((uint32_t)R.D.status); w32((uint32_t)R.D.detail);
    if (R.D.status == 0) {
      w32(R.D.num_faces); w32(R.D.num_points); w32(R.D.num_attributes);
      fwrite(R.arena.data() + R.L.faces, 4, (size_t)3 * R.D.num_faces, f);
      for (uint32_t a = 0; a < R.D.num_attributes; ++a) {
        const AttrDesc &A = R.D.att[a];
        const uint32_t ncp = A.seq_type == 0 ? 0u : A.nc_portable;
        w32(A.num_entries); w32(ncp);
        fwrite(R.arena.data() + R.L.map[a], 4, R.D.num_points, f);
        fwrite(R.arena.data() + R.L.work[a], 4, (size_t)A.num_entries * ncp, f);
      }
    }
    fclose(f);
    printf("status %d detail %d\n", R.D.status, R.D.detail);
    return 0;
  }
  if (mode == "fuzz") {
    const int iters = atoi(argv[3]);
    Rng rng{(uint64_t)strtoull(argv[4], nullptr, 10)};
    const bool force = argc > 5;
    int ok = 0, invalid = 0, notimpl = 0, notgen = 0;
    for (int it = 0; it < iters; ++it) {
      std::vector<uint8_t> m = in;
      const int kind = (int)(rng.next() % 4);
      if (kind == 0) { int k = 1 + (int)(rng.next() % 4); for (int q = 0; q < k; ++q) m[rng.next() % m.size()] ^= (uint8_t)(1u << (rng.next() % 8)); }       // bit flips
      else if (kind == 1) { int k = 1 + (int)(rng.next() % 8); for (int q = 0; q < k; ++q) m[rng.next() % m.size()] = (uint8_t)rng.next(); }                  // random bytes
      else if (kind == 2) { m.resize(11 + rng.next() % (m.size() - 11)); }                                                                                   // truncation
      else { size_t at = 11 + rng.next() % (m.size() - 11), k = 1 + rng.next() % 16; for (size_t q = 0; q < k && at + q < m.size(); ++q) m[at + q] = (uint8_t)rng.next(); }   // burst
      Result R;
      if (!run(m.data(), m.size(), force, R)) { ++notgen; continue; }
      if (R.D.status == 0) ++ok; else if (R.D.status == 1) ++invalid; else ++notimpl;
    }
    printf("ok %d invalid %d notimpl %d notgeneral %d\n", ok, invalid, notimpl, notgen);
    return 0;
  }
  return 2;
}
