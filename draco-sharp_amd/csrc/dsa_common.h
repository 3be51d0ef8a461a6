// draco-sharp_amd/csrc/dsa_common.h
// Pieces shared by the kernels (dsa_kernels.h) and the serial general path (dsa_general.h): failure latch, byte /
// bit readers, rABS, probability-table parse, prediction transforms.  Nothing here uses wave intrinsics, so the
// general path can also be compiled for the host under AddressSanitizer (tests/hostcheck: test infrastructure
// only -- the product library is built by hipcc alone and has no host decode path).
#pragma once
#include <stdint.h>

#include "dsa_types.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else
#define __device__
#define __host__
#define __forceinline__ inline
#include <string.h>
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#endif

namespace dsa {

#define WAVE 64

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ void fail(MeshDesc *d, int code, int site) {
#if defined(__HIPCC__)
  if (atomicCAS(&d->status, ST_OK, code) == ST_OK) d->detail = site;
#else
  if (d->status == ST_OK) { d->status = code; d->detail = site; }
#endif
}
// REQUIRE/NOTIMPL latch the first failure of a mesh and leave the current function with RET.
#define RET
#define REQUIRE(cond, site)                  \
  do {                                       \
    if (!(cond)) {                           \
      fail(D, ST_INVALID, (site));           \
      return RET;                            \
    }                                        \
  } while (0)
#define NOTIMPL(site)                        \
  do {                                       \
    fail(D, ST_NOTIMPL, (site));             \
    return RET;                              \
  } while (0)
// status as seen through L2 (a plain load may hit a stale L1 line after an atomic by this CU)
#if defined(__HIPCC__)
__device__ __forceinline__ int status_of(MeshDesc *d) { return __hip_atomic_load(&d->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
inline int status_of(MeshDesc *d) { return d->status; }
#endif

// Bounds-checked little-endian byte reader over one compressed stream
// (DecoderBuffer.cs:26-120).  A failed read latches ok=false and returns 0.
struct Rd {
  const uint8_t *p;
  uint32_t n, pos;
  bool ok;
  __device__ __forceinline__ Rd(const uint8_t *d, uint32_t len, uint32_t at) : p(d), n(len), pos(at), ok(at <= len) {}
  __device__ __forceinline__ uint32_t u8() {
    if (pos < n) return p[pos++];
    ok = false;
    return 0;
  }
  __device__ __forceinline__ uint32_t u16() { uint32_t a = u8(); return a | (u8() << 8); }
  __device__ __forceinline__ uint32_t u32() { uint32_t a = u8(); a |= u8() << 8; a |= u8() << 16; return a | (u8() << 24); }
  __device__ __forceinline__ float f32() { return __uint_as_float(u32()); }
  __device__ __forceinline__ uint64_t varint() {
    uint64_t r = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      uint32_t b = u8();
      r |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return r;
    }
    ok = false;
    return r;
  }
  __device__ __forceinline__ void skip(uint64_t k) {
    if (!ok || k > (uint64_t)(n - pos)) { ok = false; pos = n; }
    else pos += (uint32_t)k;
  }
};

// Up to 32 bits at an arbitrary bit position of an LSB-first bit section
// (DecoderBuffer.cs:138-154); bytes past `n` read as 0.
__device__ __forceinline__ uint32_t read_bits(const uint8_t *p, uint32_t n, uint64_t bitpos, uint32_t count) {
  uint32_t byte = (uint32_t)(bitpos >> 3), sh = (uint32_t)(bitpos & 7);
  uint64_t w = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    uint32_t b = (byte + k < n) ? p[byte + k] : 0u;
    w |= (uint64_t)b << (8 * k);
  }
  w >>= sh;
  return count >= 32 ? (uint32_t)w : ((uint32_t)w & ((1u << count) - 1u));
}

// rABS binary decoder (Entropy/AnsDecoder.cs:12-56, BitCoders/RAnsBitDecoder.cs:12-24)
struct Rabs {
  const uint8_t *buf;
  uint32_t off, state, p;   // p = 256 - prob_zero
  bool ok;
  __device__ __forceinline__ void start(const uint8_t *s, uint32_t slen, uint32_t at, uint32_t *end_pos) {
    Rd r(s, slen, at);
    uint32_t prob_zero = r.u8();
    uint64_t size = r.varint();
    uint32_t begin = r.pos;
    r.skip(size);
    ok = r.ok && size >= 1;
    *end_pos = r.pos;
    p = 256u - prob_zero;
    buf = s + begin;
    off = 0; state = 4096;
    if (!ok) return;
    uint32_t o = (uint32_t)size;
    uint32_t x = buf[o - 1] >> 6;
    if (x == 0) { off = o - 1; state = buf[o - 1] & 0x3F; }
    else if (x == 1) { if (o < 2) { ok = false; return; } off = o - 2; state = ((uint32_t)buf[o - 2] | ((uint32_t)buf[o - 1] << 8)) & 0x3FFF; }
    else if (x == 2) { if (o < 3) { ok = false; return; } off = o - 3; state = ((uint32_t)buf[o - 3] | ((uint32_t)buf[o - 2] << 8) | ((uint32_t)buf[o - 1] << 16)) & 0x3FFFFF; }
    else { ok = false; return; }
    state += 4096;
    if (state >= 4096u * 256u) ok = false;
  }
  __device__ __forceinline__ uint32_t next() {
    if (state < 4096 && off > 0) state = state * 256 + buf[--off];
    uint32_t x = state, quot = x >> 8, rem = x & 255, xn = quot * p;
    bool val = rem < p;
    state = val ? xn + rem : x - xn - p;
    return val ? 1u : 0u;
  }
};

// rANS stream tail -> initial state (Entropy/RAnsDecoder.cs:20-54)
__device__ __forceinline__ bool rans_init(const uint8_t *buf, uint32_t size, uint32_t l_base, uint32_t *state, uint32_t *off) {
  if (size < 1) return false;
  uint32_t x = buf[size - 1] >> 6, st, o;
  if (x == 0) { o = size - 1; st = buf[size - 1] & 0x3F; }
  else if (x == 1) { if (size < 2) return false; o = size - 2; st = ((uint32_t)buf[size - 2] | ((uint32_t)buf[size - 1] << 8)) & 0x3FFF; }
  else if (x == 2) { if (size < 3) return false; o = size - 3; st = ((uint32_t)buf[size - 3] | ((uint32_t)buf[size - 2] << 8) | ((uint32_t)buf[size - 1] << 16)) & 0x3FFFFF; }
  else { if (size < 4) return false; o = size - 4; st = ((uint32_t)buf[size - 4] | ((uint32_t)buf[size - 3] << 8) | ((uint32_t)buf[size - 2] << 16) | ((uint32_t)buf[size - 1] << 24)) & 0x3FFFFFFF; }
  st += l_base;
  if (st >= l_base * 256u) return false;
  *state = st; *off = o;
  return true;
}

// Reads the probability table of an rANS symbol stream into prob[0..n) (lane 0)
// (Entropy/RAnsSymbolDecoder.cs:21-48).  Returns false on malformed input.
__device__ bool read_prob_table(Rd &r, uint32_t n, uint32_t *prob) {
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t pd = r.u8();
    uint32_t token = pd & 3;
    if (token == 3) {
      uint32_t offset = pd >> 2;
      if (i + offset >= n) return false;
      for (uint32_t j = 0; j <= offset; ++j) prob[i + j] = 0;
      i += offset;
    } else {
      uint32_t pr = pd >> 2;
      for (uint32_t k = 0; k < token; ++k) pr |= r.u8() << (8 * (k + 1) - 2);
      prob[i] = pr;
    }
  }
  return r.ok;
}
// Same walk without storing (k_locate only needs to know where the table ends); counts the non-zero frequencies.
__device__ bool skip_prob_table(Rd &r, uint32_t n, uint32_t *distinct) {
  uint32_t nz = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t pd = r.u8();
    uint32_t token = pd & 3;
    if (token == 3) {
      uint32_t offset = pd >> 2;
      if (i + offset >= n) return false;
      i += offset;
    } else {
      uint32_t pr = pd >> 2;
      for (uint32_t k = 0; k < token; ++k) pr |= r.u8() << (8 * (k + 1) - 2);
      nz += pr != 0;
    }
  }
  *distinct = nz;
  return r.ok;
}
__device__ __forceinline__ uint32_t rans_precision_bits(uint32_t max_bit_length) {   // Entropy/RAnsSymbolCoding.cs:10-27
  uint32_t p = (3 * max_bit_length) / 2;
  return p < 12 ? 12 : (p > 20 ? 20 : p);
}

__device__ __forceinline__ uint32_t data_type_length(uint32_t dt) {   // Constants.cs:134-150
  switch (dt) {
    case 1: case 2: case 11: return 1;
    case 3: case 4: return 2;
    case 5: case 6: case 9: return 4;
    case 7: case 8: case 10: return 8;
    default: return 0;
  }
}

// Internal corner ids are "quad coded": corner k of face f is 4*f + k (face = c >> 2, k = c & 3, no division), and a
// face record is 32 bytes: {v0, v1, v2, flags, o0, o1, o2, 0} (vertices, opposite corners).
__device__ __forceinline__ uint32_t qnext(uint32_t c) { return (c & 3u) == 2u ? c - 2u : c + 1u; }
__device__ __forceinline__ uint32_t qprev(uint32_t c) { return (c & 3u) == 0u ? c + 2u : c - 1u; }
__device__ __forceinline__ uint32_t fv_idx(uint32_t c) { return 2u * c - (c & 3u); }        // dword index of the vertex slot
__device__ __forceinline__ uint32_t fo_idx(uint32_t c) { return 2u * c - (c & 3u) + 4u; }   // dword index of the opposite slot

// Parallelogram operands of entry p (MeshPredictionSchemeParallelogramDecoder.cs:56-89):
// para[3p..3p+2] = entries (next, prev, opposite), or next = INVALID when the entry falls back to delta.
__device__ __forceinline__ void para_operands_of(uint32_t p, const uint32_t *frec, const uint32_t *d2c, const int32_t *v2d, uint32_t F, uint32_t NV, uint32_t *para) {
  uint32_t en = DSA_INVALID, ep = 0, eo = 0;
  if (p > 0) {
    const uint32_t c0 = d2c[p];
    const uint32_t oci = (c0 < 4 * F && (c0 & 3u) != 3u) ? frec[fo_idx(c0)] : DSA_INVALID;
    if (oci != DSA_INVALID && oci < 4 * F && (oci & 3u) != 3u) {
      const uint32_t *fr = frec + (size_t)(oci >> 2) * 8;
      const uint32_t fx = fr[0], fy = fr[1], fz = fr[2];
      const uint32_t k = oci & 3u;
      const uint32_t a = k == 0 ? fx : (k == 1 ? fy : fz), b = k == 0 ? fy : (k == 1 ? fz : fx), c = k == 0 ? fz : (k == 1 ? fx : fy);
      if (a < NV && b < NV && c < NV) {
        const int32_t vo = v2d[a], vn = v2d[b], vp = v2d[c];
        if (vo >= 0 && vn >= 0 && vp >= 0 && (uint32_t)vo < p && (uint32_t)vn < p && (uint32_t)vp < p) { en = (uint32_t)vn; ep = (uint32_t)vp; eo = (uint32_t)vo; }
      }
    }
  }
  para[3 * p] = en; para[3 * p + 1] = ep; para[3 * p + 2] = eo;
}

// ---- staging of a stream that is consumed from its tail (rANS: state = state * 256 + buf[--offset], RAnsDecoder.cs:58-61)
// Global memory answers in 0.5-1 us when the chip is busy, and the memory counter of a wave is in order (waiting for
// a load waits for every older store as well), so the serial decoders touch global memory only at the boundaries of
// 16-symbol blocks: the stream sits in an LDS ring of eight 16-byte chunks (chunk c = arena bytes
// [base - 16(c+1), base - 16c), base = the 16-byte boundary behind the first byte to take), two chunks are requested
// at a boundary and written into the ring at the next one.  The q-th byte taken, counted from `base` down, is ring
// byte (q ^ 15) & 127.  A 12..15-bit rANS step takes at most two bytes, a block at most two chunks.
#define LN_RING_CHUNKS 8u
#define LN_BLOCK 16u
// 16 bytes of the arena as four dwords (global_load_dwordx4 on the device)
struct Chunk { uint32_t d[4]; };
__device__ __forceinline__ Chunk ln_load_chunk(const uint8_t *arena, uint64_t base, uint64_t lowest, uint32_t c) {
  const uint64_t want = 16ull * (c + 1ull);
  const uint64_t at = (base >= want && base - want >= lowest) ? base - want : lowest;     // below the stream: never consumed
#if defined(__HIPCC__)
  const uint4 v = *(const uint4 *)(arena + at);
  Chunk r; r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w;
  return r;
#else
  Chunk r; memcpy(r.d, arena + at, 16);
  return r;
#endif
}

struct OctParams { int32_t max_q, center; };

__device__ __forceinline__ void oct_invert_diamond(int32_t center, int32_t &s, int32_t &t) {   // OctahedronToolBox.cs:152-196
  int32_t ss, st;
  if (s >= 0 && t >= 0) { ss = 1; st = 1; }
  else if (s <= 0 && t <= 0) { ss = -1; st = -1; }
  else { ss = s > 0 ? 1 : -1; st = t > 0 ? 1 : -1; }
  int32_t cs = ss * center, ct = st * center;
  int32_t us = s + s - cs, ut = t + t - ct, tmp = us;
  if (ss * st >= 0) { us = -ut; ut = -tmp; } else { us = ut; ut = tmp; }
  us += cs; ut += ct;
  s = us / 2; t = ut / 2;
}
__device__ __forceinline__ int32_t oct_mod_max(const OctParams &o, int32_t x) {   // OctahedronToolBox.cs:206-213
  if (x > o.center) return x - o.max_q;
  return x < -o.center ? x + o.max_q : x;
}
__device__ __forceinline__ void oct_rotate(int32_t &x, int32_t &y, int rot) {
  int32_t a = x, b = y;
  if (rot == 1) { x = b; y = -a; } else if (rot == 2) { x = -a; y = -b; } else if (rot == 3) { x = -b; y = a; }
}
// PredictionSchemeNormalOctahedron(Canonicalized)DecodingTransform.ComputeOriginalValue
__device__ __forceinline__ void oct_original(const OctParams &o, bool canonical, int32_t ps, int32_t pt, int32_t c0, int32_t c1,
                                             int32_t &os, int32_t &ot) {
  ps -= o.center; pt -= o.center;
  int32_t aps = ps < 0 ? -ps : ps, apt = pt < 0 ? -pt : pt;
  bool in_d = (uint32_t)aps + (uint32_t)apt <= (uint32_t)o.center;
  if (!in_d) oct_invert_diamond(o.center, ps, pt);
  bool bottom_left = true;
  int rot = 0;
  if (canonical) {
    bottom_left = (ps == 0 && pt == 0) || (ps < 0 && pt <= 0);
    if (ps == 0) rot = pt == 0 ? 0 : (pt > 0 ? 3 : 1);
    else if (ps > 0) rot = pt >= 0 ? 2 : 1;
    else rot = pt <= 0 ? 0 : 3;
    if (!bottom_left) oct_rotate(ps, pt, rot);
  }
  os = oct_mod_max(o, (int32_t)((uint32_t)ps + (uint32_t)c0));
  ot = oct_mod_max(o, (int32_t)((uint32_t)pt + (uint32_t)c1));
  if (canonical && !bottom_left) oct_rotate(os, ot, (4 - rot) % 4);
  if (!in_d) oct_invert_diamond(o.center, os, ot);
  os += o.center; ot += o.center;
}

// The canonicalised transform as a recursion in the canonical frame.  ComputeOriginalValue maps the previous value p
// (centred) through T = R_k . I^inv (I = InvertDiamond if p lies outside the diamond, R_k = the rotation that takes the result to
// the bottom-left quadrant), adds the correction there (w = T p + corr), wraps (ModMax) and maps back: p' = I^inv R_-k w.
// With q = R_-k w and w (after ModMax) within the square the next step's frame follows from w alone:
//   |w|_1 <= center (w inside the diamond; strictly, when inv):  I(p') = q again (I undoes itself on the open diamond), the
//       rotation counts add up (rotcount(R_-k w) = k + rotcount(w) mod 4: the four half-open quadrants are each other's
//       images) and the next canonical point is w turned into the bottom-left quadrant: (-|w.x|, -|w.y|), swapped when
//       rotcount(w) is odd; inv stays (it ends on the edge itself, which I leaves in place);
//   |w|_1 > center (the normal crosses to the other half of the octahedron): with y = I(q), the next step starts from y
//       whether p' = q (it will invert it) or p' = y (inside the diamond): canonical point and rotation count of y, inv flips.
// (I and the rotations do not commute on the axes, so y is computed from q, as the reference does, not from w.)
// The state (u, k, inv) advances in 10-30 dependent operations; the value, (inv ? I(q) : q) + center, hangs off the chain.
// A correction that leaves the square, and out-of-range garbage, take the reference's own function and re-derive the state
// from its result.
struct OctLane { int32_t ux, uy, vs, vt; uint32_t k; bool inv, regular; };
__device__ __forceinline__ uint32_t oct_rotcount(int32_t x, int32_t y) {
  return (y < 0 && x >= 0) ? 1u : ((x > 0 && y >= 0) ? 2u : ((y > 0 && x <= 0) ? 3u : 0u));
}
__device__ __forceinline__ void oct_lane_state(const OctParams &o, OctLane &s) {       // from s.vs, s.vt, as ComputeOriginalValue starts
  int32_t ps = (int32_t)((uint32_t)s.vs - (uint32_t)o.center), pt = (int32_t)((uint32_t)s.vt - (uint32_t)o.center);
  const int32_t aps = ps < 0 ? -ps : ps, apt = pt < 0 ? -pt : pt;
  const bool in_d = (uint32_t)aps + (uint32_t)apt <= (uint32_t)o.center;
  // values far outside the square (only a damaged stream has them) stay on the reference's function
  const bool tame = (uint32_t)aps <= 2u * (uint32_t)o.center + 2u && (uint32_t)apt <= 2u * (uint32_t)o.center + 2u;
  if (!tame) { s.regular = false; s.ux = 0; s.uy = 0; s.k = 0; s.inv = false; return; }
  if (!in_d) oct_invert_diamond(o.center, ps, pt);
  const uint32_t rot = oct_rotcount(ps, pt);
  oct_rotate(ps, pt, (int)rot);
  s.ux = ps; s.uy = pt; s.k = rot; s.inv = !in_d;
  s.regular = ps <= 0 && pt <= 0 && ps >= -o.center && pt >= -o.center;
}
// One entry on the fast path only: advances (u, k, inv) and returns the value; the result says whether the step was entitled
// to (when it is not, state and value are garbage and the caller redoes the entry with oct_lane_exact).
__device__ __forceinline__ bool oct_lane_fast(const OctParams &o, OctLane &s, int32_t cx, int32_t cy, int32_t &os, int32_t &ot) {
  const int32_t C = o.center;
  // ModMax once, as the reference does (corrections are stored modulo max_q: -2 arrives as max_q - 2); whatever is still
  // outside the square afterwards is not this path's business
  const int32_t wx = oct_mod_max(o, (int32_t)((uint32_t)s.ux + (uint32_t)cx)), wy = oct_mod_max(o, (int32_t)((uint32_t)s.uy + (uint32_t)cy));
  const int32_t ax = wx < 0 ? -wx : wx, ay = wy < 0 ? -wy : wy;
  const bool sane = (uint32_t)(cx + (1 << 30)) <= (1u << 31) && (uint32_t)(cy + (1 << 30)) <= (1u << 31);   // no overflow above (|u| <= center < 2^29)
  const bool ok = s.regular && sane && (uint32_t)ax <= (uint32_t)C && (uint32_t)ay <= (uint32_t)C;
  const int32_t l1 = ax + ay;                   // meaningful when ok
  const bool out = l1 > C;
  const uint32_t k = s.k;
  int32_t qs = wx, qt = wy;
  oct_rotate(qs, qt, (int)((4u - k) & 3u));
  int32_t ys = qs, yt = qt;
  if (s.inv || out) oct_invert_diamond(C, ys, yt);
  os = (s.inv ? ys : qs) + C; ot = (s.inv ? yt : qt) + C;
  // ---- the next frame
  const int32_t bx = out ? ys : wx, by = out ? yt : wy;
  const int32_t abx = bx < 0 ? -bx : bx, aby = by < 0 ? -by : by;
  const uint32_t kk = oct_rotcount(bx, by);
  const bool odd = (kk & 1u) != 0;
  s.ux = -(odd ? aby : abx); s.uy = -(odd ? abx : aby);
  s.k = out ? kk : ((abx | aby) == 0 ? 0u : ((k + kk) & 3u));
  s.inv = out ? !s.inv : (s.inv && l1 < C);
  return ok;
}
// One entry by the reference's function on the previous value (s.vs, s.vt), then the state from its result.
__device__ __forceinline__ void oct_lane_exact(const OctParams &o, OctLane &s, int32_t cx, int32_t cy, int32_t &os, int32_t &ot) {
  oct_original(o, true, s.vs, s.vt, cx, cy, os, ot);
  s.vs = os; s.vt = ot;
  oct_lane_state(o, s);
}


// ---- The same step on both components at once (16-bit halves of one register), for octahedra of up to 14 bits.
// With x = I^s(p) (s: p outside the diamond), k = the rotation that takes x to the bottom-left quadrant and ModMax commuting with
// the rotations of the square,  p' = I^s( ModMax( x + R_-k c ) ):  the correction is turned into the frame of x instead of x into
// the canonical frame -- R_-k c = ((odd ? cy : cx) negated if k in {1,2}, (odd ? cx : cy) negated if k in {2,3}), and
//   k in {1,2}  <=>  x > 0 || (x == 0 && y < 0)          k in {2,3}  <=>  y > 0 || (y == 0 && x > 0)        odd  <=>  exactly one of them
// (oct_rotcount).  I(s, t) = (sgn(s) (C - |t|), sgn(t) (C - |s|)) with sgn(0) taken from the other coordinate (oct_invert_diamond);
// a point of the square outside the diamond has no zero coordinate, so the forward I needs no such care, the final one does.
// Valid for |p| <= C componentwise (kept by the step itself) and 0 <= c <= max_q; anything else goes to oct_original.
// pk_*: two int16 lanes of a uint32_t: v_pk_* instructions on the device, plain C on the host (tests/hostcheck).
#if defined(__HIP_DEVICE_COMPILE__)
typedef short pk_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (pk_s16x2)(__builtin_bit_cast(pk_s16x2, a) + __builtin_bit_cast(pk_s16x2, b))); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (pk_s16x2)(__builtin_bit_cast(pk_s16x2, a) - __builtin_bit_cast(pk_s16x2, b))); }
__device__ __forceinline__ uint32_t pk_max_i(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_s16x2, a), __builtin_bit_cast(pk_s16x2, b))); }
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pk_u16x2, a), __builtin_bit_cast(pk_u16x2, b))); }
__device__ __forceinline__ uint32_t pk_sign(uint32_t a) { return __builtin_bit_cast(uint32_t, (pk_s16x2)(__builtin_bit_cast(pk_s16x2, a) >> (pk_s16x2)(15))); }
#else
__host__ __device__ __forceinline__ uint32_t pk_make(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
__host__ __device__ __forceinline__ int32_t pk_lo(uint32_t a) { return (int32_t)(int16_t)(a & 0xFFFFu); }
__host__ __device__ __forceinline__ int32_t pk_hi(uint32_t a) { return (int32_t)(int16_t)(a >> 16); }
__host__ __device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_make(pk_lo(a) + pk_lo(b), pk_hi(a) + pk_hi(b)); }
__host__ __device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_make(pk_lo(a) - pk_lo(b), pk_hi(a) - pk_hi(b)); }
__host__ __device__ __forceinline__ uint32_t pk_max_i(uint32_t a, uint32_t b) { return pk_make(pk_lo(a) > pk_lo(b) ? pk_lo(a) : pk_lo(b), pk_hi(a) > pk_hi(b) ? pk_hi(a) : pk_hi(b)); }
__host__ __device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) { const uint32_t al = a & 0xFFFFu, bl = b & 0xFFFFu, ah = a >> 16, bh = b >> 16; return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16); }
__host__ __device__ __forceinline__ uint32_t pk_sign(uint32_t a) { return pk_make(pk_lo(a) >> 15, pk_hi(a) >> 15); }
#endif
__device__ __forceinline__ uint32_t pk_swap(uint32_t a) { return (a >> 16) | (a << 16); }
__device__ __forceinline__ uint32_t pk_pick(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }   // v_bfi_b32
__device__ __forceinline__ uint32_t pk_abs(uint32_t a) { return pk_max_i(a, pk_sub(0u, a)); }
#define OCT_PK_MAX_BITS 14
// all ones (both halves) when the centred point P lies outside the diamond; A = |P|, SA = A with its halves exchanged
__device__ __forceinline__ uint32_t oct_pk_outside(uint32_t Cv, uint32_t P, uint32_t &A, uint32_t &SA) {
  A = pk_abs(P);
  SA = pk_swap(A);
  return pk_sign(pk_sub(Cv, pk_add(A, SA)));
}
// INV = false: no lane of the wave is outside (OUT == 0 everywhere): the step without either inversion
template <bool INV>
__device__ __forceinline__ uint32_t oct_pk_step(uint32_t Cv, uint32_t Mv, uint32_t P, uint32_t A, uint32_t SA, uint32_t OUT, uint32_t Cc) {
  uint32_t X = P, NG = pk_sign(P);
  if (INV) {
    const uint32_t A1 = pk_pick(OUT, pk_sub(Cv, SA), A);
    X = pk_sub(A1 ^ NG, NG);                        // a magnitude of 0 loses its sign here
    NG = pk_sign(X);
  }
  const uint32_t PS = pk_sign(pk_sub(0u, X));
  // low half: x > 0 || (y < 0 && x >= 0); high half: y > 0 || (x > 0 && y >= 0)
  const uint32_t U = pk_swap(pk_pick(0x0000FFFFu, PS, NG));       // (y < 0, x > 0)
  const uint32_t FL = PS | (U & ~NG);
  const uint32_t ODD = FL ^ pk_swap(FL);
  uint32_t CP = pk_pick(ODD, pk_swap(Cc), Cc);
  CP = pk_sub(CP ^ FL, FL);
  uint32_t t = pk_add(pk_add(X, Cv), CP);           // in (-max_q, 2 max_q): into [0, max_q)
  t = pk_min_u(t, pk_add(t, Mv));
  t = pk_min_u(t, pk_sub(t, Mv));
  const uint32_t W = pk_sub(t, Cv);
  if (!INV) return W;
  const uint32_t IW = pk_sub(Cv, pk_swap(pk_abs(W)));
  const uint32_t SG = pk_sign(pk_add(pk_add(W, W), pk_swap(pk_sign(W))));      // s < 0 || (s == 0 && t < 0)
  return pk_pick(OUT, pk_sub(IW ^ SG, SG), W);
}
// One stream, entry by entry, as a lane of k_predict_oct_streams runs it (the kernel interleaves 64 of these; the host check
// runs this function): packed steps while the stream behaves, the reference's function for an entry that does not.
struct OctPkLane { uint32_t P, Cv, Mv, q; int32_t vs, vt; bool wild; };
__device__ __forceinline__ void oct_pk_init(OctPkLane &s, const OctParams &o, uint32_t q) {
  s.q = q;
  s.Cv = (uint32_t)o.center * 0x00010001u; s.Mv = (uint32_t)o.max_q * 0x00010001u;
  s.P = ((0u - (uint32_t)o.center) & 0xFFFFu) * 0x00010001u;    // the value (0, 0), centred
  s.vs = 0; s.vt = 0; s.wild = false;
}
// the entry by the reference's function, from the lane's state whichever form it has; leaves the packed form behind when the
// result is back inside the square
__device__ __forceinline__ void oct_pk_careful(OctPkLane &s, const OctParams &o, int32_t cx, int32_t cy, int32_t &os, int32_t &ot) {
  if (!s.wild) { s.vs = (int32_t)(int16_t)(s.P & 0xFFFFu) + o.center; s.vt = (int32_t)(int16_t)(s.P >> 16) + o.center; }
  oct_original(o, true, s.vs, s.vt, cx, cy, os, ot);
  s.vs = os; s.vt = ot;
  const int32_t x = (int32_t)((uint32_t)os - (uint32_t)o.center), y = (int32_t)((uint32_t)ot - (uint32_t)o.center);
  s.wild = !(x >= -o.center && x <= o.center && y >= -o.center && y <= o.center);
  if (!s.wild) s.P = ((uint32_t)x & 0xFFFFu) | ((uint32_t)y << 16);
}
__device__ __forceinline__ bool oct_pk_entitled(const OctPkLane &s, int32_t cx, int32_t cy) {
  return !s.wild && ((((uint32_t)cx | (uint32_t)cy) >> s.q) == 0u);
}
__device__ __forceinline__ void oct_pk_entry(OctPkLane &s, const OctParams &o, int32_t cx, int32_t cy, int32_t &os, int32_t &ot) {
  if (!oct_pk_entitled(s, cx, cy)) { oct_pk_careful(s, o, cx, cy, os, ot); return; }
  uint32_t A, SA;
  const uint32_t OUT = oct_pk_outside(s.Cv, s.P, A, SA);
  const uint32_t Cc = ((uint32_t)cx & 0xFFFFu) | ((uint32_t)cy << 16);
  s.P = OUT ? oct_pk_step<true>(s.Cv, s.Mv, s.P, A, SA, OUT, Cc) : oct_pk_step<false>(s.Cv, s.Mv, s.P, A, SA, OUT, Cc);
  const uint32_t v = pk_add(s.P, s.Cv);
  os = (int32_t)(v & 0xFFFFu); ot = (int32_t)(v >> 16);
}

// x / y for y > 0, truncated toward zero as the C# and C++ operators do, without the compiler's 64-bit division routine (a hundred
// vector instructions on the device): for |x| < 2^52 the quotient of the magnitudes from a double division, which is off by at
// most two, then put right by comparing remainders -- exact whatever the rounding was.  Larger dividends take the operator.
__device__ __forceinline__ int64_t div_trunc_pos(int64_t x, int64_t y) {
  const uint64_t ax = x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x, uy = (uint64_t)y;
  if (ax >> 52) return x / y;
  uint64_t q = (uint64_t)((double)ax / (double)uy);
  int64_t r = (int64_t)(ax - q * uy);
  if (r < 0) { --q; r += y; }
  if (r < 0) { --q; r += y; }
  if (r < 0) { --q; r += y; }
  if (r >= y) { ++q; r -= y; }
  if (r >= y) { ++q; r -= y; }
  if (r >= y) { ++q; r -= y; }
  return x < 0 ? -(int64_t)q : (int64_t)q;
}
// GeometricNormal, from the summed face normals n around the vertex to the value (MeshPredictionSchemeGeometricNormalPredictorArea.cs:43-63,
// MeshPredictionSchemeGeometricNormalDecoder.cs:56-69, OctahedronToolBox.cs:28-77,121-137 with the bitstream's 64-bit arithmetic,
// D-9, D-23..D-25): scale into 2^29, project on the octahedron of the transform, flip, canonical (s, t), then the octahedral
// transform's ComputeOriginalValue with the correction (c0, c1).  Shared by the general path and k_predict_geometric.
__device__ __forceinline__ void geometric_normal_finish(const OctParams &o, bool canonical, const uint64_t n[3], bool flip, int32_t c0, int32_t c1,
                                                        int32_t &os, int32_t &ot) {
  const int32_t max_value = o.max_q - 1;
  int64_t nv[3] = {(int64_t)n[0], (int64_t)n[1], (int64_t)n[2]};
  uint64_t as = 0;
  bool sat = false;
  for (int k = 0; k < 3; ++k) {
    const uint64_t x = nv[k] < 0 ? (uint64_t)0 - (uint64_t)nv[k] : (uint64_t)nv[k];
    if (x > 0x7FFFFFFFFFFFFFFFull || as > 0x7FFFFFFFFFFFFFFFull - x) sat = true; else as += x;
  }
  const int64_t abs_sum = sat ? 0x7FFFFFFFFFFFFFFFll : (int64_t)as, upper = (int64_t)1 << 29;
  if (abs_sum > upper) { const int64_t q = abs_sum >> 29; for (int k = 0; k < 3; ++k) nv[k] = div_trunc_pos(nv[k], q); }      // abs_sum / upper
  int32_t v3[3] = {(int32_t)nv[0], (int32_t)nv[1], (int32_t)nv[2]};
  auto abs64 = [](int32_t x) { return x < 0 ? -(int64_t)x : (int64_t)x; };
  const int64_t s3 = abs64(v3[0]) + abs64(v3[1]) + abs64(v3[2]);
  if (s3 == 0) v3[0] = o.center;
  else {
    v3[0] = (int32_t)div_trunc_pos((int64_t)v3[0] * o.center, s3);
    v3[1] = (int32_t)div_trunc_pos((int64_t)v3[1] * o.center, s3);
    const int32_t rest = o.center - (int32_t)abs64(v3[0]) - (int32_t)abs64(v3[1]);
    v3[2] = v3[2] >= 0 ? rest : -rest;
  }
  if (flip) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
  int32_t ps, pt;
  if (v3[0] >= 0) { ps = v3[1] + o.center; pt = v3[2] + o.center; }
  else {
    const int32_t a1 = (int32_t)abs64(v3[1]), a2 = (int32_t)abs64(v3[2]);
    ps = v3[1] < 0 ? a2 : max_value - a2;
    pt = v3[2] < 0 ? a1 : max_value - a1;
  }
  if ((ps == 0 && pt == 0) || (ps == 0 && pt == max_value) || (ps == max_value && pt == 0)) { ps = max_value; pt = max_value; }
  else if (ps == 0 && pt > o.center) pt = o.center - (pt - o.center);
  else if (ps == max_value && pt < o.center) pt = o.center + (o.center - pt);
  else if (pt == max_value && ps < o.center) ps = o.center + (o.center - ps);
  else if (pt == 0 && ps > o.center) ps = o.center - (ps - o.center);
  oct_original(o, canonical, ps, pt, c0, c1, os, ot);
}

__device__ __forceinline__ int32_t wrap_original(int32_t pred, int32_t corr, int32_t mn, int32_t mx, int32_t max_dif) {
  int32_t p = pred > mx ? mx : (pred < mn ? mn : pred);       // PredictionSchemeWrapTransform.cs:67-86
  int32_t o = (int32_t)((uint32_t)p + (uint32_t)corr);       // PredictionSchemeWrapDecodingTransform.cs:46-67
  if (o > mx) o -= max_dif; else if (o < mn) o += max_dif;
  return o;
}

}  // namespace dsa
