// draco-sharp_amd/csrc/dsa_encode.h  (included by dsa_api.hip)
//
// Encode direction of the C-ABI (include/draco_mi355x.h, dsa_encode_*): the drop-in for
//     DracoEncoder.Encode(BinaryWriter, Config, PointCloud, ...)        src/Draco/IO/DracoEncoder.cs:22-41
// for a batch of triangle meshes with per-vertex positions / normals / texture coordinates
// (BASELINE.json configs[4]: "batch quantize + parallelogram predict + rANS encode as HIP kernels").
//
// Split of the work:
//   GPU   (dsa_encode_conn.h)  k_enc_connectivity corner table, Edgebreaker symbols, depth-first attribute order, parallelogram
//                                                 operand entries, one wave per mesh (DSA_ENC_HOST_CONN=1: by the host coder)
//   GPU   (this file)          k_enc_bounds      quantisation range per attribute   AttributeQuantizationTransform.cs:66-108
//                              k_enc_quantize    floats -> portable ints, normals -> octahedral (s,t)   :136-177, OctahedronToolBox.cs:28-119
//                              k_enc_gather      vertex order -> traversal order, wrap bounds           PredictionSchemeWrapTransform.cs:88-100
//                              k_enc_corr        prediction, correction, zig-zag, symbol statistics     MeshPredictionSchemeParallelogramEncoder.cs:35-56,
//                                                                                                       PredictionSchemeWrapEncodingTransform.cs:45-90,
//                                                                                                       ...NormalOctahedronCanonicalizedEncodingTransform.cs:47-83
//                              k_enc_rans        rANS coding of every stream, one wave per stream        RAnsEncoder.cs:22-30, AnsEncoder.cs:34-64, SymbolEncoding.cs:92-193
//                              k_enc_plan        symbol-scheme choice + frequency-table normalisation from the histograms, one lane per
//                                                stream (dsa_symbol_plan.h, the code the host coder runs)   SymbolEncoding.cs:8-40, RAnsSymbolEncoder.cs:15-123
//   host  (dsa_encode_host.h)  input checks; at the end the stream layout: bit-packing of the Edgebreaker symbols, table bytes, section order
//                              (threads over meshes).  DSA_ENC_HOST_PLAN=1: the symbol plans by the host between the two device phases.
// The result is byte-identical to the CPU coder of dsa_encode_host.h (tests/test_gpu_encode.py), hence decodes
// bit-exactly to the quantised input.
#pragma once
#include <chrono>
#include <memory>
#include <thread>

#include "dsa_encode_host.h"
#include "dsa_encode_conn.h"

namespace dsa {

struct EncStream {                 // one per (mesh, attribute); lives in device memory, mirrored on the host
  uint64_t src;                    // f32 source values, vertex order, nc_out per vertex
  uint64_t e2v, ops;               // per mesh: entry -> vertex; i32[3*entries] parallelogram operand entries (next, prev, opposite) or -1
  uint64_t vals, d, syms, bl;      // i32[nv*nc] vertex order, i32[nv*nc] traversal order, u32[nv*nc] symbols, u8[nv] bit length per entry
  uint64_t hist_raw;               // u32[hist_cap]
  uint64_t out_rans, out_bits;     // coded bytes
  uint64_t prob, cum;              // u32[num_symbols] (filled by the host between the two device phases)
  uint32_t nv, nc_out, nc, kind;   // kind 0: quantised + wrap, 1: normals (octahedral, canonicalised delta), 2: uint8 integers + wrap (src: bytes)
  uint32_t bits, prediction, hist_cap, out_cap;
  float qmin[4], qrange;
  int32_t wrap_mn, wrap_mx;
  uint32_t max_value, overflow;
  unsigned long long total_bl;
  uint32_t hist_tag[33];
  uint32_t method, precision_bits, num_symbols;
  uint32_t rans_len, bits_len;
  uint64_t plan_order, plan_tmp;   // u32[table_cap] each: scratch of k_enc_plan
  uint32_t usbl, plan_status;      // raw scheme: unique-symbols bit length; dsa::plan::PLAN_* of k_enc_plan
};

__device__ __forceinline__ uint32_t enc_msb(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
__device__ __forceinline__ uint32_t enc_zigzag(int32_t v) { return v >= 0 ? (uint32_t)v << 1 : (((uint32_t)(-(v + 1))) << 1) | 1u; }

// Quantisation range: per-component min / max, range = largest extent (1 if degenerate).
__global__ __launch_bounds__(256) void k_enc_bounds(uint8_t *arena, EncStream *streams, uint32_t ns) {
  const uint32_t si = blockIdx.x;
  if (si >= ns) return;
  EncStream &S = streams[si];
  if (S.kind != 0) return;
  __shared__ float s_mn[4][256], s_mx[4][256];
  const float *src = (const float *)(arena + S.src);
  const uint32_t nc = S.nc_out, tid = threadIdx.x;
  float mn[4], mx[4];
  for (uint32_t c = 0; c < 4; ++c) { mn[c] = src[c < nc ? c : 0]; mx[c] = mn[c]; }
  for (uint32_t v = tid; v < S.nv; v += 256)
    for (uint32_t c = 0; c < nc; ++c) { const float x = src[(size_t)v * nc + c]; if (x < mn[c]) mn[c] = x; if (x > mx[c]) mx[c] = x; }
  for (uint32_t c = 0; c < 4; ++c) { s_mn[c][tid] = mn[c]; s_mx[c][tid] = mx[c]; }
  __syncthreads();
  for (uint32_t h = 128; h >= 1; h >>= 1) {
    if (tid < h) for (uint32_t c = 0; c < 4; ++c) {
      if (s_mn[c][tid + h] < s_mn[c][tid]) s_mn[c][tid] = s_mn[c][tid + h];
      if (s_mx[c][tid + h] > s_mx[c][tid]) s_mx[c][tid] = s_mx[c][tid + h];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float range = 0.0f;
    for (uint32_t c = 0; c < nc; ++c) { S.qmin[c] = s_mn[c][0]; const float dlt = __fsub_rn(s_mx[c][0], s_mn[c][0]); if (dlt > range) range = dlt; }
    if (range == 0.0f) range = 1.0f;
    S.qrange = range;
  }
}

// OctahedronToolBox.cs:28-119 (float vector -> canonical octahedral coordinates), in double as the host coder
__device__ void enc_oct_from_float(const float *in, int32_t bits, int32_t &s, int32_t &t) {
  const int32_t max_q = (1 << bits) - 1, max_value = max_q - 1, center = max_value / 2;
  const double v0 = in[0], v1 = in[1], v2 = in[2];
  const double abs_sum = __dadd_rn(__dadd_rn(fabs(v0), fabs(v1)), fabs(v2));
  double s0, s1, s2;
  if (abs_sum > 1e-6) { const double sc = __ddiv_rn(1.0, abs_sum); s0 = __dmul_rn(v0, sc); s1 = __dmul_rn(v1, sc); s2 = __dmul_rn(v2, sc); }
  else { s0 = 1; s1 = 0; s2 = 0; }
  int32_t i0 = (int32_t)floor(__dadd_rn(__dmul_rn(s0, (double)center), 0.5));
  int32_t i1 = (int32_t)floor(__dadd_rn(__dmul_rn(s1, (double)center), 0.5));
  int32_t i2 = center - abs(i0) - abs(i1);
  if (i2 < 0) { if (i1 > 0) i1 += i2; else i1 -= i2; i2 = 0; }
  if (s2 < 0) i2 = -i2;
  if (i0 >= 0) { s = i1 + center; t = i2 + center; }
  else {
    s = i1 < 0 ? abs(i2) : max_value - abs(i2);
    t = i2 < 0 ? abs(i1) : max_value - abs(i1);
  }
  // canonicalize
  if ((s == 0 && t == 0) || (s == 0 && t == max_value) || (s == max_value && t == 0)) { s = max_value; t = max_value; }
  else if (s == 0 && t > center) t = center - (t - center);
  else if (s == max_value && t < center) t = center + (center - t);
  else if (t == max_value && s < center) s = center + (center - s);
  else if (t == 0 && s > center) s = center - (s - center);
}

__global__ __launch_bounds__(256) void k_enc_quantize(uint8_t *arena, EncStream *streams, uint32_t ns) {
  const uint32_t si = blockIdx.y;
  if (si >= ns) return;
  const EncStream &S = streams[si];
  const float *src = (const float *)(arena + S.src);
  int32_t *vals = (int32_t *)(arena + S.vals);
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (S.kind == 0) {                // Quantizer: floor((v - min) * (max_q / range) + 0.5), every step rounded to f32
    const float inv_delta = __fdiv_rn((float)(int32_t)((1u << S.bits) - 1u), S.qrange);
    const uint32_t nc = S.nc_out, total = S.nv * nc;
    for (uint32_t i = tid; i < total; i += stride) {
      const float v = __fsub_rn(src[i], S.qmin[i % nc]);
      vals[i] = (int32_t)floorf(__fadd_rn(__fmul_rn(v, inv_delta), 0.5f));
    }
  } else if (S.kind == 2) {         // an integer attribute: its values as they are (SequentialIntegerAttributeEncoder.cs: no transform)
    const uint8_t *srcb = arena + S.src;
    const uint32_t total = S.nv * S.nc;
    for (uint32_t i = tid; i < total; i += stride) vals[i] = (int32_t)srcb[i];
  } else {
    for (uint32_t v = tid; v < S.nv; v += stride) {
      int32_t s, t;
      enc_oct_from_float(src + (size_t)v * 3, (int32_t)S.bits, s, t);
      vals[2 * v] = s; vals[2 * v + 1] = t;
    }
  }
}

// traversal order + bounds of the wrap transform over all values of the attribute
__global__ __launch_bounds__(256) void k_enc_gather(uint8_t *arena, EncStream *streams, uint32_t ns) {
  const uint32_t si = blockIdx.x;
  if (si >= ns) return;
  EncStream &S = streams[si];
  __shared__ int32_t s_mn[256], s_mx[256];
  const int32_t *vals = (const int32_t *)(arena + S.vals);
  int32_t *d = (int32_t *)(arena + S.d);
  const uint32_t *e2v = (const uint32_t *)(arena + S.e2v);
  const uint32_t nc = S.nc, tid = threadIdx.x;
  int32_t mn = 0x7FFFFFFF, mx = (int32_t)0x80000000;
  for (uint32_t e = tid; e < S.nv; e += 256) {
    const uint32_t v = e2v[e];
    for (uint32_t c = 0; c < nc; ++c) { const int32_t x = vals[(size_t)v * nc + c]; d[(size_t)e * nc + c] = x; if (x < mn) mn = x; if (x > mx) mx = x; }
  }
  s_mn[tid] = mn; s_mx[tid] = mx;
  __syncthreads();
  for (uint32_t h = 128; h >= 1; h >>= 1) {
    if (tid < h) { if (s_mn[tid + h] < s_mn[tid]) s_mn[tid] = s_mn[tid + h]; if (s_mx[tid + h] > s_mx[tid]) s_mx[tid] = s_mx[tid + h]; }
    __syncthreads();
  }
  if (tid == 0) { S.wrap_mn = s_mn[0]; S.wrap_mx = s_mx[0]; }
}

// corrections -> symbols, per-entry bit lengths, statistics
__global__ __launch_bounds__(256) void k_enc_corr(uint8_t *arena, EncStream *streams, uint32_t ns) {
  const uint32_t si = blockIdx.y;
  if (si >= ns) return;
  EncStream &S = streams[si];
  __shared__ uint32_t s_tag[33];
  __shared__ uint32_t s_max;
  __shared__ unsigned long long s_bl;
  // the block counts its symbols in LDS and adds what it counted to the stream's histogram once (alphabets of up to 12 bits; a
  // global atomic per symbol was a third of the attribute kernels' time)
  __shared__ uint32_t s_hist[4098];
  const bool lds_hist = S.hist_cap <= 4098u;
  if (lds_hist) for (uint32_t i = threadIdx.x; i < S.hist_cap; i += blockDim.x) s_hist[i] = 0;
  if (threadIdx.x < 33) s_tag[threadIdx.x] = 0;
  if (threadIdx.x == 0) { s_max = 0; s_bl = 0; }
  __syncthreads();
  const int32_t *d = (const int32_t *)(arena + S.d);
  const int32_t *ops = (const int32_t *)(arena + S.ops);
  uint32_t *syms = (uint32_t *)(arena + S.syms);
  uint8_t *bl = arena + S.bl;
  uint32_t *hist = (uint32_t *)(arena + S.hist_raw);
  const uint32_t nc = S.nc;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  // wrap transform, PredictionSchemeWrapEncodingTransform.cs:45-90 (E-5) + WrapTransform.cs:88-100
  const int32_t mn = S.wrap_mn, mx = S.wrap_mx, max_dif = 1 + mx - mn;
  int32_t max_corr = max_dif / 2;
  const int32_t min_corr = -max_corr;
  if ((max_dif & 1) == 0) max_corr -= 1;
  // octahedron
  const int32_t o_max_q = (1 << S.bits) - 1, o_center = (o_max_q - 1) / 2;
  for (uint32_t p = tid; p < S.nv; p += stride) {
    uint32_t mc = 0;
    if (S.kind != 1) {
      int32_t vn = -1, vp = -1, vo = -1;
      if (S.prediction == 1 && p > 0) { vn = ops[3 * p]; vp = ops[3 * p + 1]; vo = ops[3 * p + 2]; }
      for (uint32_t c = 0; c < nc; ++c) {
        int32_t pred;
        if (vn >= 0) pred = d[(size_t)vn * nc + c] + d[(size_t)vp * nc + c] - d[(size_t)vo * nc + c];
        else pred = p > 0 ? d[(size_t)(p - 1) * nc + c] : 0;
        const int32_t pc = pred > mx ? mx : (pred < mn ? mn : pred);
        int32_t cr = d[(size_t)p * nc + c] - pc;
        if (cr < min_corr) cr += max_dif; else if (cr > max_corr) cr -= max_dif;
        const uint32_t sy = enc_zigzag(cr);
        syms[(size_t)p * nc + c] = sy;
        if (sy < S.hist_cap) atomicAdd(lds_hist ? &s_hist[sy] : &hist[sy], 1u); else S.overflow = 1;
        mc = sy > mc ? sy : mc;
      }
    } else {                        // PredictionSchemeNormalOctahedronCanonicalizedEncodingTransform.cs:47-83
      int32_t os = d[2 * p] - o_center, ot = d[2 * p + 1] - o_center;
      int32_t ps = (p > 0 ? d[2 * (p - 1)] : 0) - o_center, pt = (p > 0 ? d[2 * (p - 1) + 1] : 0) - o_center;
      const int32_t aps = ps < 0 ? -ps : ps, apt = pt < 0 ? -pt : pt;
      if (!((uint32_t)aps + (uint32_t)apt <= (uint32_t)o_center)) { oct_invert_diamond(o_center, os, ot); oct_invert_diamond(o_center, ps, pt); }
      const bool bottom_left = (ps == 0 && pt == 0) || (ps < 0 && pt <= 0);
      if (!bottom_left) {
        int rot;
        if (ps == 0) rot = pt == 0 ? 0 : (pt > 0 ? 3 : 1);
        else if (ps > 0) rot = pt >= 0 ? 2 : 1;
        else rot = pt <= 0 ? 0 : 3;
        oct_rotate(os, ot, rot); oct_rotate(ps, pt, rot);
      }
      int32_t c0 = os - ps, c1 = ot - pt;
      if (c0 < 0) c0 += o_max_q;
      if (c1 < 0) c1 += o_max_q;
      const uint32_t sy[2] = {(uint32_t)c0, (uint32_t)c1};     // positive: no zig-zag
      for (uint32_t c = 0; c < 2; ++c) {
        syms[2 * p + c] = sy[c];
        if (sy[c] < S.hist_cap) atomicAdd(lds_hist ? &s_hist[sy[c]] : &hist[sy[c]], 1u); else S.overflow = 1;
        mc = sy[c] > mc ? sy[c] : mc;
      }
    }
    const uint32_t b = (mc > 0 ? enc_msb(mc) : 0u) + 1u;
    bl[p] = (uint8_t)b;
    atomicAdd(&s_tag[b], 1u);
    atomicMax(&s_max, mc);
    atomicAdd(&s_bl, (unsigned long long)b);
  }
  __syncthreads();
  if (lds_hist) for (uint32_t i = threadIdx.x; i < S.hist_cap; i += blockDim.x) { const uint32_t c = s_hist[i]; if (c) atomicAdd(&hist[i], c); }
  if (threadIdx.x < 33 && s_tag[threadIdx.x]) atomicAdd(&S.hist_tag[threadIdx.x], s_tag[threadIdx.x]);
  if (threadIdx.x == 0) { atomicMax(&S.max_value, s_max); atomicAdd(&S.total_bl, s_bl); }
}

// Symbol-scheme choice and frequency-table normalisation, one lane per stream: dsa_symbol_plan.h, the code the host coder
// runs, on the histograms k_enc_corr left in device memory.  Sequential per stream (a stable sort and a fix-up loop over the
// alphabet), thousands of streams side by side; its tables go straight to k_enc_rans, no host in between.
__global__ __launch_bounds__(WAVE) void k_enc_plan(uint8_t *arena, EncStream *streams, uint32_t ns, int force_scheme, int compression_level) {
  const uint32_t si = blockIdx.x * WAVE + threadIdx.x;
  if (si >= ns) return;
  EncStream &S = streams[si];
  if (S.overflow) return;
  if (S.max_value >= S.hist_cap) { S.overflow = 1; return; }
  const uint32_t *raw = (const uint32_t *)(arena + S.hist_raw);
  uint32_t *prob = (uint32_t *)(arena + S.prob), *cum = (uint32_t *)(arena + S.cum);
  uint32_t *order = (uint32_t *)(arena + S.plan_order), *tmp = (uint32_t *)(arena + S.plan_tmp);
  int method = 1, usbl = 0;
  int rc = plan::choose_scheme((const uint32_t *)S.hist_tag, raw, S.max_value, (uint64_t)S.nv * S.nc, S.nc, (uint64_t)S.total_bl, force_scheme, compression_level, &method, &usbl);
  int pb = 12;
  uint32_t nsym = 0;
  if (rc == plan::PLAN_OK)
    rc = method == 0 ? plan::rans_tables(5, (const uint32_t *)S.hist_tag, (size_t)33, prob, cum, order, tmp, &pb, &nsym)
                     : plan::rans_tables(usbl, raw, (size_t)S.max_value + 1, prob, cum, order, tmp, &pb, &nsym);
  S.plan_status = (uint32_t)rc;
  if (rc != plan::PLAN_OK) { S.overflow = 1; return; }
  S.method = (uint32_t)method; S.usbl = (uint32_t)usbl; S.precision_bits = (uint32_t)pb; S.num_symbols = nsym;
}

// rANS coding, one WAVE per stream.  The coder state is a serial chain (x' = (x / p) << bits + x % p + cum after the renormalisation
// bytes), so it lives in scalar registers; what the wave does in parallel is everything off the chain: 64 symbols at a time are
// looked up (frequency, cumulative count, and the reciprocal that turns the division into a multiply-high with one correction),
// and the bytes collect in a register, a lane each, and leave 64 at a time.
// Symbols are fed last -> first (SymbolEncoding.cs:177-183); bytes are written in coding order, the decoder reads
// them from the end (RAnsEncoder.cs:22-30, AnsEncoder.cs:34-64).
__global__ __launch_bounds__(WAVE) void k_enc_rans(uint8_t *arena, EncStream *streams, uint32_t ns) {
  const uint32_t si = blockIdx.x, lane = threadIdx.x;
  if (si >= ns) return;
  EncStream &S = streams[si];
  if (S.overflow) return;
  const uint32_t *prob = (const uint32_t *)(arena + S.prob), *cum = (const uint32_t *)(arena + S.cum);
  const uint32_t *syms = (const uint32_t *)(arena + S.syms);
  const uint8_t *bl = arena + S.bl;
  uint8_t *out = arena + S.out_rans;
  const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.precision_bits), precision = 1u << pb, l_base = precision * 4u;
  const bool tagged = S.method == 0;
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tagged ? S.nv : S.nv * S.nc));
  const uint32_t cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.out_cap);
  uint32_t state = l_base, len = 0;                 // wave-uniform
  uint32_t held = 0, bytes = 0;                     // bytes of lane j of `bytes`: out[len - held + j], j < held
  auto emit = [&](uint32_t byte) {
    bytes = lane == held ? byte : bytes;
    ++held; ++len;
    if (held == WAVE) {
      const uint32_t at = len - WAVE + lane;
      if (at < cap) out[at] = (uint8_t)bytes;
      held = 0;
    }
  };
  for (uint32_t hi = n; hi > 0;) {
    const uint32_t cnt = hi < WAVE ? hi : WAVE;
    // lane j holds the j-th symbol of this stretch in coding order
    uint32_t p = 1, c = 0, magic = 0;
    if (lane < cnt) {
      const uint32_t k = hi - 1 - lane;
      const uint32_t sym = tagged ? (uint32_t)bl[k] : syms[k];
      p = prob[sym]; c = cum[sym];
      if (p == 0) p = 1;                              // (a symbol that occurs has a frequency; a zero here must not spin the loop below)
      magic = p > 1 ? 0xFFFFFFFFu / p + 1u : 0u;      // ceil(2^32 / p)
    }
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)j), cj = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)j);
      const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)magic, (int)j);
      const uint32_t lim = pj << 10;                // (l_base / precision) * 256 * p; p <= precision <= 2^20
      while (state >= lim) { emit(state & 0xFFu); state >>= 8; }
      // state / p: the multiply-high by ceil(2^32 / p) is the quotient or one more (state < 2^32, error < state / 2^32 < 1)
      uint32_t q = pj > 1 ? __umulhi(state, mj) : state;
      uint32_t r = state - q * pj;
      if ((int32_t)r < 0) { --q; r += pj; }
      state = (q << pb) + r + cj;
    }
    hi -= cnt;
  }
  const uint32_t fs = state - l_base;
  uint32_t v, nb;
  if (fs < (1u << 6)) { v = fs; nb = 1; }
  else if (fs < (1u << 14)) { v = 0x4000u + fs; nb = 2; }
  else if (fs < (1u << 22)) { v = 0x800000u + fs; nb = 3; }
  else { v = 0xC0000000u + fs; nb = 4; }
  for (uint32_t i = 0; i < nb; ++i) emit((v >> (8 * i)) & 0xFFu);
  if (lane < held) { const uint32_t at = len - held + lane; if (at < cap) out[at] = (uint8_t)bytes; }
  if (lane != 0) return;
  S.rans_len = len;
  if (len > cap || fs >= (1u << 30)) S.overflow = 1;
  // tagged scheme: the values follow as raw LSB-first bit fields of their entry's length (SymbolEncoding.cs:117-137)
  uint32_t blen = 0;
  if (tagged) {
    uint8_t *bits = arena + S.out_bits;
    uint64_t acc = 0;
    uint32_t nacc = 0;
    for (uint32_t e = 0; e < S.nv; ++e) {
      const uint32_t b = bl[e];
      for (uint32_t c = 0; c < S.nc; ++c) {
        const uint64_t val = (uint64_t)syms[(size_t)e * S.nc + c] & (b >= 32 ? 0xFFFFFFFFull : ((1ull << b) - 1ull));
        acc |= val << nacc;
        nacc += b;
        while (nacc >= 8) { if (blen < cap) bits[blen] = (uint8_t)(acc & 0xFF); ++blen; acc >>= 8; nacc -= 8; }
      }
    }
    if (nacc > 0) { if (blen < cap) bits[blen] = (uint8_t)(acc & 0xFF); ++blen; }
    if (blen > cap) S.overflow = 1;
  }
  S.bits_len = blen;
}

}  // namespace dsa

namespace dsa {
// Many small pieces in one transfer: k_enc_pack gathers pieces of the arena into one buffer (then one copy to the host),
// k_enc_unpack scatters one uploaded buffer into the arena.  A batch has thousands of such pieces (symbols and split events per
// mesh, histograms, tables and coded bytes per stream); as copies of their own they cost more than the kernels between them.
struct PackItem { uint64_t arena_off, packed_off; uint32_t len, pad; };
__global__ __launch_bounds__(256) void k_enc_pack(const uint8_t *arena, uint8_t *packed, const PackItem *items, uint32_t n) {
  if (blockIdx.x >= n) return;
  const PackItem it = items[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < it.len; i += 256) packed[it.packed_off + i] = arena[it.arena_off + i];
}
__global__ __launch_bounds__(256) void k_enc_unpack(uint8_t *arena, const uint8_t *packed, const PackItem *items, uint32_t n) {
  if (blockIdx.x >= n) return;
  const PackItem it = items[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < it.len; i += 256) arena[it.arena_off + i] = packed[it.packed_off + i];
}

}  // namespace dsa

// ------------------------------------------------------------------------------------------------ host side
struct dsa_encoded {
  dsa_context *ctx = nullptr;
  std::vector<std::vector<uint8_t>> streams;
  std::vector<int32_t> status;
  std::vector<std::string> messages;
};

extern "C" {

void dsa_encode_default_options(dsa_encode_options *o) {
  if (!o) return;
  synth::Options d;
  o->position_bits = d.pos_bits; o->texcoord_bits = d.uv_bits; o->normal_bits = d.normal_bits;
  o->single_connectivity = d.single_connectivity; o->symbol_scheme = d.force_scheme; o->compression_level = d.compression_level;
  o->position_prediction = d.pos_prediction; o->texcoord_prediction = d.uv_prediction;
}

static dsa_status encode_chunk(dsa_context *ctx, EncLane &lane, uint32_t n, uint32_t batch_n, const dsa_mesh_input *meshes, const dsa_encode_options *options, dsa_encoded **out);
static dsa_status encode_batch(dsa_context *ctx, uint32_t n, const dsa_mesh_input *meshes, const dsa_encode_options *options, dsa_encoded **out);
dsa_status dsa_encode_batch(dsa_context *ctx, uint32_t n, const dsa_mesh_input *meshes, const dsa_encode_options *options, dsa_encoded **out) {
  if (!ctx || !out || (n && !meshes)) return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "null argument");
  DSA_GUARD(ctx, encode_batch(ctx, n, meshes, options, out));     // host vectors and threads inside: nothing may unwind into the caller
}
// A batch is coded in chunks, several of them in flight (each on a lane of its own: stream + pinned staging + device memory).  The
// device stages of a chunk are bound by latency -- the walks of k_enc_connectivity take a memory round trip per step, 0.1 - 0.2 s
// whatever the number of meshes -- so the more chunks are under way the better: the uploads of the chunks go over the link one
// after the other (a turn each: the first chunk's kernels start after its own upload, not after everybody's), the kernels of one
// run beside the walks of the others and beside the host's stream layout of those that are done.  The walks need the faces only:
// those go first (phase A of every chunk in front of any phase B, hostutil::UploadTurns), the attribute values follow while the
// walks run, on a stream of their own.  Streams of one priority share four hardware queues, on which the kernels of different
// streams wait for each other: four lanes, their walk streams at another priority.  Small batches are one chunk.
static dsa_status encode_batch(dsa_context *ctx, uint32_t n, const dsa_mesh_input *meshes, const dsa_encode_options *options, dsa_encoded **out) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint32_t max_lanes = [&]() { const char *e = getenv("DSA_ENC_LANES"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= 16 ? v : 4); }();
  const uint32_t chunk_max = [&]() { const char *e = getenv("DSA_ENC_CHUNK"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 ? v : 0); }();
  // as many chunks as lanes, of 512 to 1024 meshes (a chunk's device memory: about 9 MB per 64k-triangle mesh)
  const uint32_t chunk = chunk_max ? std::min(std::max(n, 1u), chunk_max) : (n <= 512 ? std::max(n, 1u) : std::min(1024u, std::max(512u, (n + max_lanes - 1) / max_lanes)));
  const uint32_t chunks = (n + chunk - 1) / chunk, lanes = std::max(1u, std::min(chunks, max_lanes));
  // (Tapering the last chunks -- the batch is over when its last chunk is, and what shrinks with a chunk is everything around its
  // walks -- was measured twice and did not beat equal chunks for 4096 meshes (470 - 490 against 410 ms while the stream layout was
  // slow, 376 - 426 against 370 - 383 since); it helps 8192 meshes (605 against 680 ms) and costs 2048 (296 against 235).)
  std::vector<uint32_t> bounds(chunks + 1, 0);
  for (uint32_t c = 0; c <= chunks; ++c) bounds[c] = (uint32_t)std::min<uint64_t>(n, (uint64_t)c * chunk);
  while (ctx->enc_lanes.size() < lanes) {
    std::unique_ptr<EncLane> l(new EncLane());
    l->device = ctx->device;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    HIP_TRY(ctx, hipStreamCreateWithFlags(&l->st, hipStreamNonBlocking));
    // (a priority has its own few hardware queues: the walk streams are dealt over two, so that four walks run side by side)
    HIP_TRY(ctx, hipStreamCreateWithPriority(&l->walk_st, hipStreamNonBlocking, (ctx->enc_lanes.size() & 1) ? greatest : least));
    HIP_TRY(ctx, hipEventCreateWithFlags(&l->tables_done, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&l->walk_done, hipEventDisableTiming));
    ctx->enc_lanes.push_back(std::move(l));
  }
  hostutil::UploadTurns upload_turn(chunks);
  for (uint32_t l = 0; l < lanes; ++l) ctx->enc_lanes[l]->upload_turn = &upload_turn;
  struct Unhook { dsa_context *c; ~Unhook() { for (auto &l : c->enc_lanes) l->upload_turn = nullptr; } } unhook{ctx};
  std::unique_ptr<dsa_encoded> E(new dsa_encoded());
  E->ctx = ctx;
  E->streams.resize(n); E->status.assign(n, DSA_OK); E->messages.resize(n);
  std::atomic<uint32_t> next{0};
  std::atomic<int> failed{DSA_OK};
  std::vector<std::string> errs(lanes);
  auto work = [&](uint32_t l) noexcept {
    try {
      dsa_context sink;                              // receives this lane's error text (set_err writes ctx->err: not from several threads)
      sink.device = ctx->device;
      if (hipSetDevice(ctx->device) != hipSuccess) { int ok = DSA_OK; failed.compare_exchange_strong(ok, DSA_ERR_DEVICE); return; }
      for (;;) {
        const uint32_t c = next.fetch_add(1, std::memory_order_relaxed);
        if (c >= chunks) break;
        if (failed.load(std::memory_order_relaxed) != DSA_OK) { upload_turn.finish_a(c, false); break; }     // (a chunk given up is not one the others' uploads wait for)
        const uint32_t base = bounds[c], cnt = bounds[c + 1] - base;
        if (cnt == 0) { upload_turn.finish_a(c, false); continue; }
        ctx->enc_lanes[l]->upload_chunk = c;
        dsa_encoded *part = nullptr;
        const auto t_chunk = std::chrono::steady_clock::now();
        const dsa_status st = encode_chunk(&sink, *ctx->enc_lanes[l], cnt, n, meshes + base, options, &part);
        if (getenv("DSA_ENC_TIMING")) fprintf(stderr, "[dsa_encode_batch] chunk %u (%u meshes) returned after %8.2f ms\n", c, cnt, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_chunk).count());
        if (st != DSA_OK) { errs[l] = sink.err; int ok = DSA_OK; failed.compare_exchange_strong(ok, st); break; }
        std::unique_ptr<dsa_encoded> owner(part);
        for (uint32_t i = 0; i < cnt; ++i) { E->streams[base + i].swap(part->streams[i]); E->status[base + i] = part->status[i]; E->messages[base + i].swap(part->messages[i]); }
      }
    } catch (const std::bad_alloc &) { int ok = DSA_OK; failed.compare_exchange_strong(ok, DSA_ERR_OUT_OF_MEMORY); }
    catch (...) { int ok = DSA_OK; failed.compare_exchange_strong(ok, DSA_ERR_DEVICE); }
  };
  {
    struct Joiner { std::vector<std::thread> t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner;
    try { joiner.t.reserve(lanes); for (uint32_t l = 1; l < lanes; ++l) joiner.t.emplace_back(work, l); } catch (...) {}
    work(0);
  }
  if (failed.load() != DSA_OK) {
    for (const std::string &e : errs) if (!e.empty()) return set_err(ctx, (dsa_status)failed.load(), "%s", e.c_str());
    return set_err(ctx, (dsa_status)failed.load(), "encoding failed");
  }
  if (getenv("DSA_ENC_TIMING")) fprintf(stderr, "[dsa_encode_batch] batch of %u done\n", n);
  *out = E.release();
  return DSA_OK;
}
static dsa_status encode_chunk(dsa_context *ctx, EncLane &lane, uint32_t n, uint32_t batch_n, const dsa_mesh_input *meshes, const dsa_encode_options *options, dsa_encoded **out) {
  hostutil::TurnGuard turn(lane.upload_turn, lane.upload_chunk);      // (whatever happens below, the other chunks' uploads do not wait for this one's)
  HIP_TRY(ctx, hipSetDevice(lane.device));
  dsa_encode_options od;
  dsa_encode_default_options(&od);
  if (options) od = *options;
  if (od.position_bits < 1 || od.position_bits > 20 || od.texcoord_bits < 1 || od.texcoord_bits > 20 || od.normal_bits < 2 || od.normal_bits > 20)
    return set_err(ctx, DSA_ERR_INVALID_ARGUMENT, "quantisation bits out of range (positions/texcoords 1..20, normals 2..20)");
  synth::Options opt;
  opt.pos_bits = od.position_bits; opt.uv_bits = od.texcoord_bits; opt.normal_bits = od.normal_bits;
  opt.single_connectivity = od.single_connectivity; opt.force_scheme = od.symbol_scheme; opt.compression_level = od.compression_level;
  opt.pos_prediction = od.position_prediction; opt.uv_prediction = od.texcoord_prediction;
  dsa_encoded *E = new (std::nothrow) dsa_encoded();
  if (!E) return set_err(ctx, DSA_ERR_OUT_OF_MEMORY, "host allocation failed");
  std::unique_ptr<dsa_encoded> E_owner(E);      // released into *out at the very end; every other exit frees it
  E->ctx = ctx;
  E->streams.resize(n); E->status.assign(n, DSA_OK); E->messages.resize(n);
  // ---- host phase 1: connectivity, traversal order, operand entries (threads over meshes)
  std::vector<synth::MeshPlan> plans(n);
  std::vector<synth::MeshIn> ins(n);
  std::vector<std::vector<uint32_t>> e2v(n);
  std::vector<std::vector<int32_t>> ops(n);
  // DSA_ENC_TIMING=1 (diagnostics): wall time of every phase on stderr
  static const bool timing = getenv("DSA_ENC_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    static const auto t_zero = now;
    fprintf(stderr, "[dsa_encode_batch] lane %p at %9.2f ms: %-28s %8.2f ms\n", (void *)&lane, std::chrono::duration<double, std::milli>(now - t_zero).count(), what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // Both device stages are the serial algorithms on one lane per mesh / per stream: a batch takes 150 - 250 ms (connectivity) and
  // about 40 ms (plans) whatever its size, which the host threads beat on a small batch (measured: 64 meshes 84 ms on the host, 128 meshes 252 ms on the device).  Below 256 meshes the host does both
  // (DSA_ENC_HOST_CONN / DSA_ENC_HOST_PLAN = 1 or 0 force one or the other; read per call: the tests compare the paths).
  auto choice = [&](const char *name) { const char *e = getenv(name); return e ? atoi(e) != 0 : batch_n < 256; };
  const bool host_conn = choice("DSA_ENC_HOST_CONN");
  // meshes to a wave of the walks (k_enc_connectivity: one lane per mesh); DSA_ENC_WALK_LANES = 1 .. 64 for measurements
  const uint32_t walk_lanes = [&]() { const char *e = getenv("DSA_ENC_WALK_LANES"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= 64 ? v : 16); }();
  const bool host_plan = choice("DSA_ENC_HOST_PLAN");
  auto plan_one = [&](uint32_t i) {
    const dsa_mesh_input &m = meshes[i];
    synth::MeshIn &in = ins[i];
    in.pos = m.positions; in.nv = m.num_vertices; in.faces = m.faces; in.nf = m.num_faces; in.normals = m.normals; in.uvs = m.texcoords;
    in.generic = (m.generic && m.generic_components >= 1 && m.generic_components <= 4) ? m.generic : nullptr;
    try {
      synth::check(m.positions && m.faces && m.num_vertices >= 3 && m.num_faces >= 1, "mesh needs positions and faces");
      for (size_t k = 0; k < (size_t)m.num_faces * 3; ++k) synth::check(m.faces[k] < m.num_vertices, "face index out of range");
      synth::check(host_conn || (uint64_t)m.num_faces * 3 <= (uint64_t)dsa::EC_CORNER_MASK, "mesh too large for the device connectivity coder");
      synth::Options mo = opt;                                   // (the components of the generic attribute are the mesh's own)
      mo.generic_components = in.generic ? (int32_t)m.generic_components : 1;
      if (!host_conn) { synth::plan_attributes(in, mo, plans[i]); return; }       // the rest of the plan comes from the device
      synth::plan_mesh(in, mo, plans[i]);
      const synth::MeshPlan &pl = plans[i];
      const uint32_t V = m.num_vertices;
      e2v[i].resize(V);
      ops[i].assign((size_t)3 * V, -1);
      for (uint32_t p = 0; p < V; ++p) {
        const uint32_t ci = pl.seq.data_to_corner[p];
        e2v[i][p] = pl.ct.vertex(ci);
        if (p == 0) continue;
        const uint32_t oci = pl.ct.opposite(ci);
        if (oci == synth::kInvalid) continue;
        const int32_t vo = pl.seq.vertex_to_data[pl.ct.vertex(oci)];
        const int32_t vn = pl.seq.vertex_to_data[pl.ct.vertex(synth::CornerTable::next(oci))];
        const int32_t vp = pl.seq.vertex_to_data[pl.ct.vertex(synth::CornerTable::prev(oci))];
        if (vo < (int32_t)p && vn < (int32_t)p && vp < (int32_t)p) { ops[i][3 * p] = vn; ops[i][3 * p + 1] = vp; ops[i][3 * p + 2] = vo; }
      }
    } catch (const std::exception &e) { E->status[i] = DSA_ERR_INVALID_DATA; E->messages[i] = e.what(); }
  };
  hostutil::parallel_for(n, plan_one);          // capped thread count, every thread joined on every path (dsa_host_util.h)
  lap("host checks / plan");
  // ---- device layout
  std::vector<dsa::EncStream> hs;
  std::vector<dsa::EncConn> hc(host_conn ? 0 : n);
  std::vector<uint32_t> first_stream(n + 1, 0);
  // What the host provides (faces, raw attribute values; with host connectivity the traversal order and operands) lies at the
  // front of the arena in one run, so that it travels in a few large transfers out of pinned staging; everything else behind it.
  auto al = [](uint64_t b) { return (b + 255) & ~255ull; };
  uint64_t in_total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (E->status[i] != DSA_OK) continue;
    const uint64_t V = meshes[i].num_vertices, F = meshes[i].num_faces;
    in_total += host_conn ? al(4 * V) + al(12 * V) : al((V <= 65536 ? 6 : 12) * F);      // (faces of a mesh of up to 65 536 vertices travel as 16-bit indices)
    for (auto &a : plans[i].atts) in_total += al(4 * V * (uint64_t)a.nc_out);
  }
  uint64_t cur = in_total, cur_in = 0;
  auto take = [&](uint64_t bytes) { uint64_t at = cur; cur = (cur + bytes + 255) & ~255ull; return at; };
  auto take_in = [&](uint64_t bytes) { uint64_t at = cur_in; cur_in = (cur_in + bytes + 255) & ~255ull; return at; };
  struct Upload { uint64_t off; const void *src; size_t bytes; bool narrow; };       // narrow: src is u32[bytes / 2], the staging copy keeps the low halves
  std::vector<Upload> uploads_a, uploads;          // phase A: what the walks need (the faces); the rest
  std::vector<uint64_t> faces_at(n, 0);
  if (!host_conn)
    for (uint32_t i = 0; i < n; ++i) {
      if (E->status[i] != DSA_OK) continue;
      // Half of what the walks wait for is the upload of the faces: indices below 65 536 are narrowed to 16 bits by the copy into
      // pinned staging (which reads them anyway) and widened by the first kernel of the chunk.
      const bool narrow = meshes[i].num_vertices <= 65536;
      faces_at[i] = take_in((narrow ? 6ull : 12ull) * meshes[i].num_faces);
      uploads_a.push_back({faces_at[i], meshes[i].faces, (narrow ? 6ull : 12ull) * meshes[i].num_faces, narrow});
    }
  for (uint32_t i = 0; i < n; ++i) {
    first_stream[i] = (uint32_t)hs.size();
    if (E->status[i] != DSA_OK) continue;
    const uint32_t V = meshes[i].num_vertices;
    const uint64_t o_e2v = host_conn ? take_in(4ull * V) : take(4ull * V), o_ops = host_conn ? take_in(12ull * V) : take(12ull * V);
    if (host_conn) {
      uploads.push_back({o_e2v, e2v[i].data(), 4ull * V, false});
      uploads.push_back({o_ops, ops[i].data(), 12ull * V, false});
    } else {
      const uint32_t F = meshes[i].num_faces;
      dsa::EncConn &C = hc[i];
      memset(&C, 0, sizeof(C));
      C.F = F; C.V = V; C.split_cap = F; C.fail_key = 0xFFFFFFFFu;
      if (V <= 65536) { C.faces_narrow = 1; C.faces16 = faces_at[i]; C.faces = take(12ull * F); } else C.faces = faces_at[i];
      C.opp = take(12ull * F); C.voff = take(4ull * (V + 1)); C.vcur = take(4ull * V); C.vlist = take(12ull * F); C.vcorner = take(4ull * V);
      C.vvis = take(V); C.frec = take(32ull * F);
      C.stack = take(4ull * F); C.processed = take(4ull * F); C.init_corners = take(4ull * F);
      C.symbols = take(F); C.start_bits = take(F); C.splits = take(12ull * C.split_cap);
      C.d2c = take(4ull * V); C.v2d = take(4ull * V);
      C.e2v = o_e2v; C.ops = o_ops;
    }
    for (auto &a : plans[i].atts) {
      dsa::EncStream S;
      memset(&S, 0, sizeof(S));
      const bool integer = a.att_type == 4;                  // the generic uint8 attribute
      const void *src = a.att_type == 0 ? (const void *)meshes[i].positions : (a.att_type == 1 ? (const void *)meshes[i].normals : (integer ? (const void *)meshes[i].generic : (const void *)meshes[i].texcoords));
      S.nv = V; S.nc_out = (uint32_t)a.nc_out; S.nc = (uint32_t)a.nc; S.kind = a.seq_type == 3 ? 1u : (integer ? 2u : 0u);
      S.bits = integer ? 9u : (uint32_t)a.bits; S.prediction = (uint32_t)a.prediction;      // (9: the zig-zagged corrections of bytes are below 512)
      const uint64_t src_bytes = (integer ? 1ull : 4ull) * V * S.nc_out;
      S.src = take_in(src_bytes);
      uploads.push_back({S.src, src, src_bytes, false});
      S.e2v = o_e2v; S.ops = o_ops;
      S.vals = take(4ull * V * S.nc); S.d = take(4ull * V * S.nc); S.syms = take(4ull * V * S.nc); S.bl = take(V);
      S.hist_cap = (1u << S.bits) + 2u;                      // zig-zag of a wrapped correction / a positive octahedral correction fits
      S.hist_raw = take(4ull * S.hist_cap);
      S.out_cap = 4u * V * S.nc + 16u;
      S.out_rans = take(S.out_cap); S.out_bits = take(S.out_cap);
      const uint64_t table_cap = std::max<uint64_t>(S.hist_cap, 64);   // the tagged scheme's alphabet is 33 bit lengths
      S.prob = take(4ull * table_cap); S.cum = take(4ull * table_cap);
      S.plan_order = take(4ull * table_cap); S.plan_tmp = take(4ull * table_cap);
      hs.push_back(S);
    }
  }
  first_stream[n] = (uint32_t)hs.size();
  const uint32_t ns = (uint32_t)hs.size();
  uint8_t *arena = nullptr;
  dsa::EncStream *d_streams = nullptr;
  dsa::EncConn *d_conns = nullptr;
  auto cleanup = [&]() {};      // the lane owns its device memory (EncLane::Buf): nothing to release per chunk
#define ENC_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { return set_err(ctx, e_ == hipErrorOutOfMemory ? DSA_ERR_OUT_OF_MEMORY : DSA_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
  // pieces of the arena -> one host buffer (items[k].packed_off filled in); host buffer -> pieces of the arena
  // (with `view`: no copy into `host`; *view points at the pieces in the lane's pinned staging buffer, valid until the gather after next)
  auto gather = [&](std::vector<dsa::PackItem> &items, std::vector<uint8_t> &host, const uint8_t **view) -> dsa_status {
    uint64_t total = 0;
    for (auto &it : items) { it.packed_off = total; total += ((uint64_t)it.len + 15) & ~15ull; }
    if (view) *view = nullptr; else host.resize(total);
    if (items.empty() || total == 0) return DSA_OK;
    uint8_t *d_packed = nullptr; dsa::PackItem *d_items = nullptr;
    hipError_t e = lane.packed.ensure(total);
    if (e == hipSuccess) e = lane.items.ensure(sizeof(dsa::PackItem) * items.size());
    d_packed = (uint8_t *)lane.packed.p; d_items = (dsa::PackItem *)lane.items.p;
    if (e == hipSuccess) e = hipMemcpyAsync(d_items, items.data(), sizeof(dsa::PackItem) * items.size(), hipMemcpyHostToDevice, lane.st);
    if (e == hipSuccess) { hipLaunchKernelGGL(dsa::k_enc_pack, dim3((uint32_t)items.size()), dim3(256), 0, lane.st, arena, d_packed, d_items, (uint32_t)items.size()); e = hipGetLastError(); }
    // through pinned staging (a pageable destination is staged by the runtime at a fraction of the link's rate)
    hostutil::Staging &stg = lane.stage[lane.next];
    lane.next ^= 1;
    if (e == hipSuccess) e = stg.acquire((size_t)total);
    if (e == hipSuccess) e = hipMemcpyAsync(stg.buf.p, d_packed, total, hipMemcpyDeviceToHost, lane.st);
    if (e == hipSuccess) e = hipStreamSynchronize(lane.st);
    if (e == hipSuccess) { if (view) *view = stg.buf.p; else hostutil::parallel_memcpy(host.data(), stg.buf.p, (size_t)total); }
    return e == hipSuccess ? DSA_OK : (e == hipErrorOutOfMemory ? DSA_ERR_OUT_OF_MEMORY : DSA_ERR_DEVICE);
  };
  auto scatter = [&](std::vector<dsa::PackItem> &items, const std::vector<uint8_t> &host) -> dsa_status {
    if (items.empty() || host.empty()) return DSA_OK;
    uint8_t *d_packed = nullptr; dsa::PackItem *d_items = nullptr;
    hipError_t e = lane.packed.ensure(host.size());
    if (e == hipSuccess) e = lane.items.ensure(sizeof(dsa::PackItem) * items.size());
    d_packed = (uint8_t *)lane.packed.p; d_items = (dsa::PackItem *)lane.items.p;
    if (e == hipSuccess) e = hipMemcpyAsync(d_items, items.data(), sizeof(dsa::PackItem) * items.size(), hipMemcpyHostToDevice, lane.st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_packed, host.data(), host.size(), hipMemcpyHostToDevice, lane.st);
    if (e == hipSuccess) { hipLaunchKernelGGL(dsa::k_enc_unpack, dim3((uint32_t)items.size()), dim3(256), 0, lane.st, arena, d_packed, d_items, (uint32_t)items.size()); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(lane.st);
    return e == hipSuccess ? DSA_OK : (e == hipErrorOutOfMemory ? DSA_ERR_OUT_OF_MEMORY : DSA_ERR_DEVICE);
  };
#define ENC_ST(call) do { dsa_status s_ = (call); if (s_ != DSA_OK) { return set_err(ctx, s_, "%s failed", #call); } } while (0)
  if (ns) {
    hipStream_t st = lane.st;
    ENC_TRY(lane.arena.ensure(cur ? cur : 256));
    ENC_TRY(lane.streams.ensure(sizeof(dsa::EncStream) * ns));
    arena = (uint8_t *)lane.arena.p; d_streams = (dsa::EncStream *)lane.streams.p;
    if (lane.walk_st) ENC_TRY(hipStreamSynchronize(lane.walk_st));     // (idle unless a previous chunk on this lane ended in an error)
    ENC_TRY(hipMemsetAsync(arena, 0, cur, st));              // histograms start at zero
    // uploads in pieces through the lane's two pinned staging buffers: host threads fill one while the DMA engine drains the
    // other (a pageable source would be staged by the runtime, one thread, a few GB/s)
    auto upload = [&](const std::vector<Upload> &ups) -> hipError_t {
      const uint64_t piece_cap = 192ull << 20;
      size_t i0 = 0;
      while (i0 < ups.size()) {
        const uint64_t lo = ups[i0].off;
        size_t i1 = i0 + 1;
        while (i1 < ups.size() && ups[i1].off + ups[i1].bytes - lo <= piece_cap) ++i1;
        const uint64_t hi = ups[i1 - 1].off + ups[i1 - 1].bytes;
        hostutil::Staging &stg = lane.stage[lane.next];
        lane.next ^= 1;
        hipError_t e = stg.acquire((size_t)(hi - lo));
        if (e != hipSuccess) return e;
        uint8_t *h = stg.buf.p;
        hostutil::parallel_for((uint32_t)(i1 - i0), [&](uint32_t k) {
          const Upload &u = ups[i0 + k];
          if (!u.narrow) { memcpy(h + (u.off - lo), u.src, u.bytes); return; }
          const uint32_t *src = (const uint32_t *)u.src;
          uint16_t *dst = (uint16_t *)(h + (u.off - lo));               // (offsets are multiples of 256)
          for (size_t e = 0, ne = u.bytes / 2; e < ne; ++e) dst[e] = (uint16_t)src[e];
        }, 2);
        e = hipMemcpyAsync(arena + lo, h, (size_t)(hi - lo), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = stg.submitted(st);
        if (e != hipSuccess) return e;
        i0 = i1;
      }
      return hipSuccess;
    };
    turn.acquire_a();
    ENC_TRY(upload(host_conn ? uploads : uploads_a));
    if (host_conn) turn.release();
    ENC_TRY(hipMemcpyAsync(d_streams, hs.data(), sizeof(dsa::EncStream) * ns, hipMemcpyHostToDevice, st));
    uint32_t maxv = 0;
    for (auto &S : hs) maxv = std::max(maxv, S.nv);
    const uint32_t gx = std::max(1u, std::min(64u, (maxv + 2047) / 2048));
    lap("layout + uploads queued");
    // ---- device phase 0: corner table, Edgebreaker symbols, attribute order, operand entries (one wave per mesh); meshes that
    // failed the host's checks have F = 0 and no arrays
    if (!host_conn) {
      ENC_TRY(lane.conns.ensure(sizeof(dsa::EncConn) * n));
      d_conns = (dsa::EncConn *)lane.conns.p;
      for (uint32_t i = 0; i < n; ++i) if (E->status[i] != DSA_OK) { memset(&hc[i], 0, sizeof(hc[i])); hc[i].status = dsa::ENC_ISOLATED; }
      ENC_TRY(hipMemcpyAsync(d_conns, hc.data(), sizeof(dsa::EncConn) * n, hipMemcpyHostToDevice, st));
      uint32_t maxf = 0;
      for (uint32_t i = 0; i < n; ++i) maxf = std::max(maxf, hc[i].F);
      const dim3 gt(std::max(1u, std::min(128u, (3u * maxf + 1023u) / 1024u)), n);        // table kernels: blocks per mesh x meshes
      hipLaunchKernelGGL(dsa::k_enc_table_clear, gt, dim3(256), 0, st, arena, d_conns, n);
      hipLaunchKernelGGL(dsa::k_enc_table_count, gt, dim3(256), 0, st, arena, d_conns, n);
      hipLaunchKernelGGL(dsa::k_enc_table_offsets, dim3(n), dim3(WAVE), 0, st, arena, d_conns, n);
      hipLaunchKernelGGL(dsa::k_enc_table_lists, gt, dim3(256), 0, st, arena, d_conns, n);
      hipLaunchKernelGGL(dsa::k_enc_table_opposites, gt, dim3(256), 0, st, arena, d_conns, n);
      hipLaunchKernelGGL(dsa::k_enc_table_corners, gt, dim3(256), 0, st, arena, d_conns, n);
      // the walks on their stream; the attribute values travel and are quantised meanwhile
      ENC_TRY(hipEventRecord(lane.tables_done, st));
      ENC_TRY(hipStreamWaitEvent(lane.walk_st, lane.tables_done, 0));
      hipLaunchKernelGGL(dsa::k_enc_connectivity, dim3((n + walk_lanes - 1) / walk_lanes), dim3(WAVE), 0, lane.walk_st, arena, d_conns, n, walk_lanes);
      ENC_TRY(hipEventRecord(lane.walk_done, lane.walk_st));
      turn.release();
      turn.acquire_b();
      ENC_TRY(upload(uploads));
      turn.release();
      hipLaunchKernelGGL(dsa::k_enc_bounds, dim3(ns), dim3(256), 0, st, arena, d_streams, ns);
      hipLaunchKernelGGL(dsa::k_enc_quantize, dim3(gx, ns), dim3(256), 0, st, arena, d_streams, ns);
      ENC_TRY(hipStreamWaitEvent(st, lane.walk_done, 0));
      hipLaunchKernelGGL(dsa::k_enc_operands, gt, dim3(256), 0, st, arena, d_conns, n);
    } else {
      hipLaunchKernelGGL(dsa::k_enc_bounds, dim3(ns), dim3(256), 0, st, arena, d_streams, ns);
      hipLaunchKernelGGL(dsa::k_enc_quantize, dim3(gx, ns), dim3(256), 0, st, arena, d_streams, ns);
    }
    // ---- device phase 1: quantise, order, correct, count
    hipLaunchKernelGGL(dsa::k_enc_gather, dim3(ns), dim3(256), 0, st, arena, d_streams, ns);
    hipLaunchKernelGGL(dsa::k_enc_corr, dim3(gx, ns), dim3(256), 0, st, arena, d_streams, ns);
    if (!host_plan) {       // device phase 2 follows at once: tables by k_enc_plan, no host round trip
      hipLaunchKernelGGL(dsa::k_enc_plan, dim3((ns + WAVE - 1) / WAVE), dim3(WAVE), 0, st, arena, d_streams, ns, (int)opt.force_scheme, (int)opt.compression_level);
      hipLaunchKernelGGL(dsa::k_enc_rans, dim3(ns), dim3(WAVE), 0, st, arena, d_streams, ns);
    }
    ENC_TRY(hipMemcpyAsync(hs.data(), d_streams, sizeof(dsa::EncStream) * ns, hipMemcpyDeviceToHost, st));
    if (!host_conn) ENC_TRY(hipMemcpyAsync(hc.data(), d_conns, sizeof(dsa::EncConn) * n, hipMemcpyDeviceToHost, st));
    ENC_TRY(hipStreamSynchronize(st));
    lap("device phases 0 + 1");
    if (!host_conn) {
      // what the stream layout needs of the connectivity: symbols, start-face bits, split events, two counts
      std::vector<dsa::PackItem> conn_items;
      for (uint32_t i = 0; i < n; ++i) {
        if (E->status[i] != DSA_OK) continue;
        const dsa::EncConn &C = hc[i];
        if (C.status != dsa::ENC_OK) {
          E->status[i] = DSA_ERR_INVALID_DATA; E->messages[i] = dsa::enc_conn_message(C.status);
          for (uint32_t sk = first_stream[i]; sk < first_stream[i + 1]; ++sk) hs[sk].overflow = 1;       // its attribute streams are not coded
          continue;
        }
        conn_items.push_back({C.symbols, 0, C.num_symbols, i});
        conn_items.push_back({C.start_bits, 0, C.num_start_bits, i});
        conn_items.push_back({C.splits, 0, 12u * C.num_splits, i});
        plans[i].interior_edges = (int64_t)C.interior_edges;
      }
      std::vector<uint8_t> unused;
      const uint8_t *conn_host = nullptr;
      ENC_ST(gather(conn_items, unused, &conn_host));
      hostutil::parallel_for((uint32_t)(conn_items.size() / 3), [&](uint32_t m) {
        const size_t k = 3 * (size_t)m;
        const uint32_t i = conn_items[k].pad;
        const dsa::EncConn &C = hc[i];
        synth::EbResult &eb = plans[i].eb;
        eb.num_split_symbols = C.num_split_symbols;
        if (conn_items[k].len) eb.symbols.assign(conn_host + conn_items[k].packed_off, conn_host + conn_items[k].packed_off + conn_items[k].len);
        if (conn_items[k + 1].len) eb.start_face_bits.assign(conn_host + conn_items[k + 1].packed_off, conn_host + conn_items[k + 1].packed_off + conn_items[k + 1].len);
        eb.splits.resize(C.num_splits);
        const uint32_t *sp = C.num_splits ? (const uint32_t *)(conn_host + conn_items[k + 2].packed_off) : nullptr;
        for (size_t q = 0; q < eb.splits.size(); ++q) eb.splits[q] = {sp[3 * q], sp[3 * q + 1], sp[3 * q + 2]};
      }, 8);
    }
  }
  lap("connectivity results");
  // ---- host phase 2: scheme choice and rANS tables from the device statistics
  std::vector<synth::SymbolPlan> splans(ns);
  std::vector<std::vector<uint32_t>> hists(ns);
  std::vector<int> stream_mesh(ns, 0);
  for (uint32_t i = 0; i < n; ++i) for (uint32_t s = first_stream[i]; s < first_stream[i + 1]; ++s) stream_mesh[s] = (int)i;
  if (host_plan) {
  {
    std::vector<dsa::PackItem> items;
    for (uint32_t s = 0; s < ns; ++s) {
      if (hs[s].overflow || hs[s].max_value >= hs[s].hist_cap) { hs[s].overflow = 1; continue; }
      items.push_back({hs[s].hist_raw, 0, 4u * (hs[s].max_value + 1u), s});
    }
    std::vector<uint8_t> host;
    ENC_ST(gather(items, host, nullptr));
    for (auto &it : items) {
      hists[it.pad].resize(it.len / 4);
      memcpy(hists[it.pad].data(), host.data() + it.packed_off, it.len);
    }
  }
  std::vector<std::string> plan_error(ns);
  auto plan_stream = [&](uint32_t s) {
    const uint32_t i = (uint32_t)stream_mesh[s];
    if (E->status[i] != DSA_OK) return;
    try {
      synth::check(!hs[s].overflow, "symbol outside the histogram range");
      synth::SymbolStats stt;
      stt.n = (size_t)hs[s].nv * hs[s].nc; stt.nc = (int)hs[s].nc; stt.max_value = hs[s].max_value; stt.total_bl = hs[s].total_bl;
      stt.tag_freq.assign(hs[s].hist_tag, hs[s].hist_tag + 33);
      stt.raw_freq.assign(hists[s].begin(), hists[s].end());
      synth::plan_symbols(stt, opt.force_scheme, opt.compression_level, splans[s]);
      hs[s].method = (uint32_t)splans[s].method;
      hs[s].precision_bits = (uint32_t)splans[s].coder.precision_bits;
      hs[s].num_symbols = splans[s].coder.num_symbols;
      synth::check(splans[s].coder.num_symbols <= std::max<uint32_t>(hs[s].hist_cap, 64), "alphabet larger than the table region");
    } catch (const std::exception &e) { plan_error[s] = e.what(); if (plan_error[s].empty()) plan_error[s] = "symbol plan failed"; hs[s].overflow = 1; }
  };
  hostutil::parallel_for(ns, plan_stream);
  {
    std::vector<dsa::PackItem> items;
    std::vector<uint8_t> host;
    for (uint32_t s = 0; s < ns; ++s) {
      const uint32_t i = (uint32_t)stream_mesh[s];
      if (E->status[i] != DSA_OK) continue;
      if (!plan_error[s].empty()) { E->status[i] = DSA_ERR_INVALID_DATA; E->messages[i] = plan_error[s]; continue; }
      const uint32_t bytes = 4u * splans[s].coder.num_symbols;
      for (int t = 0; t < 2; ++t) {
        const std::vector<uint32_t> &src = t == 0 ? splans[s].coder.prob : splans[s].coder.cum;
        items.push_back({t == 0 ? hs[s].prob : hs[s].cum, (uint64_t)host.size(), bytes, s});
        host.insert(host.end(), (const uint8_t *)src.data(), (const uint8_t *)src.data() + bytes);
        host.resize((host.size() + 15) & ~(size_t)15);
      }
    }
    ENC_ST(scatter(items, host));
  }
  } else {
    for (uint32_t s = 0; s < ns; ++s) {
      const uint32_t i = (uint32_t)stream_mesh[s];
      if (E->status[i] != DSA_OK || !hs[s].overflow) continue;
      E->status[i] = DSA_ERR_INVALID_DATA;
      E->messages[i] = hs[s].plan_status ? dsa::plan::plan_message((int)hs[s].plan_status) : "symbol outside the histogram range";
    }
  }
  lap("histograms + symbol plans");
  // ---- device phase 2: entropy coding
  std::vector<std::vector<uint8_t>> rans(ns), bits(ns);
  if (ns) {
    hipStream_t st = lane.st;
    if (host_plan) {
      ENC_TRY(hipMemcpyAsync(d_streams, hs.data(), sizeof(dsa::EncStream) * ns, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(dsa::k_enc_rans, dim3(ns), dim3(WAVE), 0, st, arena, d_streams, ns);
      ENC_TRY(hipMemcpyAsync(hs.data(), d_streams, sizeof(dsa::EncStream) * ns, hipMemcpyDeviceToHost, st));
      ENC_TRY(hipStreamSynchronize(st));
    }
    // coded bytes of every stream (and, when k_enc_plan made them, the probability tables: the stream carries them)
    std::vector<dsa::PackItem> items;
    for (uint32_t s = 0; s < ns; ++s) {
      if (hs[s].overflow || E->status[stream_mesh[s]] != DSA_OK) continue;
      items.push_back({hs[s].out_rans, 0, hs[s].rans_len, s});
      items.push_back({hs[s].out_bits, 0, hs[s].bits_len, s});
      items.push_back({hs[s].prob, 0, host_plan ? 0u : 4u * hs[s].num_symbols, s});
    }
    std::vector<uint8_t> unused;
    const uint8_t *host = nullptr;
    ENC_ST(gather(items, unused, &host));
    hostutil::parallel_for((uint32_t)(items.size() / 3), [&](uint32_t m) {
      const size_t k = 3 * (size_t)m;
      const uint32_t s = items[k].pad;
      if (items[k].len) rans[s].assign(host + items[k].packed_off, host + items[k].packed_off + items[k].len);
      if (items[k + 1].len) bits[s].assign(host + items[k + 1].packed_off, host + items[k + 1].packed_off + items[k + 1].len);
      if (!host_plan) {                      // the bytes in front of the payload: scheme, (raw: unique-symbols bit length), table
        synth::SymbolPlan &pl = splans[s];
        pl.method = (int)hs[s].method;
        pl.coder.num_symbols = hs[s].num_symbols;
        const uint32_t *pr = (const uint32_t *)(host + items[k + 2].packed_off);
        pl.coder.prob.assign(pr, pr + hs[s].num_symbols);
        pl.head.u8((uint8_t)pl.method);
        if (pl.method != 0) pl.head.u8((uint8_t)hs[s].usbl);
        pl.coder.write_table(pl.head);
      }
    }, 8);
  }
#undef ENC_TRY
#undef ENC_ST
  lap("device phase 2 + downloads");
  cleanup();
  // ---- host phase 3: stream layout (threads over meshes; write_stream may throw like any part of the host coder)
  auto layout_one = [&](uint32_t i) {
    if (E->status[i] != DSA_OK) return;
    bool bad = false;
    for (uint32_t s = first_stream[i]; s < first_stream[i + 1]; ++s) bad = bad || hs[s].overflow;
    if (bad) { E->status[i] = DSA_ERR_INVALID_DATA; E->messages[i] = "entropy coding failed"; return; }
    synth::ByteWriter w;
    const uint32_t s0 = first_stream[i];
    try {
    synth::write_stream(w, ins[i], plans[i],
      [&](synth::ByteWriter &bw, size_t k) {               // SequentialIntegerAttributeEncoder.cs:55-128
        const dsa::EncStream &S = hs[s0 + k];
        const synth::PortableAttr &a = plans[i].atts[k];
        if (S.kind == 1) { bw.i8(0); bw.i8(3); } else { bw.i8((int8_t)a.prediction); bw.i8(1); }
        bw.u8(1);
        bw.bytes(splans[s0 + k].head.d);
        bw.varint(rans[s0 + k].size());
        bw.bytes(rans[s0 + k]);
        if (S.method == 0) bw.bytes(bits[s0 + k]);
        if (S.kind == 1) { const int32_t max_q = (1 << S.bits) - 1; bw.i32(max_q); bw.i32((max_q - 1) / 2); }
        else { bw.i32(S.wrap_mn); bw.i32(S.wrap_mx); }
      },
      [&](synth::ByteWriter &bw, size_t k) {               // AttributeQuantizationTransform.cs:123-134 / AttributeOctahedronTransform.cs:44-47
        const dsa::EncStream &S = hs[s0 + k];
        if (S.kind == 0) { for (uint32_t c = 0; c < S.nc_out; ++c) bw.f32(S.qmin[c]); bw.f32(S.qrange); bw.u8((uint8_t)S.bits); }
        else if (S.kind == 1) bw.u8((uint8_t)S.bits);            // (an integer attribute has no transform to describe)
      });
    } catch (const std::exception &e) { E->status[i] = DSA_ERR_INVALID_DATA; E->messages[i] = e.what(); return; }
    E->streams[i].swap(w.d);
  };
  hostutil::parallel_for(n, layout_one);
  lap("stream layout");
  *out = E_owner.release();
  return DSA_OK;
}

uint32_t dsa_encoded_size(const dsa_encoded *e) { return e ? (uint32_t)e->streams.size() : 0; }

dsa_status dsa_encoded_stream(const dsa_encoded *e, uint32_t mesh, const uint8_t **bytes, size_t *length) {
  if (!e || mesh >= e->streams.size() || !bytes || !length) return DSA_ERR_INVALID_ARGUMENT;
  if (e->status[mesh] != DSA_OK) return set_err(e->ctx, (dsa_status)e->status[mesh], "mesh %u: %s", mesh, e->messages[mesh].c_str());
  *bytes = e->streams[mesh].data();
  *length = e->streams[mesh].size();
  return DSA_OK;
}

void dsa_encoded_free(dsa_encoded *e) { delete e; }

}  // extern "C"
