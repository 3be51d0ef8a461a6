// draco-sharp_amd/csrc/dsa_lanes.h
// Lane-per-chain kernels: the serial recurrences of the format that have no use for cross-lane help run with ONE
// LANE PER CHAIN, 64 chains (of 64 different meshes) per wave.  The chip has 256 scalar units but 65 536 vector
// lanes: a wave-per-mesh machine spends a scalar + a vector instruction stream on one chain, a lane-per-chain wave
// spends one vector instruction stream on 64 of them, so these kernels take a few per cent of the issue slots
// the wave-per-stream forms need and run at the latency of their dependent chain instead -- which is what bounds
// a 4096-mesh batch either way.  What makes the chain short is kept on chip per lane:
//   k_symbols_lanes   rANS (Entropy/RAnsDecoder.cs:20-99, RAnsSymbolDecoder.cs:21-59): per lane a 256-bucket
//                     first-symbol LUT and the cumulative frequencies of the non-zero symbols in LDS (lane regions
//                     at an odd dword stride: equal offsets of different lanes fall into different banks); one LUT
//                     read + one batch of five cumulative reads per symbol, stream bytes through a per-lane dword
//                     reservoir loaded one dword ahead, results stored four at a time.
//   k_predict_lanes   PredictionSchemeDeltaDecoder.cs:23-37, MeshPredictionSchemeParallelogramDecoder.cs:29-54 with
//                     PredictionSchemeWrapDecodingTransform.cs:46-75, and the octahedral delta
//                     (PredictionSchemeNormalOctahedron(Canonicalized)DecodingTransform.cs): one lane per attribute,
//                     operand indices and corrections loaded two entries ahead, the two latest results forwarded
//                     in registers.
// The per-lane bodies use no wave intrinsics, so tests/hostcheck compiles them for the host under ASan
// (test infrastructure; the product has no host decode path).
#pragma once
#include <stdint.h>

#include "dsa_common.h"
#include "dsa_locate.h"

namespace dsa {
namespace lanes {

// behaviour switches handed to the kernels (dsa_batch_decode reads them from the environment once; diagnostics)
#define LN_FLAG_SYMBOLS 1u    // raw rANS streams that fit a tier are decoded by k_symbols_lanes
#define LN_FLAG_PREDICT 2u    // prediction inverse by k_predict_lanes
#define LN_FLAG_OCT 8u        // octahedral-delta attributes (normals) by k_predict_oct_lanes: one lane per stream (diagnostic option)

// ------------------------------------------------------------------------------------------------ rANS, lane per stream
#define LN_MAX_PRECISION 15u          // cumulative frequencies are 16-bit in LDS (2^15 is the largest total)
#define LN_T0_SYMS 136u               // tiers by the number of non-zero symbols of the largest table among a wave's 64 streams:
#define LN_T1_SYMS 312u               //   136 -> 1476 B of LDS per lane (92 KB per wave), 312 -> 1828 B (114 KB), 440 -> 2084 B (130 KB)
#define LN_T2_SYMS 440u
#define LN_CUM_PAD 8u
#define LN_LUT_BITS 10u               // first-symbol LUT: 1024 buckets (4 slots each at 12-bit precision: at most 4 range starts per bucket)
// bytes of one lane's LDS region: 1024 x u8 LUT, (syms + pad) x u16 cumulative frequencies, 128-byte ring of stream
// bytes, the 16 symbol indices of the current block; padded so that the dword stride between lanes is odd (equal
// offsets of different lanes then fall into different banks)
__host__ __device__ constexpr uint32_t ln_sym_stride(uint32_t syms) {
  return ((1024u + 2u * (syms + LN_CUM_PAD) + 128u + 32u) / 4u) % 2u ? 1024u + 2u * (syms + LN_CUM_PAD) + 160u : 1024u + 2u * (syms + LN_CUM_PAD) + 164u;
}
__host__ __device__ constexpr uint32_t ln_sym_tier(uint32_t distinct) { return distinct <= LN_T0_SYMS ? LN_T0_SYMS : (distinct <= LN_T1_SYMS ? LN_T1_SYMS : LN_T2_SYMS); }

__host__ __device__ __forceinline__ bool ln_sym_eligible(const AttrDesc &a, const MeshLayout &L, uint32_t ai, uint32_t flags) {
  return (flags & LN_FLAG_SYMBOLS) && a.source == SRC_RAW && a.precision_bits <= LN_MAX_PRECISION && a.num_distinct >= 1 &&
         a.num_distinct <= LN_T2_SYMS && a.num_entries != 0 && (uint64_t)L.out_cap[ai] >= 4ull * a.num_distinct;
}

// The stream is consumed from its tail (RAnsDecoder.cs:58-61: state = state * 256 + buf[--offset]).  Global memory
// answers in 0.5-1 us when the chip is busy, so nothing inside the symbol loop waits for it: the stream is staged in
// a per-lane LDS ring of eight 16-byte chunks (chunk c = bytes [base - 16(c+1), base - 16c) in memory order; base =
// the 16-byte boundary behind the first byte to take), two chunks are requested at every 16-symbol block boundary and
// written into the ring at the next one.  The q-th byte taken (counted from `base` down) sits at ring byte
// (q ^ 15) & 127; every step reads the next two bytes while the table search runs and keeps 0, 1 or 2 of them.
// diagnostic build only (-DLN_STAMPS): shader-clock shares of the segments of a symbol step, summed per wave
#if defined(LN_STAMPS) && defined(__HIPCC__)
#define LN_STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (i) ln_seg[i] += t_ - ln_t; ln_t = t_; }
#else
#define LN_STAMP(i)
#endif

// Tables of one stream (RAnsSymbolDecoder.cs:21-59, RAnsDecoder.cs:69-88 restated as a search structure):
//   symtab[k]  k-th symbol with a non-zero frequency (global scratch: read off the chain)
//   cum[k]     its cumulative frequency; cum[distinct] = 2^P; LN_CUM_PAD entries of 0xFFFF behind it
//   lut[b]     k >> 1 for the k whose range holds slot b << (P - 10): the search starts at an even entry
// Returns 0, or the failure site.
__device__ __forceinline__ int ln_sym_build(Rd &r, uint32_t nsym, uint32_t P, uint32_t cap, uint8_t *lut, uint16_t *cum, uint32_t *symtab) {
  const uint32_t precision = 1u << P;
  uint32_t k = 0, run = 0;
  for (uint32_t i = 0; i < nsym; ++i) {
    const uint32_t pd = r.u8(), token = pd & 3u;
    if (token == 3u) {
      const uint32_t offset = pd >> 2;
      if (i + offset >= nsym) return 400;
      i += offset;
    } else {
      uint32_t pr = pd >> 2;
      for (uint32_t j = 0; j < token; ++j) pr |= r.u8() << (8 * (j + 1) - 2);
      if (pr) {
        if (k >= cap || pr > precision - run) return 401;
        symtab[k] = i; cum[k] = (uint16_t)run;
        run += pr; ++k;
      }
    }
  }
  if (!r.ok) return 400;
  if (run != precision || k == 0) return 401;
  cum[k] = (uint16_t)precision;
  for (uint32_t j = 1; j < LN_CUM_PAD; ++j) cum[k + j] = 0xFFFFu;
  const uint32_t sh = P - LN_LUT_BITS;
  uint32_t kk = 0;
  for (uint32_t b = 0; b < (1u << LN_LUT_BITS); ++b) {
    const uint32_t slot = b << sh;
    while (cum[kk + 1] <= slot) ++kk;
    lut[b] = (uint8_t)(kk >> 1);
  }
  return 0;
}

// RAnsDecoder.Read for every value of the stream (RAnsDecoder.cs:56-67) + zig-zag (BitUtilities.cs:94-103) unless the
// transform's corrections are positive; out[] is 16-byte aligned.  Blocks of 16 symbols: inside a block only LDS is
// touched (tables, byte ring, the block's symbol indices); at a block boundary the symbol ids of the block are
// gathered from symtab[] -- to be zig-zagged and stored, four at a time, one boundary later -- and the byte ring is
// topped up the same way.
__device__ __forceinline__ void ln_sym_decode(const uint8_t *arena, uint64_t stream_off, uint32_t rans_off, uint32_t x, uint32_t off, uint32_t P,
                                              const uint8_t *lut, const uint16_t *cum, uint32_t *ring, uint16_t *kbuf, const uint32_t *symtab,
                                              uint32_t num_values, bool positive, int32_t *out, uint32_t *dbg) {
  const uint32_t mask = (1u << P) - 1u, l_base = 4u << P, sh = P - LN_LUT_BITS;
#if defined(LN_STAMPS) && defined(__HIPCC__)
  unsigned long long ln_t = 0, ln_seg[4] = {0, 0, 0, 0};
#endif
  const uint64_t lowest = stream_off & ~15ull;
  const uint64_t last = stream_off + rans_off + (off ? off - 1u : 0u);
  const uint64_t base = (last + 16ull) & ~15ull;
  // the ring starts with six chunks (the only loads this lane waits for), two more are on their way
  uint32_t loaded = 0;
  for (; loaded < 6; ++loaded) {
    const Chunk c = ln_load_chunk(arena, base, lowest, loaded);
    for (int k = 0; k < 4; ++k) ring[((loaded & (LN_RING_CHUNKS - 1u)) << 2) + k] = c.d[k];
  }
  Chunk in0 = ln_load_chunk(arena, base, lowest, loaded), in1 = ln_load_chunk(arena, base, lowest, loaded + 1);
  bool inflight = true;
  const uint8_t *ring8 = (const uint8_t *)ring;
  uint32_t q = (uint32_t)(base - 1 - last);        // index of the next byte, counted from `base` down
  const uint32_t q_end = q + off;                  // bytes are left while q < q_end
  uint32_t g[LN_BLOCK];
#pragma unroll
  for (uint32_t j = 0; j < LN_BLOCK; ++j) g[j] = 0;
  auto zz = [&](uint32_t v) -> uint32_t { return positive ? v : ((v & 1u) ? (uint32_t)(-(int32_t)(v >> 1) - 1) : (v >> 1)); };
  // the first step may need bytes as well (a stream whose initial state is below l_base cannot exist, but is cheap to allow)
  while (x < l_base && q < q_end) { x = (x << 8) | ring8[(q ^ 15u) & 127u]; ++q; }
  // one block: cnt symbols, LDS only.  x >= l_base (or the stream is exhausted) on entry to every step.
  auto decode_block = [&](uint32_t cnt) {
    for (uint32_t j = 0; j < cnt; ++j) {
      // the next two stream bytes, read while the search runs
      LN_STAMP(0);
      const uint32_t b0 = ring8[(q ^ 15u) & 127u], b1 = ring8[((q + 1u) ^ 15u) & 127u];
      const uint32_t rem = x & mask;
      uint32_t k = 2u * (uint32_t)lut[rem >> sh];
      LN_STAMP(1);
      // entries k .. k + 7 in four aligned dwords; a bucket of four slots holds at most four range starts behind
      // its first symbol (k or k + 1), so at 12-bit precision the symbol is one of k .. k + 5
      const uint32_t *cw = (const uint32_t *)(cum + k);
      const uint32_t w0 = cw[0], w1 = cw[1], w2 = cw[2], w3 = cw[3];
      LN_STAMP(2);
      uint32_t cs = w0 & 0xFFFFu, cn = w0 >> 16;
      if (rem >= cn) { ++k; cs = cn; cn = w1 & 0xFFFFu; }
      if (rem >= cn) { ++k; cs = cn; cn = w1 >> 16; }
      if (rem >= cn) { ++k; cs = cn; cn = w2 & 0xFFFFu; }
      if (rem >= cn) { ++k; cs = cn; cn = w2 >> 16; }
      if (rem >= cn) { ++k; cs = cn; cn = w3 & 0xFFFFu; }
      while (rem >= cn) { ++k; cs = cn; cn = cum[k + 1]; }       // only above 12-bit precision (wider buckets)
      x = (cn - cs) * (x >> P) + rem - cs;
      kbuf[j] = (uint16_t)k;
      // x >= 4 after a step (RAnsDecoder.cs:63-65 with x >= l_base before it), so two bytes reach l_base for P <= 15
      uint32_t nb = (x < l_base ? 1u : 0u) + (x < (l_base >> 8) ? 1u : 0u);
      const uint32_t left = q_end - q;
      nb = nb < left ? nb : left;
      x = (x << (8u * nb)) | (((b0 << 8) | b1) >> (16u - 8u * nb));
      q += nb;
      while (x < l_base && q < q_end) { x = (x << 8) | ring8[(q ^ 15u) & 127u]; ++q; }   // a state below 4: malformed streams only
      LN_STAMP(3);
    }
  };
  // Block boundary.  Order matters for the in-order memory counter: everything that was requested one boundary ago is
  // consumed first (symbol ids, then the two chunks), only then are this boundary's stores and requests issued.  The
  // first boundary has nothing to store; it is peeled so that no branch joins a path with stores in flight.
  auto boundary = [&](uint32_t b0, uint32_t cnt, bool store_prev) {
    uint32_t z[LN_BLOCK];
#pragma unroll
    for (uint32_t q = 0; q < LN_BLOCK; ++q) z[q] = zz(g[q]);
    if (inflight) {
      for (int k = 0; k < 4; ++k) { ring[((loaded & (LN_RING_CHUNKS - 1u)) << 2) + k] = in0.d[k]; ring[(((loaded + 1u) & (LN_RING_CHUNKS - 1u)) << 2) + k] = in1.d[k]; }
      loaded += 2;
    }
    if (store_prev) {                                  // the previous block was a full one
      int32_t *o = out + (b0 - LN_BLOCK);
#pragma unroll
      for (uint32_t q = 0; q < LN_BLOCK; q += 4) {
#if defined(__HIPCC__)
        *(uint4 *)(o + q) = make_uint4(z[q], z[q + 1], z[q + 2], z[q + 3]);
#else
        for (uint32_t t = 0; t < 4; ++t) o[q + t] = (int32_t)z[q + t];
#endif
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < LN_BLOCK; ++j) g[j] = symtab[kbuf[j < cnt ? j : 0u]];
    // two more chunks are requested while the ring has room for them
    inflight = loaded + 2u - (q >> 4) <= LN_RING_CHUNKS;
    if (inflight) { in0 = ln_load_chunk(arena, base, lowest, loaded); in1 = ln_load_chunk(arena, base, lowest, loaded + 1); }
  };
  {
    const uint32_t cnt = num_values < LN_BLOCK ? num_values : LN_BLOCK;
    decode_block(cnt);
    boundary(0, cnt, false);
  }
  for (uint32_t b0 = LN_BLOCK; b0 < num_values; b0 += LN_BLOCK) {
    const uint32_t cnt = num_values - b0 < LN_BLOCK ? num_values - b0 : LN_BLOCK;
    decode_block(cnt);
    boundary(b0, cnt, true);
  }
#if defined(LN_STAMPS) && defined(__HIPCC__)
  if (dbg) { dbg[10] = (uint32_t)(ln_seg[1] >> 12) | ((uint32_t)(ln_seg[2] >> 12) << 16); dbg[11] = (uint32_t)(ln_seg[3] >> 12); }   // slots no other kernel writes
#endif
  // the last block (possibly partial)
  const uint32_t tail = num_values % LN_BLOCK ? num_values % LN_BLOCK : LN_BLOCK, tbase = num_values - tail;
#pragma unroll
  for (uint32_t j = 0; j < LN_BLOCK; ++j) if (j < tail) out[tbase + j] = (int32_t)zz(g[j]);
}

// One stream, start to finish (the body of a lane of k_symbols_lanes; tests/hostcheck calls it directly).
__device__ __forceinline__ void ln_symbols_stream(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t cap, uint16_t *lds) {
  const AttrDesc &a = D->att[ai];
  const uint8_t *s = arena + L.stream;
  uint8_t *lut = (uint8_t *)lds;
  uint16_t *cum = lds + (1u << LN_LUT_BITS) / 2;
  uint32_t *ring = (uint32_t *)(cum + cap + LN_CUM_PAD);          // cap is a multiple of 8: dword aligned
  uint16_t *kbuf = (uint16_t *)(ring + 4 * LN_RING_CHUNKS);
  uint32_t *symtab = (uint32_t *)(arena + L.out[ai]);
  Rd r(s, L.stream_len, a.off_table);
  const int site = ln_sym_build(r, a.num_symbols, a.precision_bits, cap, lut, cum, symtab);
  if (site) { fail(D, ST_INVALID, site); return; }
  const uint8_t *buf = s + a.off_rans;
  uint32_t x = 0, off = 0;
  if (!rans_init(buf, a.size_rans, 4u << a.precision_bits, &x, &off)) { fail(D, ST_INVALID, 402); return; }
  const bool positive = a.have_scheme && (a.pred_transform == 2 || a.pred_transform == 3);   // D-4
  ln_sym_decode(arena, L.stream, a.off_rans, x, off, a.precision_bits, lut, cum, ring, kbuf, symtab, a.num_entries * a.nc_portable, positive, (int32_t *)(arena + L.work[ai]), ai == 0 ? D->dbg : nullptr);
}

// ------------------------------------------------------------------------------------- prediction inverse, lane per attribute
template <int NC>
struct Vec { int32_t v[NC]; };

// Difference / Parallelogram + wrap transform (PredictionSchemeDeltaDecoder.cs:23-37,
// MeshPredictionSchemeParallelogramDecoder.cs:29-54, PredictionSchemeWrapDecodingTransform.cs:46-75), in place on
// w[entries][NC]: corrections in, portable values out.  para[3p..3p+2] = entries (next, prev, opposite) of the
// parallelogram of entry p, next = INVALID where the entry falls back to delta (para_operands_of); para == nullptr:
// delta for every entry.  Software pipeline of depth one: while entry p is computed, the operands of entry p + 1
// that are finished (index < p) and its correction are already on their way and the indices of entry p + 2 are
// loaded; an operand that is entry p itself is forwarded in registers.  The loop is unrolled twice so that the two
// slots swap roles instead of being copied.
template <int NC>
struct PredSlot { uint32_t en, ep, eo; Vec<NC> vn, vp, vo, corr; };

// Every load of the pipeline is unconditional (indices clamped into the array): an operand that is not final yet is
// loaded anyway and replaced by the forwarded register, which costs an address-unit slot instead of an exec-mask
// region.  A delta entry is written as the parallelogram (p - 1, 0, 0): o[p-1] + o[0] - o[0].
template <int NC, bool PARA>
__device__ __forceinline__ void ln_predict_wrap(int32_t *w, const uint32_t *para, uint32_t entries, int32_t mn, int32_t mx) {
  typedef Vec<NC> V;
  const int32_t max_dif = 1 + mx - mn;
  V *wv = (V *)w;
  const uint32_t last = entries - 1;
  auto load_idx = [&](uint32_t q, uint32_t &en, uint32_t &ep, uint32_t &eo) {
    if (PARA) {
      const Vec<3> t = *(const Vec<3> *)(para + 3 * (size_t)(q < last ? q : last));
      en = (uint32_t)t.v[0]; ep = (uint32_t)t.v[1]; eo = (uint32_t)t.v[2];
      if (en == DSA_INVALID) { en = q - 1; ep = 0; eo = 0; }      // delta fallback (entry 0: en = INVALID again, forwarded r1 = 0)
    }
  };
  auto request = [&](PredSlot<NC> &s, uint32_t q) {
    s.corr = wv[q < last ? q : last];
    if (PARA) {
      s.vn = wv[s.en < last ? s.en : last];
      s.vp = wv[s.ep < last ? s.ep : last];
      s.vo = wv[s.eo < last ? s.eo : last];
    }
  };
  V r1;                                            // value of the previous entry
#pragma unroll
  for (int c = 0; c < NC; ++c) r1.v[c] = 0;
  auto compute = [&](const PredSlot<NC> &s, uint32_t q) {
    V o;
    const bool fn = s.en + 1 == q, fp = s.ep + 1 == q, fo = s.eo + 1 == q;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      int32_t pred;
      if (PARA) {
        const int32_t a = fn ? r1.v[c] : s.vn.v[c], b = fp ? r1.v[c] : s.vp.v[c], d = fo ? r1.v[c] : s.vo.v[c];
        pred = (int32_t)((uint32_t)a + (uint32_t)b - (uint32_t)d);
      } else {
        pred = r1.v[c];                            // delta; entry 0: r1 = 0
      }
      o.v[c] = wrap_original(pred, s.corr.v[c], mn, mx, max_dif);
    }
    wv[q] = o;
    r1 = o;
  };
  PredSlot<NC> s0, s1;
  s0.en = s0.ep = s0.eo = s1.en = s1.ep = s1.eo = 0;
  load_idx(0, s0.en, s0.ep, s0.eo);
  load_idx(1, s1.en, s1.ep, s1.eo);
  request(s0, 0);
  uint32_t p = 0;
  while (p < entries) {
    // even step: s0 = entry p, s1 = entry p + 1 (indices known); the indices of entry p + 2 go to s0 afterwards
    uint32_t n2 = 0, p2 = 0, o2 = 0;
    load_idx(p + 2, n2, p2, o2);
    request(s1, p + 1);
    compute(s0, p);
    s0.en = n2; s0.ep = p2; s0.eo = o2;
    if (++p >= entries) break;
    // odd step: the roles of the slots are swapped
    load_idx(p + 2, n2, p2, o2);
    request(s0, p + 1);
    compute(s1, p);
    s1.en = n2; s1.ep = p2; s1.eo = o2;
    ++p;
  }
}

// Octahedral delta (PredictionSchemeDeltaDecoder.cs:23-37 with PredictionSchemeNormalOctahedron(Canonicalized)
// DecodingTransform.ComputeOriginalValue), in place on w[entries][2].  The chain has no structure a wave could scan
// when the values keep changing their class (which quadrant / which side of the diamond the previous value lies in:
// the normals of a height field straddle x = 0 all the time), and a wave-uniform machine then spends ~15 scalar + ~15
// vector instructions per entry of ONE stream (2 + 2 G per 4096-mesh batch, 10 ms as the last kernel of the decode).
// One lane per stream: the same instructions serve up to 64 streams, and what counts is the length of the dependent chain
// per entry (below).  Blocks of 8 entries, fully unrolled: the corrections of the block are in registers, those of the next
// two blocks are loading, the results leave with four 16-byte stores at the block's end -- nothing inside a block waits
// for memory, and the kernel fits the registers one entropy-decode wave leaves behind (it starts beside them).
// The canonicalised transform runs as the recursion in the canonical frame of dsa_common.h (OctLane, oct_lane_fast, oct_lane_exact).
#define LN_OCT_BLOCK 8u
__device__ __forceinline__ void ln_predict_oct(int32_t *w, uint32_t entries, int32_t max_q, bool canonical) {
  OctParams o;
  const int q = 32 - __clz(max_q);
  const int32_t max_value = (1 << q) - 2;
  o.center = max_value / 2;
  o.max_q = (1 << q) - 1;
  if (!canonical) {      // the plain transform (no stock encoder writes it): the reference's step, entry by entry from memory
    int32_t ps = 0, pt = 0;
    for (uint32_t e = 0; e < entries; ++e) {
      int32_t os, ot;
      oct_original(o, false, ps, pt, w[2 * (size_t)e], w[2 * (size_t)e + 1], os, ot);
      w[2 * (size_t)e] = os; w[2 * (size_t)e + 1] = ot;
      ps = os; pt = ot;
    }
    return;
  }
  OctLane st;
  st.vs = 0; st.vt = 0;
  oct_lane_state(o, st);
  // entries [from, to) one by one from memory: the fast step where it is entitled to, else the reference's
  auto rolled = [&](uint32_t from, uint32_t to) {
    for (uint32_t e = from; e < to; ++e) {
      const int32_t cx = w[2 * (size_t)e], cy = w[2 * (size_t)e + 1];
      int32_t os, ot;
      OctLane t = st;
      if (oct_lane_fast(o, t, cx, cy, os, ot)) { st = t; st.vs = os; st.vt = ot; }
      else oct_lane_exact(o, st, cx, cy, os, ot);
      w[2 * (size_t)e] = os; w[2 * (size_t)e + 1] = ot;
    }
  };
  // Whole blocks while the next block is a whole block too (every load of the loop is in range and a constant distance
  // ahead); the last one or two blocks go through the rolled loop.
  constexpr uint32_t CH = LN_OCT_BLOCK / 2;                // 16-byte chunks (two entries) per block
  const uint32_t full = entries / LN_OCT_BLOCK;
  const uint32_t nb = full >= 1u ? full - 1u : 0u;
  if (nb) {
    Chunk cur[CH], nxt[CH];          // an entry's value takes the place of its correction in cur
    auto load = [&](uint32_t c) -> Chunk {
      Chunk r;
#if defined(__HIPCC__)
      const uint4 v = ((const uint4 *)w)[c];
      r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w;
#else
      memcpy(r.d, w + 4 * (size_t)c, 16);
#endif
      return r;
    };
#pragma unroll
    for (uint32_t k = 0; k < CH; ++k) cur[k] = load(k);
    for (uint32_t b = 0; b < nb; ++b) {
      const uint32_t c0 = b * CH;
#pragma unroll
      for (uint32_t k = 0; k < CH; ++k) nxt[k] = load(c0 + CH + k);
      // the block on the fast path alone (short straight-line code); a block in which some entry was not entitled to it is
      // done again from its first entry by the rolled loop, from memory (its inputs are still there)
      const OctLane st0 = st;
      bool all_ok = true;
#pragma unroll
      for (uint32_t j = 0; j < LN_OCT_BLOCK; ++j) {
        int32_t os, ot;
        all_ok = oct_lane_fast(o, st, (int32_t)cur[j / 2].d[(j & 1u) * 2u], (int32_t)cur[j / 2].d[(j & 1u) * 2u + 1u], os, ot) && all_ok;
        cur[j / 2].d[(j & 1u) * 2u] = (uint32_t)os; cur[j / 2].d[(j & 1u) * 2u + 1u] = (uint32_t)ot;
        st.vs = os; st.vt = ot;
      }
      if (all_ok) {
#pragma unroll
        for (uint32_t k = 0; k < CH; ++k) {
#if defined(__HIPCC__)
          ((uint4 *)w)[c0 + k] = make_uint4(cur[k].d[0], cur[k].d[1], cur[k].d[2], cur[k].d[3]);
#else
          memcpy(w + 4 * (size_t)(c0 + k), cur[k].d, 16);
#endif
        }
      } else {
        st = st0;
        rolled(b * LN_OCT_BLOCK, (b + 1) * LN_OCT_BLOCK);
      }
#pragma unroll
      for (uint32_t k = 0; k < CH; ++k) cur[k] = nxt[k];
    }
  }
  rolled(nb * LN_OCT_BLOCK, entries);
}

__host__ __device__ __forceinline__ bool ln_oct_eligible(const AttrDesc &a, uint32_t flags) {
  return (flags & LN_FLAG_OCT) && a.have_scheme && a.source != SRC_BYTES && (a.pred_transform == 2 || a.pred_transform == 3) && a.pred_kind == 0 && a.num_entries != 0;
}

// One attribute (the body of a lane of k_predict_lanes).  phase 0: schemes that need no traversal data (difference,
// octahedral delta); phase 1: parallelogram.
__device__ __forceinline__ void ln_predict_attribute(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t phase) {
  const AttrDesc &a = D->att[ai];
  if (!a.have_scheme || a.source == SRC_BYTES || a.num_entries == 0) return;
  if ((a.pred_kind == 1) != (phase == 1)) return;
  int32_t *w = (int32_t *)(arena + L.work[ai]);
  if (a.pred_transform == 1) {
    const uint32_t *para = (const uint32_t *)(arena + L.para);
    const uint32_t nc = a.nc_portable, e = a.num_entries;
    const int32_t mn = a.wrap_min, mx = a.wrap_max;
    if (a.pred_kind == 1) {
      if (nc == 1) ln_predict_wrap<1, true>(w, para, e, mn, mx);
      else if (nc == 2) ln_predict_wrap<2, true>(w, para, e, mn, mx);
      else if (nc == 3) ln_predict_wrap<3, true>(w, para, e, mn, mx);
      else if (nc == 4) ln_predict_wrap<4, true>(w, para, e, mn, mx);
      else fail(D, ST_NOTIMPL, 500);
    } else {
      if (nc == 1) ln_predict_wrap<1, false>(w, para, e, mn, mx);
      else if (nc == 2) ln_predict_wrap<2, false>(w, para, e, mn, mx);
      else if (nc == 3) ln_predict_wrap<3, false>(w, para, e, mn, mx);
      else if (nc == 4) ln_predict_wrap<4, false>(w, para, e, mn, mx);
      else fail(D, ST_NOTIMPL, 500);
    }
  } else {
    if (a.pred_kind != 0) { fail(D, ST_NOTIMPL, 501); return; }
    ln_predict_oct(w, a.num_entries, a.oct_max_q, a.pred_transform == 3);
  }
}

#if defined(__HIPCC__)
template <uint32_t SYMS>
__global__ __launch_bounds__(WAVE) void k_symbols_lanes(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t flags) {
  __shared__ uint32_t ln_lds[WAVE * ln_sym_stride(SYMS) / 4];
  const uint32_t lane = threadIdx.x, mesh = blockIdx.x * WAVE + lane, ai = blockIdx.y;
  bool mine = false;
  uint32_t distinct = 0;
  MeshDesc *D = nullptr;
  if (mesh < n) {
    D = &descs[mesh];
    if (D->status == ST_OK && !D->general && ai < D->num_attributes) {
      mine = ln_sym_eligible(D->att[ai], layouts[mesh], ai, flags);
      distinct = mine ? D->att[ai].num_distinct : 0u;
    }
  }
  // the wave's LDS is sized by the largest table among its 64 streams: one launch per tier, the others leave
  uint32_t mx = distinct;
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, d, WAVE); mx = o > mx ? o : mx; }
  if (mx == 0 || ln_sym_tier(mx) != SYMS || !mine) return;
  ln_symbols_stream(arena, layouts[mesh], D, ai, SYMS, (uint16_t *)((uint8_t *)ln_lds + lane * ln_sym_stride(SYMS)));
}
#endif

#if defined(__HIPCC__)
// Octahedral-delta attributes, LPW meshes per wave (see k_predict_lanes below for why not 64).
template <uint32_t LPW>
__global__ __launch_bounds__(WAVE) void k_predict_oct_lanes(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t flags) {
  __builtin_amdgcn_s_setprio(3);   // a handful of long chains that hundreds of short-lived waves wait for
  const uint32_t lane = threadIdx.x, mesh = blockIdx.x * LPW + lane, ai = blockIdx.y;
  if (lane >= LPW || mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || ai >= D->num_attributes) return;
  const AttrDesc &a = D->att[ai];
  if (!ln_oct_eligible(a, flags)) return;
  ln_predict_oct((int32_t *)(arena + layouts[mesh].work[ai]), a.num_entries, a.oct_max_q, a.pred_transform == 3);
}

// LPW meshes per wave (the other lanes stay idle): a fully divergent wave access costs the CU's address unit one
// cycle per active lane, so the chains of a batch are spread over all CUs rather than packed 64 to a wave.
template <uint32_t LPW>
__global__ __launch_bounds__(WAVE) void k_predict_lanes(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t phase) {
  const uint32_t lane = threadIdx.x, mesh = blockIdx.x * LPW + lane, ai = blockIdx.y;
  if (lane >= LPW || mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK || D->general || ai >= D->num_attributes) return;
  ln_predict_attribute(arena, layouts[mesh], D, ai, phase);
}
#endif

}  // namespace lanes
}  // namespace dsa
