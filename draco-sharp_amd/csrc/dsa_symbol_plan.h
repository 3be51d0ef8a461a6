// draco-sharp_amd/csrc/dsa_symbol_plan.h
//
// Symbol-scheme choice and rANS frequency-table normalisation of the encode direction, written once for the host coder
// (dsa_encode_host.h) and for the device (k_enc_plan, dsa_encode.h): plain sequential code over raw arrays, every
// floating-point operation an IEEE add / multiply / divide / floor in a fixed order (the library is built with
// -ffp-contract=off on both sides), so that the two produce the same tables bit for bit.
//   scheme choice        Entropy/SymbolEncoding.cs:8-40 (E-2 corrected), RAnsSymbolCoding.cs:29-41
//   table normalisation  Entropy/RAnsSymbolEncoder.cs:15-123
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define DSA_PLAN_HD __host__ __device__ inline
#else
#define DSA_PLAN_HD inline
#endif

namespace dsa {
namespace plan {

// log2 of a positive finite double from exact operations only (exponent extraction, then atanh series of the mantissa
// reduced to [sqrt(1/2), sqrt(2))): the same bits on every IEEE machine, about 1e-15 from the true value.  The scheme choice
// compares sums of f * log2(f / n); libm's log2 differs in the last place between the host and the device library.
DSA_PLAN_HD double det_log2(double x) {
  union { double d; uint64_t u; } v;
  v.d = x;
  int e = (int)((v.u >> 52) & 0x7FF) - 1023;
  if (e == -1023) {                                   // subnormal: scale up (never met: arguments are ratios of counts)
    v.d = x * 4503599627370496.0;
    e = (int)((v.u >> 52) & 0x7FF) - 1023 - 52;
  }
  v.u = (v.u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;      // mantissa in [1, 2)
  double m = v.d;
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  const double t = (m - 1.0) / (m + 1.0), t2 = t * t;               // ln m = 2 atanh t, |t| < 0.1716
  double s = 0.0;
  for (int k = 27; k >= 1; k -= 2) s = s * t2 + 1.0 / (double)k;    // t^26 / 27 < 1e-21
  return (double)e + 2.0 * t * s * 1.4426950408889634;
}

// -(sum of f * log2(f / n)) over the non-zero frequencies, truncated; the number of non-zero frequencies in *num_unique.
template <class F>
DSA_PLAN_HD int64_t shannon_bits(const F *f, size_t count, double n, int *num_unique) {
  double bits = 0.0;
  int u = 0;
  for (size_t i = 0; i < count; ++i)
    if (f[i] > 0) { ++u; bits += (double)f[i] * det_log2((double)f[i] / n); }
  *num_unique = u;
  return (int64_t)(-bits);
}
DSA_PLAN_HD int64_t approx_table_bits(int max_value, int num_unique) {   // RAnsSymbolCoding.cs:29-41
  const int64_t zero_bits = 8 * ((int64_t)num_unique + (max_value - num_unique) / 64);
  return 8 * (int64_t)num_unique + zero_bits;
}
DSA_PLAN_HD int plan_msb(uint32_t v) { int r = 0; while (v >>= 1) ++r; return r; }

enum { PLAN_OK = 0, PLAN_EMPTY_TOP = 1, PLAN_SUM = 2, PLAN_UNIQUE = 3, PLAN_PROB = 4 };

// Frequencies -> probabilities that sum to 2^precision_bits, and their running sums (RAnsSymbolEncoder.cs:15-123).
// order / tmp: scratch of num_symbols entries each (the symbols sorted by probability, stably: what std::stable_sort gives).
template <class F>
DSA_PLAN_HD int rans_tables(int max_bit_length, const F *freq, size_t count, uint32_t *prob, uint32_t *cum, uint32_t *order, uint32_t *tmp,
                            int *precision_bits_out, uint32_t *num_symbols_out) {
  const int p = (3 * max_bit_length) / 2;
  const int precision_bits = p < 12 ? 12 : (p > 20 ? 20 : p);
  const uint32_t precision = 1u << precision_bits;
  uint64_t total = 0;
  uint32_t max_valid = 0;
  for (size_t i = 0; i < count; ++i) { total += (uint64_t)freq[i]; if (freq[i]) max_valid = (uint32_t)i; }
  const uint32_t num_symbols = max_valid + 1;
  *precision_bits_out = precision_bits;
  *num_symbols_out = num_symbols;
  const double total_d = (double)total, prec_d = (double)precision;
  int64_t total_prob = 0;
  for (uint32_t i = 0; i < num_symbols; ++i) {
    const double pr = (double)freq[i] / total_d;
    uint32_t rp = (uint32_t)(pr * prec_d + 0.5);
    if (rp == 0 && freq[i] > 0) rp = 1;
    prob[i] = rp;
    total_prob += rp;
  }
  if (total_prob != (int64_t)precision) {
    // stable sort by probability, ascending: least-significant-digit radix sort, three 8-bit digits cover 2^20 + a margin
    for (uint32_t i = 0; i < num_symbols; ++i) order[i] = i;
    for (int pass = 0; pass < 3; ++pass) {
      uint32_t cnt[257];
      for (int d = 0; d <= 256; ++d) cnt[d] = 0;
      const int shift = 8 * pass;
      for (uint32_t i = 0; i < num_symbols; ++i) ++cnt[((prob[order[i]] >> shift) & 255u) + 1u];
      for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
      for (uint32_t i = 0; i < num_symbols; ++i) { const uint32_t s = order[i]; tmp[cnt[(prob[s] >> shift) & 255u]++] = s; }
      for (uint32_t i = 0; i < num_symbols; ++i) order[i] = tmp[i];
    }
    if (total_prob < (int64_t)precision) {
      prob[order[num_symbols - 1]] += (uint32_t)((int64_t)precision - total_prob);
    } else {
      int64_t error = total_prob - (int64_t)precision;
      while (error > 0) {
        const double rel = prec_d / (double)total_prob;
        for (int64_t j = (int64_t)num_symbols - 1; j >= 0; --j) {
          const uint32_t sid = order[j];
          if (prob[sid] <= 1) { if (j == (int64_t)num_symbols - 1) return PLAN_EMPTY_TOP; break; }
          const double scaled = rel * (double)prob[sid];
          int32_t np = (int32_t)scaled;                       // floor: scaled >= 0
          if ((double)np > scaled) --np;
          int32_t fix = (int32_t)prob[sid] - np;
          if (fix == 0) fix = 1;
          if (fix >= (int32_t)prob[sid]) fix = (int32_t)prob[sid] - 1;
          if (fix > error) fix = (int32_t)error;
          prob[sid] -= (uint32_t)fix; total_prob -= fix; error -= fix;
          if (total_prob == (int64_t)precision) break;
        }
      }
    }
  }
  uint32_t c = 0;
  for (uint32_t i = 0; i < num_symbols; ++i) { cum[i] = c; c += prob[i]; if (prob[i] >= (1u << 22)) return PLAN_PROB; }
  return c == precision ? PLAN_OK : PLAN_SUM;
}

// Scheme choice (0 tagged, 1 raw) and, for the raw scheme, the unique-symbols bit length that sets the rANS precision.
template <class F>
DSA_PLAN_HD int choose_scheme(const F *tag_freq, const F *raw_freq, uint32_t max_value, uint64_t n, uint32_t nc, uint64_t total_bl,
                              int force_scheme, int compression_level, int *method_out, int *usbl_out) {
  int nu_tag = 0, nu_raw = 0;
  const int64_t tag_bits = shannon_bits(tag_freq, 33, (double)(n / nc), &nu_tag);
  const int64_t tagged_total = tag_bits + approx_table_bits(nu_tag, nu_tag) + (int64_t)total_bl * (int64_t)nc;
  const int64_t raw_total = shannon_bits(raw_freq, (size_t)max_value + 1, (double)n, &nu_raw) + approx_table_bits((int)max_value, nu_raw);
  const int max_value_bl = plan_msb(max_value > 1u ? max_value : 1u) + 1;
  int method = force_scheme;
  if (method < 0) method = (tagged_total < raw_total || max_value_bl > 18) ? 0 : 1;
  *method_out = method;
  *usbl_out = 0;
  if (method != 0) {
    int usbl = (nu_raw > 0 ? plan_msb((uint32_t)nu_raw) : 0) + 1;
    if (usbl > 18) return PLAN_UNIQUE;
    if (compression_level < 4) usbl -= 2;
    else if (compression_level < 6) usbl -= 1;
    else if (compression_level > 9) usbl += 2;
    else if (compression_level > 7) usbl += 1;
    usbl = usbl < 1 ? 1 : (usbl > 18 ? 18 : usbl);
    *usbl_out = usbl;
  }
  return PLAN_OK;
}

static inline const char *plan_message(int code) {
  switch (code) {
    case PLAN_EMPTY_TOP: return "most frequent symbol would be empty";
    case PLAN_SUM: return "probabilities do not sum to the precision";
    case PLAN_UNIQUE: return "more than 2^18 unique symbols";
    case PLAN_PROB: return "probability too large";
    default: return "symbol plan failed";
  }
}

}  // namespace plan
}  // namespace dsa
