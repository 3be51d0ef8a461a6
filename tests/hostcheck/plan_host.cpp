// tests/hostcheck/plan_host.cpp -- TEST INFRASTRUCTURE ONLY.
//
// draco-sharp_amd/csrc/dsa_symbol_plan.h (the scheme choice and table normalisation shared by the host coder and k_enc_plan)
// against a straightforward restatement with the standard library: std::log2, std::stable_sort, std::floor
// (Entropy/RAnsSymbolEncoder.cs:15-123).  Random histograms of every shape the encoder meets.
//   plan_host <seed> <cases>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../draco-sharp_amd/csrc/dsa_symbol_plan.h"

static uint64_t rs = 1;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }

static bool reference_tables(int max_bit_length, const std::vector<uint32_t> &freq, std::vector<uint32_t> &prob, int &precision_bits) {
  int p = (3 * max_bit_length) / 2;
  precision_bits = p < 12 ? 12 : (p > 20 ? 20 : p);
  const uint32_t precision = 1u << precision_bits;
  uint64_t total = 0;
  int max_valid = 0;
  for (size_t i = 0; i < freq.size(); ++i) { total += freq[i]; if (freq[i]) max_valid = (int)i; }
  const uint32_t ns = (uint32_t)max_valid + 1;
  prob.assign(ns, 0);
  const double total_d = (double)total, prec_d = (double)precision;
  int64_t total_prob = 0;
  for (uint32_t i = 0; i < ns; ++i) {
    uint32_t rp = (uint32_t)(((double)freq[i] / total_d) * prec_d + 0.5);
    if (rp == 0 && freq[i] > 0) rp = 1;
    prob[i] = rp; total_prob += rp;
  }
  if (total_prob != (int64_t)precision) {
    std::vector<int> order(ns);
    for (uint32_t i = 0; i < ns; ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return prob[a] < prob[b]; });
    if (total_prob < (int64_t)precision) prob[order.back()] += (uint32_t)(precision - total_prob);
    else {
      int64_t error = total_prob - precision;
      while (error > 0) {
        const double rel = prec_d / (double)total_prob;
        for (int j = (int)ns - 1; j >= 0; --j) {
          const int sid = order[j];
          if (prob[sid] <= 1) { if (j == (int)ns - 1) return false; break; }
          const int32_t np = (int32_t)std::floor(rel * (double)prob[sid]);
          int32_t fix = (int32_t)prob[sid] - np;
          if (fix == 0) fix = 1;
          if (fix >= (int32_t)prob[sid]) fix = (int32_t)prob[sid] - 1;
          if (fix > error) fix = (int32_t)error;
          prob[sid] -= fix; total_prob -= fix; error -= fix;
          if (total_prob == (int64_t)precision) break;
        }
      }
    }
  }
  return true;
}

int main(int argc, char **argv) {
  rs = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1) * 2654435761ull + 12345;
  const int cases = argc > 2 ? atoi(argv[2]) : 2000;
  // det_log2 against libm
  double worst = 0;
  for (int i = 0; i < 200000; ++i) {
    const double x = (i % 3 == 0) ? (double)(1 + rnd() % 1000000) / (double)(1 + rnd() % 1000000) : std::ldexp(1.0 + (double)rnd() / 4294967296.0, (int)(rnd() % 80) - 40);
    const double a = dsa::plan::det_log2(x), b = std::log2(x);
    const double err = std::fabs(a - b) / (std::fabs(b) > 1.0 ? std::fabs(b) : 1.0);
    if (err > worst) worst = err;
  }
  if (worst > 1e-13) { fprintf(stderr, "det_log2 off by %.3g\n", worst); return 1; }
  for (int it = 0; it < cases; ++it) {
    const uint32_t count = 1 + rnd() % (it % 7 == 0 ? 5000 : 300);
    std::vector<uint32_t> freq(count, 0);
    const int shape = (int)(rnd() % 5);
    for (uint32_t i = 0; i < count; ++i) {
      uint32_t f;
      if (shape == 0) f = rnd() % 50;                                        // flat
      else if (shape == 1) f = (rnd() % 100 < 70) ? 0 : 1 + rnd() % 3;         // sparse, rare symbols (many round up to 1)
      else if (shape == 2) f = (uint32_t)(100000.0 / (1.0 + i * i * 0.01));     // peaked
      else if (shape == 3) f = i == count / 2 ? 1000000 : (rnd() % 10 == 0);  // one dominant symbol
      else f = rnd() % 4 == 0 ? rnd() % 100000 : rnd() % 3;
      freq[i] = f;
    }
    uint64_t total = 0;
    for (uint32_t f : freq) total += f;
    if (total == 0) freq[rnd() % count] = 1 + rnd() % 9;
    const int mbl = 1 + (int)(rnd() % 14);
    std::vector<uint32_t> ref;
    int ref_pb = 0;
    const bool ref_ok = reference_tables(mbl, freq, ref, ref_pb);
    std::vector<uint32_t> prob(count), cum(count), order(count), tmp(count);
    int pb = 0;
    uint32_t ns = 0;
    const int rc = dsa::plan::rans_tables(mbl, freq.data(), freq.size(), prob.data(), cum.data(), order.data(), tmp.data(), &pb, &ns);
    if (!ref_ok) { if (rc != dsa::plan::PLAN_EMPTY_TOP) { fprintf(stderr, "case %d: reference refuses, core returns %d\n", it, rc); return 1; } continue; }
    if (rc != dsa::plan::PLAN_OK || pb != ref_pb || ns != ref.size()) { fprintf(stderr, "case %d: rc %d precision %d / %d symbols %u / %zu\n", it, rc, pb, ref_pb, ns, ref.size()); return 1; }
    uint32_t c = 0;
    for (uint32_t i = 0; i < ns; ++i) {
      if (prob[i] != ref[i] || cum[i] != c) { fprintf(stderr, "case %d: symbol %u prob %u / %u cum %u / %u\n", it, i, prob[i], ref[i], cum[i], c); return 1; }
      c += prob[i];
    }
  }
  printf("plan_host: det_log2 within %.2g of libm, %d tables equal\n", worst, cases);
  return 0;
}
