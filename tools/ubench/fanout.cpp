// Cost of hostutil::parallel_for's thread fan-out (threads are created per call): tools/ubench/fanout   (host only)
//   hipcc -O2 -std=c++17 -o tools/ubench/fanout tools/ubench/fanout.cpp
#include <chrono>
#include <cstdio>
#include <vector>
#include "../../draco-sharp_amd/csrc/dsa_host_util.h"
int main() {
  std::vector<int> sink(4096, 0);
  for (int rep = 0; rep < 3; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 200; ++k) hostutil::parallel_for(1024, [&](uint32_t i) { sink[i] += (int)i; }, 2);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("host_threads %u: 200 fan-outs of 1024 trivial items: %.2f ms each\n", hostutil::host_threads(), ms / 200);
  }
  std::vector<char> a(192u << 20, 1), b(192u << 20, 0);
  for (int rep = 0; rep < 3; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    hostutil::parallel_memcpy(b.data(), a.data(), a.size());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("parallel_memcpy of 192 MB: %.2f ms (%.1f GB/s)\n", ms, a.size() / ms / 1e6);
  }
  return sink[5] == 0;
}
