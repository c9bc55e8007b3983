#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cstdint>
#include "vsmatch.h"
int main(int argc, char** argv) {
  const size_t n = argc > 1 ? atoi(argv[1]) : 1500;
  std::vector<uint64_t> a(n * 4), b(n * 4);
  uint64_t x = 88172645463325252ull;
  for (auto& v : a) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
  for (size_t i = 0; i < n * 4; i++) b[i] = a[i] ^ (1ull << (i % 61));
  vsm_ctx* c; if (vsm_create(0, &c)) { printf("%s\n", vsm_last_error()); return 1; }
  std::vector<int32_t> out(n * 2); size_t cnt = 0;
  for (int i = 0; i < 50; i++) vsm_match_pair(c, a.data(), n, b.data(), n, 70, 1.2, out.data(), n, &cnt);
  std::vector<double> ts;
  for (int i = 0; i < 300; i++) {
    auto t0 = std::chrono::steady_clock::now();
    vsm_match_pair(c, a.data(), n, b.data(), n, 70, 1.2, out.data(), n, &cnt);
    ts.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  std::sort(ts.begin(), ts.end());
  printf("n=%zu matches=%zu  C caller: median %.1f us  min %.1f us\n", n, cnt, ts[150], ts[0]);
  vsm_destroy(c);
}
