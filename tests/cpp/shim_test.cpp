// shim_test.cpp -- the reference's DescriptorMatching test (test_ex3.cpp:148-185)
// against the C++ shim: std::bitset vectors in, std::pair vector out, plus the
// concurrent-callers pattern of sfm.cpp:1295 and the batched helper.
// usage: shim_test <kd0.npy> <kd1.npy> <matches.npy>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>

#include "visnav_amd/keypoints_gpu.h"

static std::vector<char> npy_payload(const char* path) {
  std::ifstream f(path, std::ios::binary);
  std::vector<char> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (all.size() < 10 || std::memcmp(all.data(), "\x93NUMPY", 6) != 0) {
    std::fprintf(stderr, "bad npy file %s\n", path);
    std::exit(2);
  }
  const size_t hlen = (unsigned char)all[8] | ((unsigned char)all[9] << 8);
  return std::vector<char>(all.begin() + 10 + hlen, all.end());
}

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  auto b0 = npy_payload(argv[1]), b1 = npy_payload(argv[2]), bm = npy_payload(argv[3]);
  std::vector<std::bitset<256>> kd0(b0.size() / 32), kd1(b1.size() / 32);
  std::memcpy((void*)kd0.data(), b0.data(), kd0.size() * 32);
  std::memcpy((void*)kd1.data(), b1.data(), kd1.size() * 32);
  std::vector<std::pair<int, int>> want(bm.size() / 8);
  std::memcpy((void*)want.data(), bm.data(), want.size() * 8);

  std::vector<std::pair<int, int>> matches = {{-1, -1}};
  visnav::matchDescriptors(kd0, kd1, matches, 70, 1.2);
  if (matches != want) {
    std::fprintf(stderr, "FAIL: %zu matches, want %zu (or order differs)\n", matches.size(), want.size());
    return 1;
  }
  // empty inputs clear the output (keypoints.h:251)
  std::vector<std::bitset<256>> none;
  visnav::matchDescriptors(kd0, none, matches, 70, 1.2);
  if (!matches.empty()) return 1;

  // concurrent callers, distinct outputs, shared read-only inputs (sfm.cpp:1295-1321)
  std::vector<std::vector<std::pair<int, int>>> res(8);
  std::vector<std::thread> th;
  for (int t = 0; t < 8; t++)
    th.emplace_back([&, t] {
      for (int rep = 0; rep < 5; rep++)
        if (t % 2) visnav::matchDescriptors(kd0, kd1, res[t], 70, 1.2);
        else visnav::matchDescriptors(kd1, kd0, res[t], 70, 1.2);
    });
  for (auto& x : th) x.join();
  for (int t = 0; t < 8; t++) {
    if (t % 2) {
      if (res[t] != want) { std::fprintf(stderr, "FAIL: thread %d\n", t); return 1; }
    } else {
      auto sw = res[t];
      for (auto& p : sw) std::swap(p.first, p.second);
      std::sort(sw.begin(), sw.end());
      if (sw != want) { std::fprintf(stderr, "FAIL: thread %d (swapped)\n", t); return 1; }
    }
  }
  // batched helper
  std::vector<std::vector<std::pair<int, int>>> out;
  visnav::matchDescriptorsBatch({&kd0, &kd1, &none}, {{0, 1}, {1, 0}, {0, 2}, {0, 1}}, out, 70, 1.2);
  if (out.size() != 4 || out[0] != want || out[3] != want || !out[2].empty() || out[1].size() != want.size()) {
    std::fprintf(stderr, "FAIL: batch\n");
    return 1;
  }
  // (with VSM_DEVICES set the batch above ran through vsm_match_pairs_multi: the test runs this
  // binary a second time with VSM_DEVICES=0,0 -- two contexts on the one GPU)
  std::printf("shim ok: %zu matches, 8 concurrent callers, batch of 4%s\n", want.size(),
              std::getenv("VSM_DEVICES") ? " over the contexts of VSM_DEVICES" : "");
  return 0;
}
