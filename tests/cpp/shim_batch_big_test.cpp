// shim_batch_big_test.cpp -- matchDescriptorsBatch (include/visnav_amd/keypoints_gpu.h) on a match_all of config-5
// SHAPE (src/sfm.cpp:1276-1321: every image pair, 2000-descriptor images) large enough that a worst-case list
// buffer (min(n1, n2) pairs per image pair) would not fit the memory bound: the batch goes through in chunks with
// list buffers sized by the data.
// usage: shim_batch_big_test <db.bin: uint64[n_img * per][4]> <n_img> <per> <sample_every> <out_sample.bin>
// prints: pairs, matches, an order-sensitive checksum over every list, peak resident set size in MB (and what it was before the
// batch call: inputs + HIP runtime + one context);
// writes the lists of every `sample_every`-th pair as int32 records [pair index, count, (first, second) * count].
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "visnav_amd/keypoints_gpu.h"

// peak resident set size so far, MB (VmHWM of /proc/self/status)
static long peak_rss_mb() {
  std::ifstream f("/proc/self/status");
  std::string line;
  while (std::getline(f, line))
    if (line.compare(0, 6, "VmHWM:") == 0) return std::atol(line.c_str() + 6) / 1024;
  return -1;
}

int main(int argc, char** argv) {
  if (argc != 6) return 2;
  const size_t n_img = (size_t)std::atol(argv[2]), per = (size_t)std::atol(argv[3]), every = (size_t)std::atol(argv[4]);
  try {
    std::vector<std::vector<std::bitset<256>>> imgs(n_img, std::vector<std::bitset<256>>(per));
    {
      std::ifstream f(argv[1], std::ios::binary);
      for (size_t i = 0; i < n_img; i++)
        if (!f.read(reinterpret_cast<char*>(imgs[i].data()), (std::streamsize)(per * 32))) {
          std::fprintf(stderr, "FAIL: short database file\n");
          return 2;
        }
    }
    std::vector<const std::vector<std::bitset<256>>*> sets;
    for (const auto& v : imgs) sets.push_back(&v);
    std::vector<std::pair<int, int>> ids;  // the enumeration of sfm.cpp:1285-1290
    for (size_t i = 0; i < n_img; i++)
      for (size_t j = 0; j < i; j++) ids.push_back({(int)i, (int)j});
    std::vector<std::vector<std::pair<int, int>>> out;
    {  // what the process holds before the batch: the inputs + the HIP runtime with the library's code objects and a context
      std::vector<std::pair<int, int>> warm;
      visnav::matchDescriptors(imgs[0], imgs[1], warm, 70, 1.2);
    }
    const long rss_before = peak_rss_mb();
    visnav::matchDescriptorsBatch(sets, ids, out, 70, 1.2);
    unsigned long long sum = 1469598103934665603ull, total = 0;
    std::ofstream smp(argv[5], std::ios::binary);
    for (size_t k = 0; k < out.size(); k++) {
      total += out[k].size();
      sum = (sum ^ (k + 1)) * 1099511628211ull;
      for (const auto& p : out[k]) {
        sum = (sum ^ (unsigned)p.first) * 1099511628211ull;
        sum = (sum ^ (unsigned)p.second) * 1099511628211ull;
      }
      if (k % every == 0) {
        const int head[2] = {(int)k, (int)out[k].size()};
        smp.write(reinterpret_cast<const char*>(head), 8);
        smp.write(reinterpret_cast<const char*>(out[k].data()), (std::streamsize)(out[k].size() * 8));
      }
    }
    std::printf("pairs %zu matches %llu checksum %016llx peak_rss_mb %ld rss_before_batch_mb %ld\n", ids.size(), total, sum, peak_rss_mb(),
                rss_before);
    visnav::vsm_detail::shutdown_device_pool();
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FAIL: %s\n", e.what());
    return 1;
  }
}
