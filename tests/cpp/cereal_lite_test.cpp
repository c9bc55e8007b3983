// cereal_lite_test.cpp -- CPU-only check of include/visnav_amd/cereal_lite.h against the Python
// twin: reads a corners.cereal written by visual-slam_amd/cereal_io.py, re-writes it (must be
// byte-identical, the test compares the files) and writes a matches.cereal from a fixed rule.
// usage: cereal_lite_test <corners_in> <corners_out> <matches_out>
#include <cstdio>

#include "visnav_amd/cereal_lite.h"

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  namespace cl = visnav::cereal_lite;
  try {
    const cl::Corners corners = cl::read_corners(argv[1]);
    cl::write_corners(argv[2], corners);
    cl::Matches matches;
    for (size_t i = 1; i < corners.size(); i++) {
      cl::MatchData md;
      // rule the Python side repeats: k-th match = (k, popcount of descriptor k of image i-1)
      const auto& d = corners[i - 1].second.corner_descriptors;
      for (size_t k = 0; k < d.size(); k++) md.matches.push_back({int(k), int(d[k].count())});
      if (!md.matches.empty()) md.inliers.push_back(md.matches.front());
      matches.push_back({{corners[i].first, corners[i - 1].first}, md});
    }
    cl::write_matches(argv[3], matches);
    std::printf("cereal_lite ok: %zu images\n", corners.size());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FAIL: %s\n", e.what());
    return 1;
  }
  return 0;
}
