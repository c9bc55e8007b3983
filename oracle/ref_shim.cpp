// ref_shim.cpp -- extern "C" wrapper around the REAL reference function.
//
// TEST INFRASTRUCTURE ONLY (see oracle/vsm_oracle.c).  This file contains no
// reference code.  oracle/Makefile compiles it with -DVSM_REF_SLICE=<path>,
// where <path> is a pipe that streams lines 247-361 of
// /root/reference/include/visnav/keypoints.h (the definition of
// visnav::matchDescriptors, which needs only <bitset>/<vector>/<utility>)
// straight into the compiler; the reference text is never written into this
// repository.  The output is oracle/_ref/libvsm_ref.so (git-ignored).
//
// keypoints.h as a whole cannot be compiled here (it includes Eigen, Sophus,
// Pangolin and OpenCV, none of which are installed), so the function's own
// lines are compiled where they lie instead of the header.
#include <bitset>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#ifndef VSM_REF_SLICE
#error "build through oracle/Makefile (needs /root/reference)"
#endif

namespace visnav {
#include VSM_REF_SLICE
}  // namespace visnav

static_assert(sizeof(std::bitset<256>) == 32, "bitset<256> must be 4 packed words");

extern "C" size_t vsm_ref_match(const uint64_t* d1, size_t n1, const uint64_t* d2,
                                size_t n2, int threshold, double dist_2_best,
                                int32_t* out_pairs, size_t cap) {
  std::vector<std::bitset<256>> v1(n1), v2(n2);
  if (n1) std::memcpy(static_cast<void*>(v1.data()), d1, 32 * n1);
  if (n2) std::memcpy(static_cast<void*>(v2.data()), d2, 32 * n2);
  std::vector<std::pair<int, int>> matches;
  matches.emplace_back(-7, -7);  // the callee must clear() this (keypoints.h:251)
  visnav::matchDescriptors(v1, v2, matches, threshold, dist_2_best);
  for (size_t i = 0; i < matches.size() && i < cap; i++) {
    out_pairs[2 * i] = matches[i].first;
    out_pairs[2 * i + 1] = matches[i].second;
  }
  return matches.size();
}

extern "C" size_t vsm_ref_match_db_pair(const uint64_t* db, const uint32_t* offsets,
                                        uint32_t img1, uint32_t img2, int threshold,
                                        double dist_2_best, int32_t* out_pairs,
                                        size_t cap) {
  return vsm_ref_match(db + 4 * size_t(offsets[img1]), offsets[img1 + 1] - offsets[img1],
                       db + 4 * size_t(offsets[img2]), offsets[img2 + 1] - offsets[img2],
                       threshold, dist_2_best, out_pairs, cap);
}
