// vsm_sfm_match.cpp -- the matching steps of `sfm` on its cached files, on the GPU, without the GUI.
//
//   vsm_sfm_match <corners.cereal> <matches.cereal> [all|stereo] [threshold=70] [dist_2_best=1.2]
//   vsm_sfm_match <corners.cereal> <matches.cereal> bow <ORBvoc.cereal> [num_bow_candidates=25] [threshold] [dist_2_best]
//
// Reads the corners.cereal that detect_keypoints() saved (src/sfm.cpp:1203-1211), matches
//   all:    every pair of images of different frames -- the pair list of match_all(), sfm.cpp:1285-1290,
//           (i, j) for j < i over the images in file order -- as ONE batch (the loop at 1295-1321);
//   stereo: camera 0 against camera 1 of every frame (match_stereo(), sfm.cpp:1234-1255);
//   bow:    the candidate pairs of match_bow(), sfm.cpp:1357-1423 -- every image's BoW vector through the
//           vocabulary tree (bow_voc.h:94-131), every image against the num_bow_candidates (sfm.cpp:209)
//           best of the images before it in the file (bow_db.h:63-131), pairs of one frame skipped;
// with visnav::matchDescriptors' semantics (keypoints.h:247-361; defaults = sfm.cpp:200-202) and writes
// the matches.cereal that sfm saves (sfm.cpp:1262-1270, 1342-1350).  MatchData::matches is filled;
// T_i_j and inliers stay empty (findInliersEssential / findInliersRansac are outside this library:
// the next step of the reference reads the file and can run them).
// Needs only include/ and libvsmatch.so -- no cereal, Eigen, TBB, Pangolin or OpenCV:
//   g++ -std=c++14 -O2 -Iinclude examples/vsm_sfm_match.cpp -Lvisual-slam_amd -lvsmatch -pthread -o vsm_sfm_match
// VSM_DEVICES=0,1,... spreads the batch over several GPUs (keypoints_gpu.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "visnav_amd/bow_db_gpu.h"
#include "visnav_amd/bow_voc_gpu.h"
#include "visnav_amd/cereal_lite.h"
#include "visnav_amd/keypoints_gpu.h"

int main(int argc, char** argv) {
  namespace cl = visnav::cereal_lite;
  const bool stereo = argc > 3 && std::strcmp(argv[3], "stereo") == 0;
  const bool bow = argc > 3 && std::strcmp(argv[3], "bow") == 0;
  const int first_opt = bow ? 6 : 4;  // where threshold and ratio sit
  if (argc < 3 || argc > first_opt + 2 || (bow && argc < 5) ||
      (argc > 3 && !stereo && !bow && std::strcmp(argv[3], "all") != 0)) {
    std::fprintf(stderr,
                 "usage: %s <corners.cereal> <matches.cereal> [all|stereo] [threshold=70] [dist_2_best=1.2]\n"
                 "       %s <corners.cereal> <matches.cereal> bow <ORBvoc.cereal> [num_bow_candidates=25] [threshold] [dist_2_best]\n",
                 argv[0], argv[0]);
    return 2;
  }
  const size_t num_bow_candidates = bow && argc > 5 ? size_t(std::atol(argv[5])) : 25;     // sfm.cpp:209
  const int threshold = argc > first_opt ? std::atoi(argv[first_opt]) : 70;               // feature_match_max_dist, sfm.cpp:200
  const double ratio = argc > first_opt + 1 ? std::atof(argv[first_opt + 1]) : 1.2;       // feature_match_test_next_best, sfm.cpp:202
  try {
    const cl::Corners corners = cl::read_corners(argv[1]);
    std::vector<const std::vector<std::bitset<256>>*> sets;
    for (const auto& kv : corners) sets.push_back(&kv.second.corner_descriptors);
    std::vector<std::pair<int, int>> ids_to_match;
    if (bow) {  // sfm.cpp:1373-1392 with the query-then-insert loop as one launch over the finished database
      const visnav::BowVocabularyGpu voc(argv[4]);
      visnav::BowDatabaseGpu<cl::FrameCamId> db;
      std::vector<visnav::BowDatabaseGpu<cl::FrameCamId>::BowVector> vectors;
      voc.transformAll(sets, vectors);  // one descent call for every image
      for (size_t i = 0; i < corners.size(); i++) db.insert(corners[i].first, vectors[i]);
      db.selectPairs(num_bow_candidates, ids_to_match);
    }
    for (size_t i = 0; i < (bow ? 0 : corners.size()); i++)
      for (size_t j = 0; j < (stereo ? corners.size() : i); j++) {
        const cl::FrameCamId &a = corners[i].first, &b = corners[j].first;
        if (stereo ? (a.frame_id == b.frame_id && a.cam_id == 0 && b.cam_id == 1)  // sfm.cpp:1235
                   : (a.frame_id != b.frame_id))                                   // sfm.cpp:1288
          ids_to_match.emplace_back(int(i), int(j));
      }
    std::vector<std::vector<std::pair<int, int>>> lists;
    visnav::matchDescriptorsBatch(sets, ids_to_match, lists, threshold, ratio);
    cl::Matches out;
    size_t total = 0;
    for (size_t k = 0; k < ids_to_match.size(); k++) {
      cl::MatchData md;
      md.matches = std::move(lists[k]);
      total += md.matches.size();
      out.push_back({{corners[ids_to_match[k].first].first, corners[ids_to_match[k].second].first}, std::move(md)});
    }
    cl::write_matches(argv[2], out);
    std::printf("vsm_sfm_match: %zu images, %zu %s pairs, %zu matches -> %s\n", corners.size(), ids_to_match.size(),
                stereo ? "stereo" : bow ? "BoW candidate" : "non-stereo", total, argv[2]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vsm_sfm_match: %s\n", e.what());
    return 1;
  }
  return 0;
}
