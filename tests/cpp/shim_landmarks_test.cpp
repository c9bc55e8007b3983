// shim_landmarks_test.cpp -- the reference's FindMatchesLandmarks test (test_ex5.cpp:151-220)
// against the C++ shim include/visnav_amd/vo_utils_gpu.h, with stand-ins for the reference's
// container types that have the same members (common_types.h:65-200; Eigen/TBB are absent).
// usage: shim_landmarks_test <dir with the raw arrays written by tests/test_gpu_parity.py>
#include <array>
#include <cstring>
#include <cstdio>
#include <fstream>
#include <map>
#include <set>
#include <unordered_map>

#include "visnav_amd/vo_utils_gpu.h"

namespace ref_types {  // shapes of common_types.h
using FrameId = int64_t;
using CamId = std::size_t;
using FeatureId = int;
using TrackId = int64_t;
struct FrameCamId {
  FrameId frame_id;
  CamId cam_id;
  bool operator<(const FrameCamId& o) const {
    return frame_id == o.frame_id ? cam_id < o.cam_id : frame_id < o.frame_id;
  }
};
struct KeypointsData {
  std::vector<std::array<double, 2>> corners;
  std::vector<double> corner_angles;
  std::vector<std::bitset<256>> corner_descriptors;
};
using Corners = std::map<FrameCamId, KeypointsData>;
using FeatureTrack = std::map<FrameCamId, FeatureId>;
struct Landmark {
  std::array<double, 3> p;
  FeatureTrack obs, outlier_obs;
};
using Landmarks = std::unordered_map<TrackId, Landmark>;
struct LandmarkMatchData {
  std::vector<std::pair<FeatureId, TrackId>> matches, inliers;
};
}  // namespace ref_types

template <class T>
static std::vector<T> raw(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  std::vector<char> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<T> v(all.size() / sizeof(T));
  std::memcpy((void*)v.data(), all.data(), v.size() * sizeof(T));
  return v;
}

int main(int argc, char** argv) {
  if (argc != 2) return 2;
  const std::string dir = std::string(argv[1]) + "/";
  using namespace ref_types;
  auto kp_desc = raw<std::bitset<256>>(dir + "kp_desc.bin");
  auto kp_xy = raw<std::array<double, 2>>(dir + "kp_xy.bin");
  auto proj = raw<std::array<double, 2>>(dir + "proj_xy.bin");
  auto tids = raw<TrackId>(dir + "track_ids.bin");
  auto obs_off = raw<uint32_t>(dir + "obs_off.bin");
  auto obs_desc = raw<std::bitset<256>>(dir + "obs_desc.bin");
  auto expected = raw<int64_t>(dir + "expected.bin");
  if (kp_desc.empty() || proj.size() != tids.size() || obs_off.size() != proj.size() + 1) return 2;

  // rebuild map-shaped containers: every landmark's observations live in synthetic images
  // (frame = observation slot, cam 0) whose descriptor lists hold them
  KeypointsData kdl;
  kdl.corners = kp_xy;
  kdl.corner_descriptors = kp_desc;
  Corners feature_corners;
  Landmarks landmarks;
  for (size_t p = 0; p < proj.size(); p++) {
    Landmark lm{};
    for (uint32_t o = obs_off[p]; o < obs_off[p + 1]; o++) {
      const FrameCamId fc{FrameId(o - obs_off[p]), 0};
      auto& kd = feature_corners[fc];
      lm.obs[fc] = FeatureId(kd.corner_descriptors.size());
      kd.corner_descriptors.push_back(obs_desc[o]);
    }
    landmarks[tids[p]] = lm;
  }
  LandmarkMatchData md;
  md.matches.push_back({-1, -1});
  visnav::find_matches_landmarks(kdl, landmarks, feature_corners, proj, tids, 20.0, 70, 1.2, md);
  std::set<std::pair<FeatureId, TrackId>> got(md.matches.begin(), md.matches.end()), want;
  for (size_t i = 0; i + 1 < expected.size(); i += 2) want.insert({FeatureId(expected[i]), expected[i + 1]});
  if (got != want || md.matches.size() != want.size()) {
    std::fprintf(stderr, "FAIL: %zu matches, want %zu\n", md.matches.size(), want.size());
    return 1;
  }
  bool threw = false;
  try {
    std::vector<TrackId> unknown(tids.size(), TrackId(-12345));
    visnav::find_matches_landmarks(kdl, landmarks, feature_corners, proj, unknown, 20.0, 70, 1.2, md);
  } catch (const std::out_of_range&) {
    threw = true;  // landmarks.at() as in the reference (vo_utils.h:151)
  }
  if (!threw) return 1;
  std::printf("landmark shim ok: %zu matches equal the reference's golden set\n", want.size());
  return 0;
}
