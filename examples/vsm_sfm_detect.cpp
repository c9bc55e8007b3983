// vsm_sfm_detect.cpp -- detect_keypoints() of `sfm` (src/sfm.cpp:1191-1214) on the GPU, without the GUI.
//
//   vsm_sfm_detect <images.txt> <corners.cereal> [num_features=1500] [rotate_features=1]
//
// images.txt: one image per line, `<frame_id> <cam_id> <path to an 8-bit binary PGM (P5)>` (the
// reference loads its images through Pangolin; PGM keeps this tool free of image libraries -- convert
// with any tool, e.g. `convert x.png x.pgm`).  All images must have one size.  Every image goes through
// detectKeypointsAndDescriptors (keypoints.h:239-245; defaults = sfm.cpp:197, 199) in ONE call
// (vsm_detect_and_describe_batch) and the result is saved as the corners.cereal that sfm saves
// (sfm.cpp:1203-1211), in the order of the list.  Then: examples/vsm_sfm_match.
// Needs only include/ and libvsmatch.so:
//   g++ -std=c++14 -O2 -Iinclude examples/vsm_sfm_detect.cpp -Lvisual-slam_amd -lvsmatch -pthread -o vsm_sfm_detect
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "visnav_amd/cereal_lite.h"
#include "visnav_amd/keypoints_orb_gpu.h"

namespace {

struct Image {  // the members of pangolin::Image<uint8_t> the shims read
  size_t pitch = 0;
  const uint8_t* ptr = nullptr;
  size_t w = 0, h = 0;
  std::vector<uint8_t> pixels;
};

// 8-bit binary PGM: "P5" <width> <height> <maxval <= 255>, '#' comments, one whitespace, then the rows
Image read_pgm(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  auto token = [&]() {
    std::string t;
    for (;;) {
      const int ch = f.get();
      if (ch == EOF) break;
      if (ch == '#') {
        while (f.good() && f.get() != '\n') {
        }
        continue;
      }
      if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r') {
        if (!t.empty()) break;
        continue;
      }
      t.push_back(char(ch));
    }
    return t;
  };
  if (token() != "P5") throw std::runtime_error(path + ": not a binary PGM (P5)");
  const long w = std::atol(token().c_str()), h = std::atol(token().c_str()), maxval = std::atol(token().c_str());
  if (w <= 0 || h <= 0 || w > 32768 || h > 32768 || maxval <= 0 || maxval > 255)
    throw std::runtime_error(path + ": need an 8-bit PGM of at most 32768 x 32768");
  Image img;
  img.w = img.pitch = size_t(w);
  img.h = size_t(h);
  img.pixels.resize(size_t(w) * size_t(h));
  f.read(reinterpret_cast<char*>(img.pixels.data()), std::streamsize(img.pixels.size()));
  if (size_t(f.gcount()) != img.pixels.size()) throw std::runtime_error(path + ": truncated");
  img.ptr = img.pixels.data();
  return img;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3 || argc > 5) {
    std::fprintf(stderr, "usage: %s <images.txt> <corners.cereal> [num_features=1500] [rotate_features=1]\n", argv[0]);
    return 2;
  }
  namespace cl = visnav::cereal_lite;
  const int num_features = argc > 3 ? std::atoi(argv[3]) : 1500;  // num_features_per_image, sfm.cpp:197
  const bool rotate = argc > 4 ? std::atoi(argv[4]) != 0 : true;  // rotate_features, sfm.cpp:199
  try {
    std::ifstream list(argv[1]);
    if (!list) throw std::runtime_error(std::string("cannot open ") + argv[1]);
    std::vector<cl::FrameCamId> ids;
    std::vector<Image> images;
    std::string line;
    while (std::getline(list, line)) {
      std::istringstream ls(line);
      cl::FrameCamId id;
      std::string path;
      if (!(ls >> id.frame_id >> id.cam_id >> path)) continue;  // blank line
      ids.push_back(id);
      images.push_back(read_pgm(path));
      images.back().ptr = images.back().pixels.data();
    }
    std::vector<const Image*> set;
    for (const Image& im : images) set.push_back(&im);
    std::vector<cl::KeypointsData> kds;
    visnav::detectKeypointsAndDescriptorsBatch(set, kds, num_features, rotate);
    cl::Corners corners;
    size_t total = 0;
    for (size_t i = 0; i < ids.size(); i++) {
      total += kds[i].corners.size();
      corners.push_back({ids[i], std::move(kds[i])});
    }
    cl::write_corners(argv[2], corners);
    std::printf("vsm_sfm_detect: %zu images, %zu keypoints -> %s\n", ids.size(), total, argv[2]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vsm_sfm_detect: %s\n", e.what());
    return 1;
  }
  return 0;
}
