// shim_bowdb_test.cpp -- BowDatabaseGpu (include/visnav_amd/bow_db_gpu.h) the way match_bow uses
// BowDatabase (src/sfm.cpp:1366-1392): insert every image, select the pairs, query single vectors.
// usage: shim_bowdb_test <off.bin uint32> <words.bin uint32> <values.bin double> <want_pairs.bin int32[n][2]> <k>
// Image i gets FrameCamId (i / 2, i % 2).  Prints the results of three queries for the caller to
// compare with the oracle ("Q <query image> <frame> <cam> <score as hex float>").
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include "visnav_amd/bow_db_gpu.h"

struct FrameCamId {  // the members of visnav::FrameCamId (common_types.h:65-75)
  int64_t frame_id;
  size_t cam_id;
};

template <class T>
static std::vector<T> raw(const char* path) {
  std::ifstream f(path, std::ios::binary);
  std::vector<char> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<T> v(all.size() / sizeof(T));
  std::memcpy((void*)v.data(), all.data(), v.size() * sizeof(T));
  return v;
}

int main(int argc, char** argv) {
  if (argc != 6) return 2;
  try {
    const auto off = raw<uint32_t>(argv[1]);
    const auto words = raw<uint32_t>(argv[2]);
    const auto values = raw<double>(argv[3]);
    const auto want = raw<int32_t>(argv[4]);
    const size_t k = (size_t)std::atoi(argv[5]), n = off.size() - 1;
    using Db = visnav::BowDatabaseGpu<FrameCamId>;
    Db db;
    std::vector<Db::BowVector> bows(n);
    for (size_t i = 0; i < n; i++) {
      for (uint32_t e = off[i]; e < off[i + 1]; e++) bows[i].push_back({words[e], values[e]});
      db.insert(FrameCamId{(int64_t)(i / 2), i % 2}, bows[i]);
    }
    std::vector<std::pair<int, int>> ids;
    db.selectPairs(k, ids);
    if (ids.size() * 2 != want.size()) {
      std::fprintf(stderr, "FAIL: %zu pairs, want %zu\n", ids.size(), want.size() / 2);
      return 1;
    }
    for (size_t i = 0; i < ids.size(); i++)
      if (ids[i].first != want[2 * i] || ids[i].second != want[2 * i + 1]) {
        std::fprintf(stderr, "FAIL: pair %zu = (%d, %d), want (%d, %d)\n", i, ids[i].first, ids[i].second, want[2 * i],
                     want[2 * i + 1]);
        return 1;
      }
    for (size_t q : {(size_t)0, n / 2, n - 1}) {
      Db::BowQueryResult r;
      db.query(bows[q], k, r);
      for (const auto& e : r) std::printf("Q %zu %lld %zu %a\n", q, (long long)e.first.frame_id, e.first.cam_id, e.second);
    }
    // query() appends (bow_db.h:122); a cleared database answers nothing; insert after clear starts again
    Db::BowQueryResult r;
    r.push_back({FrameCamId{-1, 9}, 3.0});
    db.query(bows[0], 2, r);
    if (r.size() != 3 || r[0].first.frame_id != -1) {
      std::fprintf(stderr, "FAIL: query must append\n");
      return 1;
    }
    db.clear();
    r.clear();
    db.query(bows[0], 2, r);
    db.selectPairs(k, ids);
    if (!r.empty() || !ids.empty()) {
      std::fprintf(stderr, "FAIL: results from a cleared database\n");
      return 1;
    }
    db.insert(FrameCamId{42, 1}, bows[1]);
    db.query(bows[1], 2, r);
    if (r.size() != 1 || r[0].first.frame_id != 42 || r[0].first.cam_id != 1) {
      std::fprintf(stderr, "FAIL: insert after clear\n");
      return 1;
    }
    // a copy is a database of its own (the reference's BowDatabase is copyable): after different inserts into
    // the original and the copy each answers from its own contents, whichever this thread's context saw last
    Db copy = db;
    db.insert(FrameCamId{43, 0}, bows[2]);
    copy.insert(FrameCamId{77, 1}, bows[3]);
    r.clear();
    db.query(bows[2], 1, r);
    Db::BowQueryResult rc;
    copy.query(bows[3], 1, rc);
    if (r.size() != 1 || r[0].first.frame_id != 43 || rc.size() != 1 || rc[0].first.frame_id != 77) {
      std::fprintf(stderr, "FAIL: a copy and its original answer from one device database\n");
      return 1;
    }
    std::printf("bowdb shim ok: %zu images, %zu pairs\n", n, want.size() / 2);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FAIL: %s\n", e.what());
    return 1;
  }
}
