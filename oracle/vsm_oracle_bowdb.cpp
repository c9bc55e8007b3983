// oracle/vsm_oracle_bowdb.cpp -- CPU restatement of visnav::BowDatabase and of the pair selection of
// match_bow.  TEST INFRASTRUCTURE ONLY: nothing under visual-slam_amd/ or include/ may use it.
//
// Follows /root/reference/include/visnav/bow_db.h:49-61 (insert), :63-131 (query) and
// /root/reference/src/sfm.cpp:1366-1392 (the loop that turns query results into image pairs).
// C++ because the result ORDER of query() is what std::unordered_map's iteration order and std::sort
// make of it (equal scores are not ordered by the comparator, bow_db.h:114-115): the same standard
// containers, the reference's hash of a FrameCamId (common_types.h:346-355 over hash.h:37-57) and the
// same comparator give the reference's behaviour on this toolchain.  The real header does not build
// here (tbb, cereal and Eigen are absent).  Pinned by the reference's own fixtures
// (test/ex3_test_data/bow_db.json, bow_dist.cereal; tests/test_bowdb_oracle.py) to the tolerance of
// the reference's test (1e-10, test_ex3.cpp:567-576).
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct FrameCamId {  // common_types.h:65-97
  int64_t frame_id;
  uint64_t cam_id;
  bool operator==(const FrameCamId& o) const { return frame_id == o.frame_id && cam_id == o.cam_id; }
};

// hash.h:37-57 with visnav_hash<T> = std::hash<T> (identity for integers in libstdc++)
template <class T>
void hash_combine(std::size_t& seed, const T& val) {
  const std::size_t m = 0xc6a4a7935bd1e995ull;
  std::size_t hash = std::hash<T>()(val);
  hash *= m;
  hash ^= hash >> 47;
  hash *= m;
  seed ^= hash;
  seed *= m;
  seed += 0xe6546b64;
}

struct FcidHash {  // common_types.h:346-355
  std::size_t operator()(const FrameCamId& v) const noexcept {
    std::size_t seed = 0;
    hash_combine(seed, v.frame_id);
    hash_combine(seed, v.cam_id);
    return seed;
  }
};

using Entry = std::pair<FrameCamId, double>;
using Word = std::pair<uint32_t, double>;

struct Db {
  std::unordered_map<uint32_t, std::vector<Entry>> inverted_index;  // common_types.h:336-337

  void insert(const FrameCamId& fcid, const Word* bow, size_t n) {  // bow_db.h:55-60
    for (size_t k = 0; k < n; k++) inverted_index[bow[k].first].push_back(Entry(fcid, bow[k].second));
  }

  void query(const Word* bow, size_t n, size_t num_results, std::vector<Entry>& results) const {
    std::unordered_map<FrameCamId, double, FcidHash> images_dist_map;  // bow_db.h:71-72
    for (size_t k = 0; k < n; k++) {                                  // 78
      const double q = bow[k].second;
      auto it = inverted_index.find(bow[k].first);
      if (it == inverted_index.end()) continue;  // 81-83
      for (const Entry& e : it->second) {        // 84
        auto s = images_dist_map.find(e.first);
        if (s == images_dist_map.end())
          images_dist_map[e.first] = 2 + std::abs(q - e.second) - std::abs(q) - std::abs(e.second);  // 87-89
        else
          s->second += std::abs(q - e.second) - std::abs(q) - std::abs(e.second);  // 92-94
      }
    }
    std::vector<Entry> v;  // 108-112
    for (const auto& kv : images_dist_map) v.push_back(Entry(kv.first, kv.second));
    std::sort(v.begin(), v.end(), [](const Entry& a, const Entry& b) { return a.second < b.second; });  // 114-115
    size_t j = 0;  // 117-127
    for (const Entry& e : v) {
      if (j < num_results)
        results.push_back(e);
      else
        break;
      j++;
    }
  }
};

std::vector<Word> words_of(const uint32_t* words, const double* values, size_t n) {
  std::vector<Word> v(n);
  for (size_t k = 0; k < n; k++) v[k] = Word(words[k], values[k]);
  return v;
}

}  // namespace

extern "C" {

void* vsm_oracle_bowdb_new() { return new Db(); }
void vsm_oracle_bowdb_free(void* h) { delete static_cast<Db*>(h); }
void vsm_oracle_bowdb_clear(void* h) { static_cast<Db*>(h)->inverted_index.clear(); }  // bow_db.h:133

void vsm_oracle_bowdb_insert(void* h, int64_t frame, uint64_t cam, const uint32_t* words, const double* values, size_t n) {
  const std::vector<Word> v = words_of(words, values, n);
  static_cast<Db*>(h)->insert(FrameCamId{frame, cam}, v.data(), n);
}

size_t vsm_oracle_bowdb_query(void* h, const uint32_t* words, const double* values, size_t n, size_t num_results,
                              int64_t* out_frame, uint64_t* out_cam, double* out_score, size_t cap) {
  const std::vector<Word> v = words_of(words, values, n);
  std::vector<Entry> r;
  static_cast<const Db*>(h)->query(v.data(), n, num_results, r);
  for (size_t k = 0; k < r.size() && k < cap; k++) {
    out_frame[k] = r[k].first.frame_id;
    out_cam[k] = r[k].first.cam_id;
    out_score[k] = r[k].second;
  }
  return r.size();
}

// sfm.cpp:1366-1392: every image, in key order, queries the images inserted before it and is then
// inserted itself; results of another frame become pairs (this image, result).  Also returns the
// scores (for a checker that has to tell a different choice among equal scores from an error).
size_t vsm_oracle_match_bow_pairs(const int64_t* frame, const uint64_t* cam, const uint32_t* vec_off,
                                  const uint32_t* words, const double* values, size_t n_img, size_t num_candidates,
                                  int32_t* pairs, double* pair_score, size_t cap) {
  Db db;
  std::unordered_map<FrameCamId, int, FcidHash> id_to_key;
  size_t n = 0;
  for (size_t i = 0; i < n_img; i++) {
    const FrameCamId key{frame[i], cam[i]};
    id_to_key[key] = (int)i;  // 1374-1376
    const std::vector<Word> v = words_of(words + vec_off[i], values + vec_off[i], vec_off[i + 1] - vec_off[i]);
    std::vector<Entry> r;
    db.query(v.data(), v.size(), num_candidates, r);  // 1381-1382
    for (const Entry& e : r) {
      if (key.frame_id != e.first.frame_id) {  // 1386
        if (n < cap) {
          pairs[2 * n] = (int32_t)i;
          pairs[2 * n + 1] = id_to_key.at(e.first);
          pair_score[n] = e.second;
        }
        n++;
      }
    }
    db.insert(key, v.data(), v.size());  // 1391
  }
  return n;
}

}  // extern "C"
