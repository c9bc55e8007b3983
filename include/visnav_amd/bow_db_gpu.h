// bow_db_gpu.h -- header-only C++ stand-in for visnav::BowDatabase
// (/root/reference/include/visnav/bow_db.h:45-168) whose queries run on the GPU.
//
//   visnav::BowDatabaseGpu<FrameCamId> bow_db;
//   bow_db.insert(fcid, bow_vector);                    // bow_db.h:49-61
//   bow_db.query(bow_vector, num_bow_candidates, r);    // bow_db.h:63-131: appends to r
//   bow_db.clear();                                     // bow_db.h:133
//
// and, for the loop of match_bow (src/sfm.cpp:1373-1392: every image queries the images before it
// and is then inserted), after inserting ALL images:
//
//   bow_db.selectPairs(num_bow_candidates, ids_to_match);   // one launch instead of n queries
//
// FrameCamIdT: any type with `frame_id` and `cam_id` members (visnav::FrameCamId,
// common_types.h:65-97).  Scores carry the reference's bits (same additions in the same order);
// among EQUAL scores the image inserted first comes first, where the reference leaves the order to
// std::sort and its hash map (test/src/test_ex3.cpp:523-525).  insert() only records the vector on
// the host; the first query after an insert uploads the whole database to the calling thread's
// library context (so alternating insert and query costs an upload each time: use selectPairs).
// save()/load() (bow_db.h:135-160) are cereal JSON and stay with the reference's class;
// visual-slam_amd/cereal_io.py reads and writes the same archive.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "visnav_amd/keypoints_gpu.h"

namespace visnav {
namespace vsm_detail {
// which database state (object id, version) this thread's context holds
inline std::pair<uint64_t, uint64_t>& thread_bowdb_state() {
  thread_local std::pair<uint64_t, uint64_t> st{0, 0};
  return st;
}
inline uint64_t next_bowdb_id() {
  static std::atomic<uint64_t> next{1};
  return next.fetch_add(1);
}
}  // namespace vsm_detail

template <class FrameCamIdT>
class BowDatabaseGpu {
 public:
  using WordId = unsigned int;  // common_types.h:323
  using WordValue = double;     // common_types.h:326
  using BowVector = std::vector<std::pair<WordId, WordValue>>;          // common_types.h:330
  using BowQueryResult = std::vector<std::pair<FrameCamIdT, double>>;  // common_types.h:333

  BowDatabaseGpu() : id_(vsm_detail::next_bowdb_id()) {}
  // The reference's BowDatabase is freely copyable.  A copy is a database of its own from then on: it gets a
  // fresh id, so that the (id, version) a thread's context remembers can never name two objects whose contents
  // have since diverged (both would otherwise be queried against whichever was uploaded first).
  BowDatabaseGpu(const BowDatabaseGpu& o)
      : keys_(o.keys_), off_(o.off_), words_(o.words_), values_(o.values_), id_(vsm_detail::next_bowdb_id()) {}
  BowDatabaseGpu(BowDatabaseGpu&& o) noexcept
      : keys_(std::move(o.keys_)), off_(std::move(o.off_)), words_(std::move(o.words_)), values_(std::move(o.values_)),
        id_(vsm_detail::next_bowdb_id()) {
    o.clear();
  }
  BowDatabaseGpu& operator=(const BowDatabaseGpu& o) {
    if (this != &o) {
      keys_ = o.keys_;
      off_ = o.off_;
      words_ = o.words_;
      values_ = o.values_;
      version_++;  // same object, new contents
    }
    return *this;
  }
  BowDatabaseGpu& operator=(BowDatabaseGpu&& o) noexcept {
    if (this != &o) {
      keys_ = std::move(o.keys_);
      off_ = std::move(o.off_);
      words_ = std::move(o.words_);
      values_ = std::move(o.values_);
      version_++;
      o.clear();
    }
    return *this;
  }

  void insert(const FrameCamIdT& fcid, const BowVector& bow_vector) {
    keys_.push_back(fcid);
    for (const auto& kv : bow_vector) {
      words_.push_back(kv.first);
      values_.push_back(kv.second);
    }
    off_.push_back((uint32_t)words_.size());
    version_++;
  }

  void clear() {
    keys_.clear();
    words_.clear();
    values_.clear();
    off_.assign(1, 0);
    version_++;
  }

  size_t size() const { return keys_.size(); }

  void query(const BowVector& bow_vector, size_t num_results, BowQueryResult& results) const {
    if (keys_.empty() || num_results == 0) return;
    vsm_ctx* ctx = uploaded();
    std::vector<WordId> w;
    std::vector<WordValue> v;
    for (const auto& kv : bow_vector) {
      w.push_back(kv.first);
      v.push_back(kv.second);
    }
    const uint32_t q_off[2] = {0, (uint32_t)w.size()};
    const size_t nr = std::min(num_results, keys_.size());
    std::vector<int32_t> idx(nr);
    std::vector<double> score(nr);
    uint32_t n = 0;
    if (vsm_bowdb_query(ctx, q_off, w.data(), v.data(), 1, nullptr, nr, idx.data(), score.data(), &n) != VSM_OK)
      throw std::runtime_error(std::string("visnav::BowDatabaseGpu::query: ") + vsm_last_error());
    for (uint32_t r = 0; r < n; r++) results.push_back({keys_[idx[r]], score[r]});
  }

  // ids_to_match of match_bow (sfm.cpp:1371-1392) for the images inserted so far, as indices into
  // keys(): (image, earlier image of another frame id), by image, then by score
  void selectPairs(size_t num_candidates, std::vector<std::pair<int, int>>& ids_to_match) const {
    ids_to_match.clear();
    if (keys_.empty() || num_candidates == 0) return;
    vsm_ctx* ctx = uploaded();
    std::vector<int64_t> frame(keys_.size());
    for (size_t i = 0; i < keys_.size(); i++) frame[i] = (int64_t)keys_[i].frame_id;
    static_assert(sizeof(std::pair<int, int>) == 8, "std::pair<int,int> must be two packed ints");
    const size_t cap = keys_.size() * std::min(num_candidates, keys_.size());
    ids_to_match.resize(cap);
    size_t n = 0;
    if (vsm_bow_select_pairs(ctx, frame.data(), num_candidates, reinterpret_cast<uint32_t(*)[2]>(ids_to_match.data()),
                             cap, &n) != VSM_OK)
      throw std::runtime_error(std::string("visnav::BowDatabaseGpu::selectPairs: ") + vsm_last_error());
    ids_to_match.resize(n);
  }

  const std::vector<FrameCamIdT>& keys() const { return keys_; }

 private:
  vsm_ctx* uploaded() const {
    vsm_ctx* ctx = vsm_detail::thread_context();
    auto& st = vsm_detail::thread_bowdb_state();
    if (st.first != id_ || st.second != version_) {
      st = {0, 0};
      if (vsm_bowdb_upload(ctx, off_.data(), words_.data(), values_.data(), keys_.size()) != VSM_OK)
        throw std::runtime_error(std::string("visnav::BowDatabaseGpu: ") + vsm_last_error());
      st = {id_, version_};
    }
    return ctx;
  }

  std::vector<FrameCamIdT> keys_;
  std::vector<uint32_t> off_{0};
  std::vector<WordId> words_;
  std::vector<WordValue> values_;
  uint64_t id_, version_ = 1;
};

}  // namespace visnav
