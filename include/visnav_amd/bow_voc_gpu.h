// bow_voc_gpu.h -- header-only C++ stand-in for visnav::BowVocabulary
// (/root/reference/include/visnav/bow_voc.h:50-248) whose tree descent runs on the GPU.
//
//   visnav::BowVocabularyGpu voc(vocab_path);             // bow_voc.h:55, reads ORBvoc.cereal
//   voc.transform(kd.corner_descriptors, bow_vector);     // bow_voc.h:94-131 (src/sfm.cpp:1379)
//   voc.transformAll(sets, bow_vectors);                  // the same for every image of a set, one GPU call
//
// `transform` has the reference's argument list and result: (word id, value) pairs in order of
// first appearance, zero-weight words skipped, L1-normalised.  The per-feature descent
// (transformFeatureToWord, bow_voc.h:57-92) is one vsm_bow_words call; the few hundred double
// additions of the BoW vector are done here in the reference's order.  Parity unpinned: the
// reference's vocabulary file is not in its checkout (see include/vsmatch.h).
#pragma once

#include <atomic>
#include <bitset>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "visnav_amd/cereal_lite.h"
#include "visnav_amd/keypoints_gpu.h"

namespace visnav {
namespace vsm_detail {
// A library context holds ONE vocabulary tree, and the shim keeps one context per thread: which
// vocabulary object (by a process-wide id, never reused) this thread's context holds.
inline uint64_t& thread_voc_id() {
  thread_local uint64_t id = 0;
  return id;
}
inline uint64_t next_voc_id() {
  static std::atomic<uint64_t> next{1};
  return next.fetch_add(1);
}
}  // namespace vsm_detail

class BowVocabularyGpu {
 public:
  using WordId = unsigned int;   // common_types.h:323
  using WordValue = double;      // common_types.h:326
  using TDescriptor = std::bitset<256>;

  explicit BowVocabularyGpu(const std::string& filename)
      : voc_(cereal_lite::read_vocabulary(filename)), id_(vsm_detail::next_voc_id()) {}
  explicit BowVocabularyGpu(cereal_lite::Vocabulary voc) : voc_(std::move(voc)), id_(vsm_detail::next_voc_id()) {}

  size_t size() const { return voc_.size(); }

  // every feature's (word, weight); bow_voc.h:57-92 for a whole image
  void transformFeaturesToWords(const std::vector<TDescriptor>& features, std::vector<WordId>& words,
                                std::vector<WordValue>& weights) const {
    words.assign(features.size(), 0);
    weights.assign(features.size(), 0);
    if (features.empty()) return;
    vsm_ctx* ctx = vsm_detail::thread_context();
    // one upload per calling thread and vocabulary: re-uploaded when another BowVocabularyGpu used
    // this thread's context in between (the state is per thread, nothing shared is written here)
    if (vsm_detail::thread_voc_id() != id_) {
      vsm_detail::thread_voc_id() = 0;
      if (vsm_voc_upload(ctx, voc_.node_desc.data(), voc_.child_off.data(), voc_.child_idx.data(),
                         voc_.node_id.data(), voc_.word_id.data(), voc_.weight.data(), voc_.size()) != VSM_OK)
        throw std::runtime_error(std::string("visnav::BowVocabularyGpu: ") + vsm_last_error());
      vsm_detail::thread_voc_id() = id_;
    }
    if (vsm_bow_words(ctx, reinterpret_cast<const uint64_t*>(features.data()), features.size(), words.data(),
                      weights.data()) != VSM_OK)
      throw std::runtime_error(std::string("visnav::BowVocabularyGpu::transform: ") + vsm_last_error());
  }

  // bow_voc.h:94-131; BowVectorT = std::vector<std::pair<WordId, WordValue>>
  template <class BowVectorT>
  void transform(const std::vector<TDescriptor>& features, BowVectorT& v) const {
    v.clear();
    if (voc_.size() == 0) return;
    std::vector<WordId> words;
    std::vector<WordValue> weights;
    transformFeaturesToWords(features, words, weights);
    accumulate(words.data(), weights.data(), words.size(), v);
  }

  // transform() for every image of a set with ONE descent call (the loop of match_bow, sfm.cpp:1373-1379,
  // walks the tree image by image): out[i] = the BoW vector of *sets[i].
  template <class BowVectorT>
  void transformAll(const std::vector<const std::vector<TDescriptor>*>& sets, std::vector<BowVectorT>& out) const {
    out.assign(sets.size(), BowVectorT());
    if (voc_.size() == 0) return;
    std::vector<TDescriptor> all;
    for (const auto* s : sets) all.insert(all.end(), s->begin(), s->end());
    std::vector<WordId> words;
    std::vector<WordValue> weights;
    transformFeaturesToWords(all, words, weights);
    size_t at = 0;
    for (size_t i = 0; i < sets.size(); i++) {
      accumulate(words.data() + at, weights.data() + at, sets[i]->size(), out[i]);
      at += sets[i]->size();
    }
  }

 private:
  // bow_voc.h:107-130: words in order of first appearance, zero-weight words skipped, L1-normalised
  template <class BowVectorT>
  static void accumulate(const WordId* words, const WordValue* weights, size_t n, BowVectorT& v) {
    WordValue sum_weight = 0;
    std::unordered_map<WordId, size_t> slot;  // position of a word in v (the reference searches linearly)
    for (size_t i = 0; i < n; i++) {
      if (weights[i] == 0) continue;  // bow_voc.h:114
      sum_weight += weights[i];
      auto it = slot.find(words[i]);
      if (it != slot.end()) {
        v[it->second].second += weights[i];
      } else {
        slot.emplace(words[i], v.size());
        v.push_back({words[i], weights[i]});
      }
    }
    for (auto& p : v) p.second /= sum_weight;
  }

  cereal_lite::Vocabulary voc_;
  uint64_t id_;
};

}  // namespace visnav
