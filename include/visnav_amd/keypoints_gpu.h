// keypoints_gpu.h -- header-only C++ shim with the reference's exact signature.
//
// Drop-in for the definition at /root/reference/include/visnav/keypoints.h:247-361:
//
//   void visnav::matchDescriptors(
//       const std::vector<std::bitset<256>>& corner_descriptors_1,
//       const std::vector<std::bitset<256>>& corner_descriptors_2,
//       std::vector<std::pair<int, int>>& matches, int threshold, double dist_2_best);
//
// Same argument meaning and output (matches is cleared, then filled in the
// reference's order); the work runs on an MI355X through the C ABI of
// libvsmatch.so (include/vsmatch.h).  The reference function has no error path;
// this one throws std::runtime_error when the GPU path fails -- there is no CPU
// fallback.  Re-entrant: one library context per calling thread, as the
// tbb::parallel_for callers need (src/sfm.cpp:1295-1321, 1397-1423).
//
// The device is taken from the environment variable VSM_DEVICE (default 0).  matchDescriptorsBatch
// uses every GPU listed in VSM_DEVICES (comma separated, e.g. "0,1,2,3,4,5,6,7") when it is set.
#pragma once

#include <algorithm>
#include <bitset>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vsmatch.h"

namespace visnav {
namespace vsm_detail {

static_assert(sizeof(std::bitset<256>) == 32, "std::bitset<256> must be four packed 64-bit words");
static_assert(sizeof(std::pair<int, int>) == 8, "std::pair<int,int> must be two packed int32");

struct ThreadContext {
  vsm_ctx* ctx = nullptr;
  ~ThreadContext() {
    if (ctx) vsm_destroy(ctx);
  }
};

inline vsm_ctx* thread_context() {
  thread_local ThreadContext t;
  if (!t.ctx) {
    // bit k of the bitset must be bit k%64 of word k/64 (libstdc++, libc++)
    std::bitset<256> probe;
    probe.set(0);
    probe.set(255);
    const uint64_t* w = reinterpret_cast<const uint64_t*>(&probe);
    if (w[0] != 1ull || w[3] != (1ull << 63))
      throw std::runtime_error("visnav::matchDescriptors (GPU): unexpected std::bitset<256> layout");
    const char* dev = std::getenv("VSM_DEVICE");
    if (vsm_create(dev ? std::atoi(dev) : 0, &t.ctx) != VSM_OK)
      throw std::runtime_error(std::string("visnav::matchDescriptors (GPU): ") + vsm_last_error());
  }
  return t.ctx;
}

// One context per GPU of VSM_DEVICES, created on first use, shared by the process.
// The pool has NO destructor and is never freed at exit: a function-local static built before the
// process's first HIP call is destroyed AFTER the HIP runtime's own exit handlers have run, and
// vsm_destroy (hipStreamSynchronize / hipFree / hipStreamDestroy) on a runtime that is already torn down
// aborts (seen as a failed `dev_runtime_unloaded` check under __cxa_finalize of libamdhip64 in the two-context
// sanitizer run of tools/asan_run.sh).  The driver reclaims the contexts' device memory with the process;
// a caller that wants them gone earlier calls visnav::vsm_detail::shutdown_device_pool() before exit.
struct DevicePool {
  std::vector<vsm_ctx*> ctxs;
  std::mutex mu;
  bool ready = false;
};

inline DevicePool& device_pool() {
  static DevicePool* pool = new DevicePool;  // leaked on purpose, see above
  std::lock_guard<std::mutex> lk(pool->mu);
  if (pool->ready) return *pool;
  const char* list = std::getenv("VSM_DEVICES");
  if (list && *list) {
    thread_context();  // the layout probe
    std::vector<vsm_ctx*> made;  // swapped in only when every context exists: a failure leaves no partial list behind
    for (const char* p = list; *p;) {
      char* end = nullptr;
      const long dev = std::strtol(p, &end, 10);
      if (end == p) break;
      vsm_ctx* c = nullptr;
      if (vsm_create(int(dev), &c) != VSM_OK) {
        const std::string why = vsm_last_error();
        for (vsm_ctx* m : made) vsm_destroy(m);
        throw std::runtime_error("visnav::matchDescriptorsBatch (VSM_DEVICES): " + why);  // not ready: the next call tries again
      }
      made.push_back(c);
      p = *end == ',' ? end + 1 : end;
    }
    pool->ctxs.swap(made);
  }
  pool->ready = true;
  return *pool;
}

// Destroys the pool's contexts while the HIP runtime is still up (optional; call it before main returns).
inline void shutdown_device_pool() {
  DevicePool& pool = device_pool();
  std::lock_guard<std::mutex> lk(pool.mu);
  for (vsm_ctx* c : pool.ctxs) vsm_destroy(c);
  pool.ctxs.clear();
  pool.ready = false;
}

// Image pairs per library call of matchDescriptorsBatch (VSM_BATCH_PAIRS overrides it: a test hook).
inline size_t batch_chunk_pairs() {
  const char* e = std::getenv("VSM_BATCH_PAIRS");
  const long v = e ? std::atol(e) : 0;
  return v > 0 ? size_t(v) : size_t(262144);
}

}  // namespace vsm_detail

inline void matchDescriptors(const std::vector<std::bitset<256>>& corner_descriptors_1,
                             const std::vector<std::bitset<256>>& corner_descriptors_2,
                             std::vector<std::pair<int, int>>& matches, int threshold,
                             double dist_2_best) {
  matches.clear();  // keypoints.h:251
  const size_t n1 = corner_descriptors_1.size(), n2 = corner_descriptors_2.size();
  if (n1 == 0 || n2 == 0) return;
  const size_t cap = n1 < n2 ? n1 : n2;  // the cross-check emits at most min(n1, n2) pairs
  matches.resize(cap);
  size_t count = 0;
  const int rc = vsm_match_pair(
      vsm_detail::thread_context(), reinterpret_cast<const uint64_t*>(corner_descriptors_1.data()), n1,
      reinterpret_cast<const uint64_t*>(corner_descriptors_2.data()), n2, threshold, dist_2_best,
      reinterpret_cast<int32_t*>(matches.data()), cap, &count);
  if (rc != VSM_OK) {
    matches.clear();
    throw std::runtime_error(std::string("visnav::matchDescriptors (GPU): ") + vsm_last_error());
  }
  matches.resize(count);
}

// The loop of match_all / match_bow (src/sfm.cpp:1295-1321, 1397-1423) as one
// batch: out[k] = matchDescriptors(*sets[ids_to_match[k].first],
// *sets[ids_to_match[k].second], ...).  The descriptor sets are uploaded once
// and every pair is matched on the GPU in one call.
inline void matchDescriptorsBatch(const std::vector<const std::vector<std::bitset<256>>*>& sets,
                                  const std::vector<std::pair<int, int>>& ids_to_match,
                                  std::vector<std::vector<std::pair<int, int>>>& out, int threshold,
                                  double dist_2_best) {
  out.clear();
  out.resize(ids_to_match.size());
  std::vector<uint32_t> offsets(sets.size() + 1, 0);
  for (size_t i = 0; i < sets.size(); i++) offsets[i + 1] = offsets[i] + uint32_t(sets[i]->size());
  std::vector<uint64_t> flat(size_t(offsets.back()) * 4);
  for (size_t i = 0; i < sets.size(); i++)
    if (!sets[i]->empty())
      std::copy(reinterpret_cast<const uint64_t*>(sets[i]->data()),
                reinterpret_cast<const uint64_t*>(sets[i]->data()) + sets[i]->size() * 4,
                flat.begin() + size_t(offsets[i]) * 4);
  std::vector<uint32_t> pairs(ids_to_match.size() * 2);
  for (size_t k = 0; k < ids_to_match.size(); k++) {
    const int a = ids_to_match[k].first, b = ids_to_match[k].second;
    if (a < 0 || b < 0 || size_t(a) >= sets.size() || size_t(b) >= sets.size())
      throw std::runtime_error("visnav::matchDescriptorsBatch: image index out of range");
    pairs[2 * k] = uint32_t(a);
    pairs[2 * k + 1] = uint32_t(b);
  }
  vsm_detail::DevicePool& pool = vsm_detail::device_pool();
  const bool multi = !pool.ctxs.empty();
  // the pool serves one batch at a time (database replicated device to device, pairs dealt in blocks, one host
  // thread per GPU inside the library: vsm_match_pairs_multi)
  std::unique_lock<std::mutex> lk(pool.mu, std::defer_lock);
  vsm_ctx* ctx = nullptr;
  if (multi) {
    lk.lock();
    if (vsm_db_upload_multi(pool.ctxs.data(), pool.ctxs.size(), flat.data(), offsets.data(), sets.size()) != VSM_OK)
      throw std::runtime_error(std::string("visnav::matchDescriptorsBatch (GPUs of VSM_DEVICES): ") + vsm_last_error());
  } else {
    ctx = vsm_detail::thread_context();
    if (vsm_db_upload(ctx, flat.data(), offsets.data(), sets.size()) != VSM_OK)
      throw std::runtime_error(std::string("visnav::matchDescriptorsBatch (GPU): ") + vsm_last_error());
  }
  std::vector<uint64_t>().swap(flat);  // the database is on the device(s) now
  // The pair list goes through in chunks, each with a list buffer sized by what real data needs (32 matches per
  // pair to begin with) instead of the worst case min(n1, n2) per pair -- 16 KB per pair of 2000-descriptor images,
  // 200 GB for the 12.5 M pairs of a 5000-image match_all.  A chunk that needs more says how much
  // (VSM_ERR_CAPACITY returns the exact total) and runs once more with that.  Host memory: the chunk buffer
  // (<= max(64 K, what a chunk really holds) x 8 B) + the lists handed back in `out`; device memory likewise.
  const size_t chunk_pairs = vsm_detail::batch_chunk_pairs();
  std::vector<uint64_t> offs;
  std::vector<std::pair<int, int>> all;
  for (size_t k0 = 0; k0 < ids_to_match.size(); k0 += chunk_pairs) {
    const size_t n = std::min(chunk_pairs, ids_to_match.size() - k0);
    size_t bound = 0;  // the cross-check emits at most min(n1, n2) pairs per image pair
    for (size_t k = k0; k < k0 + n; k++) bound += std::min(sets[pairs[2 * k]]->size(), sets[pairs[2 * k + 1]]->size());
    size_t cap = std::min(bound, std::max<size_t>(size_t(1) << 16, 32 * n));
    offs.resize(n + 1);
    for (int attempt = 0;; attempt++) {
      if (all.size() < std::max<size_t>(cap, 1)) all.resize(std::max<size_t>(cap, 1));
      size_t total = 0;
      const int rc = multi ? vsm_match_pairs_multi(pool.ctxs.data(), pool.ctxs.size(), pairs.data() + 2 * k0, n, threshold, dist_2_best,
                                                   offs.data(), reinterpret_cast<int32_t*>(all.data()), cap, &total)
                           : vsm_match_pairs(ctx, pairs.data() + 2 * k0, n, threshold, dist_2_best, offs.data(),
                                             reinterpret_cast<int32_t*>(all.data()), cap, &total);
      if (rc == VSM_OK) break;
      if (rc != VSM_ERR_CAPACITY || attempt > 0 || total <= cap)
        throw std::runtime_error(std::string("visnav::matchDescriptorsBatch (GPU): ") + vsm_last_error());
      cap = total;  // exact: the second attempt fits
    }
    for (size_t k = 0; k < n; k++) out[k0 + k].assign(all.begin() + offs[k], all.begin() + offs[k + 1]);
  }
}

}  // namespace visnav
