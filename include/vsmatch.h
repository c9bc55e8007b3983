/*
 * vsmatch.h -- C ABI of libvsmatch.so, the MI355X (gfx950) drop-in for the
 * brute-force ORB descriptor matcher of berkaygumus/visual-SLAM:
 *
 *     void visnav::matchDescriptors(
 *         const std::vector<std::bitset<256>>& corner_descriptors_1,
 *         const std::vector<std::bitset<256>>& corner_descriptors_2,
 *         std::vector<std::pair<int, int>>& matches,
 *         int threshold, double dist_2_best);      // include/visnav/keypoints.h:247-250
 *
 * Plain pointers and sizes only.  The C++ shim with the reference's exact
 * signature is include/visnav_amd/keypoints_gpu.h; the reference-side bindings
 * are shown in INTEGRATION.md.
 *
 * Descriptor layout everywhere: one descriptor = 4 little-endian uint64 words
 * (32 bytes); bit k of the reference's std::bitset<256> (keypoints.h:53) is bit
 * k % 64 of word k / 64 -- the in-memory image of the bitset on libstdc++ and
 * libc++ x86-64, so `reinterpret_cast<const uint64_t*>(vec.data())` is a
 * zero-copy view of `KeypointsData::corner_descriptors` (common_types.h:121).
 *
 * Match lists are int32 pairs (first, second) = (index into set 1, index into
 * set 2), the layout of `std::vector<std::pair<int,int>>`
 * (`MatchData::matches`, common_types.h:133), in the reference's order
 * (ascending `first`, keypoints.h:354-360).
 *
 * Every function returns VSM_OK (0) or an error code; vsm_last_error() gives
 * the message of the calling thread's last failure.  There is no CPU fallback:
 * without a usable GPU every entry point fails with VSM_ERR_HIP.
 *
 * Threading: a vsm_ctx serialises its calls with an internal mutex; use one
 * context per thread for concurrency (the C++ shim keeps a thread_local one,
 * which makes it re-entrant for the tbb::parallel_for of src/sfm.cpp:1295).
 *
 * Stream ordering: all work of a context runs on its own non-blocking HIP stream
 * (vsm_stream()), which has NO implicit ordering with the null stream, with another library's
 * stream or with a collective's stream.  Device buffers handed to vsm_db_attach, vsm_nn2_device,
 * vsm_match_pairs_device or vsm_match_pair_noalloc must be complete before the call: synchronise
 * the producing stream (or the device) first, or make vsm_stream() wait on an event the
 * producer recorded (hipStreamWaitEvent).  Likewise wait for vsm_stream() before another stream
 * reads the outputs of the entry points that return before their work has run.
 */
#ifndef VSMATCH_H
#define VSMATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsm_ctx vsm_ctx;

enum {
  VSM_OK = 0,
  VSM_ERR_INVALID = 1,  /* bad argument */
  VSM_ERR_HIP = 2,      /* a HIP runtime call failed / no device */
  VSM_ERR_CAPACITY = 3, /* caller's output buffer too small; *out_count / *out_total holds the need */
  VSM_ERR_NOMEM = 4
};

/* Largest descriptor set (one side of a match, or one image of a database)
 * the kernels index: 2^23 = 8,388,608 descriptors (23-bit index field of the
 * packed (distance, index) key). */
#define VSM_MAX_SET ((size_t)1 << 23)

int vsm_version(void);
const char* vsm_last_error(void);

/* Context = one HIP device + one stream + device scratch owned by the library. */
int vsm_create(int device, vsm_ctx** out);
void vsm_destroy(vsm_ctx* ctx);
/* The HIP stream (hipStream_t) all work of this context is enqueued on. */
void* vsm_stream(vsm_ctx* ctx);

/*
 * Replaces visnav::matchDescriptors, keypoints.h:247-361 (called at
 * src/sfm.cpp:1243, src/odometry.cpp:814, src/point_cloud_localization.cpp:873,
 * test/src/test_ex3.cpp:164).  d1/d2: host pointers to n1/n2 packed
 * descriptors, borrowed for the call.  out_pairs: host buffer for cap_pairs
 * (first, second) pairs; min(n1, n2) always suffices.  *out_count = number of
 * matches.  VSM_ERR_CAPACITY if it exceeds cap_pairs (nothing is written then).
 */
int vsm_match_pair(vsm_ctx* ctx, const uint64_t* d1, size_t n1, const uint64_t* d2,
                   size_t n2, int threshold, double dist_2_best, int32_t* out_pairs,
                   size_t cap_pairs, size_t* out_count);

/*
 * vsm_match_pair for callers that cannot allocate -- code under HIP graph capture, tight serving
 * loops (one call per odometry keyframe, src/odometry.cpp:814), harnesses that forbid hipMalloc.
 * It has no counterpart in the reference.  Same result as vsm_match_pair, but: every pointer is a
 * DEVICE pointer (d_d1[n1][4], d_d2[n2][4] uint64 in; d_out_pairs[cap_pairs][2] int32 and
 * d_out_count[1] uint64 out); it only enqueues kernels on vsm_stream() and returns -- the caller
 * synchronises the stream before reading the outputs; it calls no hipMalloc / hipFree / hipMemcpy
 * and creates no events (vsm_set_profiling is ignored for it).  Its scratch (two jobs and the
 * partial scan results of the train splits) is a static device array of 4 MiB + 256 B per GPU:
 *   - at most 8192 descriptors per side; larger sets (or a knob-forced split count whose results do
 *     not fit) fail with VSM_ERR_INVALID and a message, nothing is enqueued;
 *   - every call on a GPU, from any context, uses that GPU's one array: calls that overlap on one GPU
 *     race.  Calls on the same context are ordered by its stream; with several contexts on one GPU
 *     wait for one call to finish before starting the next.  Nothing in the array has to be zero
 *     (every word that is read was written by the same call).
 * *d_out_count = number of matches, which may exceed cap_pairs: then only the first cap_pairs are
 * stored (min(n1, n2) always suffices).  No VSM_ERR_CAPACITY: the host does not see the count.
 */
int vsm_match_pair_noalloc(vsm_ctx* ctx, const void* d_d1, size_t n1, const void* d_d2, size_t n2,
                           int threshold, double dist_2_best, void* d_out_pairs, size_t cap_pairs,
                           void* d_out_count);

/*
 * One direction of the scan, keypoints.h:256-290 (== 305-339 with the roles
 * swapped), before the accept test: for each query the lowest-index nearest
 * train descriptor, its distance and the second smallest distance.  best_idx
 * is -1 and both distances 256 where nothing closer than 256 exists
 * (keypoints.h:259-262).  Host pointers; int32[nq] outputs.  (Config "1M x 1M,
 * 1-NN + ratio test" of BASELINE.json.)
 */
int vsm_nn2(vsm_ctx* ctx, const uint64_t* query, size_t nq, const uint64_t* train,
            size_t nt, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
/* Same with every pointer a DEVICE pointer (inputs already in HBM, outputs stay
 * there); returns after enqueueing, the caller synchronises vsm_stream(). */
int vsm_nn2_device(vsm_ctx* ctx, const void* d_query, size_t nq, const void* d_train,
                   size_t nt, void* d_best_idx, void* d_best_dist, void* d_second_dist);

/*
 * Descriptor database = the `feature_corners` map of src/sfm.cpp:1191-1214
 * flattened: image i owns descriptors offsets[i] .. offsets[i+1]-1 of `desc`.
 * upload copies host memory into HBM; attach borrows a DEVICE pointer (e.g. a
 * buffer an RCCL broadcast has just filled) that must outlive its use.
 * offsets is a host array of n_img + 1 entries either way.  Both calls read
 * every descriptor once (a one-byte-per-descriptor table for the batched
 * matcher is built from them): the contents of an attached buffer must be
 * complete when vsm_db_attach is called (see "stream ordering" above) and stay
 * unchanged while it is attached -- attach again after rewriting it.
 */
int vsm_db_upload(vsm_ctx* ctx, const uint64_t* desc, const uint32_t* offsets, size_t n_img);
int vsm_db_attach(vsm_ctx* ctx, const void* d_desc, const uint32_t* offsets, size_t n_img);

/*
 * The match_all / match_bow call pattern, src/sfm.cpp:1295-1321 and 1397-1423:
 * matchDescriptors(db[pairs[k][0]], db[pairs[k][1]], ...) for every k, as one
 * batch.  pairs: host uint32[n_pairs][2] image indices.  out_offsets: host
 * uint64[n_pairs + 1]; pair k's matches are out_pairs[out_offsets[k] ..
 * out_offsets[k+1]) (each list in the reference's order).  *out_total =
 * out_offsets[n_pairs].  VSM_ERR_CAPACITY if that exceeds cap_pairs
 * (out_offsets is valid then, out_pairs holds the first cap_pairs).
 */
int vsm_match_pairs(vsm_ctx* ctx, const uint32_t* pairs, size_t n_pairs, int threshold,
                    double dist_2_best, uint64_t* out_offsets, int32_t* out_pairs,
                    size_t cap_pairs, size_t* out_total);
/* Same with d_pairs, d_out_offsets, d_out_pairs DEVICE pointers; blocks until
 * the batch is done (it has to read the total back). */
int vsm_match_pairs_device(vsm_ctx* ctx, const void* d_pairs, size_t n_pairs, int threshold,
                           double dist_2_best, void* d_out_offsets, void* d_out_pairs,
                           size_t cap_pairs, size_t* out_total);

/*
 * match_all on all the GPUs of a node from ONE process (src/sfm.cpp:1276-1321 is one C++ executable):
 * `ctxs` lists one context per GPU (vsm_create(device k, ...); several contexts on one GPU work too).
 * vsm_db_upload_multi puts the database on every context: host memory to the first, device to device
 * (hipMemcpyPeer, xGMI within a node) to the others.  vsm_match_pairs_multi is vsm_match_pairs over
 * all of them: blocks of 4096 consecutive pairs are dealt round-robin to the contexts (the partition
 * of visual-slam_amd/sfm_pairs.py), one host thread per context matches its share on its GPU with
 * no communication, and the lists are merged into the caller's buffers in canonical pair order.
 * Same outputs and errors as vsm_match_pairs; the contexts are busy (locked) for the whole call.
 */
int vsm_db_upload_multi(vsm_ctx* const* ctxs, size_t n_ctx, const uint64_t* desc, const uint32_t* offsets,
                        size_t n_img);
int vsm_match_pairs_multi(vsm_ctx* const* ctxs, size_t n_ctx, const uint32_t* pairs, size_t n_pairs,
                          int threshold, double dist_2_best, uint64_t* out_offsets, int32_t* out_pairs,
                          size_t cap_pairs, size_t* out_total);

/*
 * Replaces visnav::find_matches_landmarks, include/visnav/vo_utils.h:95-187 (called on
 * every odometry frame, src/odometry.cpp:832 and 887; test/src/test_ex5.cpp:200, 248).
 * The reference's containers flattened by the caller (the C++ shim
 * include/visnav_amd/vo_utils_gpu.h does it), all HOST pointers borrowed for the call:
 *   kp_desc[nk][4], kp_xy[nk][2]   kdl.corner_descriptors, kdl.corners (x, y as double)
 *   proj_xy[np][2]                 projected_points
 *   obs_off[np + 1], obs_desc      obs_desc[obs_off[p] .. obs_off[p+1]) = descriptors
 *       feature_corners.at(fcid).corner_descriptors[fid] for every (fcid, fid) of
 *       landmarks.at(projected_track_ids[p]).obs (vo_utils.h:150-154); obs_off[0] == 0
 * out_pairs: (key_point_index, projected_point_index) int32 pairs in the reference's
 * order (ascending key_point_index); the caller maps the second through
 * projected_track_ids (vo_utils.h:183-184).  nk pairs always suffice.  The radius gate
 * `(corner - projected).norm() > match_max_dist_2d` is evaluated in double exactly as the
 * reference does.  Where the reference would index projected_track_ids[-1] (no candidate,
 * threshold > 256, dist_2_best <= 1: undefined behaviour) nothing is emitted.
 */
int vsm_match_landmarks(vsm_ctx* ctx, const uint64_t* kp_desc, const double* kp_xy, size_t nk,
                        const double* proj_xy, size_t np, const uint32_t* obs_off,
                        const uint64_t* obs_desc, double match_max_dist_2d, int threshold,
                        double dist_2_best, int32_t* out_pairs, size_t cap_pairs,
                        size_t* out_count);

/*
 * Replaces visnav::computeAngles + visnav::computeDescriptors, include/visnav/keypoints.h:
 * 151-201 and 203-237 (run by detectKeypointsAndDescriptors, keypoints.h:243-244, called at
 * src/sfm.cpp:1199 and src/odometry.cpp:804-807; test/src/test_ex3.cpp:76-146), i.e.
 * everything after corner detection.  img: HOST pointer to an 8-bit image, img(x, y) =
 * img[y * pitch + x] (pangolin::ManagedImage<uint8_t>: ptr, w, h, pitch); corners_xy[n][2]:
 * kd.corners (x, y as double).  angles_out[n] (may be NULL) = kd.corner_angles: atan2 of the
 * intensity-centroid moments, or 0 when rotate_features == 0.  desc_out[n][4] (may be NULL:
 * angles only) = kd.corner_descriptors, packed as above.  With desc_out every corner must lie
 * at least EDGE_THRESHOLD = 19 pixels inside the image, as detectKeypoints guarantees
 * (keypoints.h:145); the reference reads outside the image otherwise, this returns
 * VSM_ERR_INVALID.  atan2 / cos / sin run on the host with the caller's libm.
 */
int vsm_orb_describe(vsm_ctx* ctx, const uint8_t* img, int width, int height, size_t pitch,
                     const double* corners_xy, size_t n, int rotate_features, double* angles_out,
                     uint64_t* desc_out);
/* The two halves as the reference splits them: visnav::computeAngles (keypoints.h:151-201;
 * corners may lie anywhere, pixels outside the image are skipped) and
 * visnav::computeDescriptors with the angles given by the caller (keypoints.h:203-237;
 * border rule as above). */
int vsm_orb_angles(vsm_ctx* ctx, const uint8_t* img, int width, int height, size_t pitch,
                   const double* corners_xy, size_t n, int rotate_features, double* angles_out);
int vsm_orb_descriptors(vsm_ctx* ctx, const uint8_t* img, int width, int height, size_t pitch,
                        const double* corners_xy, const double* angles, size_t n,
                        uint64_t* desc_out);

/*
 * Replaces visnav::detectKeypoints, include/visnav/keypoints.h:133-149:
 * cv::goodFeaturesToTrack(image, points, num_features, 0.01, 8) (OpenCV, a system dependency
 * of the reference) followed by the InBounds(x, y, EDGE_THRESHOLD = 19) filter.
 * corners_out[cap][2] = kd.corners in detection order (strongest first), *out_count their
 * number; VSM_ERR_CAPACITY (with *out_count = the need) if cap is too small --
 * num_features always suffices when it is > 0 (0 = no limit, as in OpenCV).
 */
int vsm_detect_keypoints(vsm_ctx* ctx, const uint8_t* img, int width, int height, size_t pitch,
                         int num_features, double* corners_out, size_t cap, size_t* out_count);
/* The OpenCV call alone, with its parameters (blockSize 3, gradientSize 3, no mask, no
 * Harris -- the defaults the reference uses); no border filter. */
int vsm_good_features_to_track(vsm_ctx* ctx, const uint8_t* img, int width, int height, size_t pitch,
                               int max_corners, double quality_level, double min_distance,
                               double* corners_out, size_t cap, size_t* out_count);
/*
 * Replaces visnav::detectKeypointsAndDescriptors, keypoints.h:239-245 (called at
 * src/sfm.cpp:1199, src/odometry.cpp:804-807, test/src/test_ex3.cpp:82-83, 193-194):
 * detection, angles and descriptors with one image upload.  corners_out[cap][2],
 * angles_out[cap], desc_out[cap][4]; the first *out_count entries are filled.
 */
int vsm_detect_and_describe(vsm_ctx* ctx, const uint8_t* img, int width, int height, size_t pitch,
                            int num_features, int rotate_features, double* corners_out,
                            double* angles_out, uint64_t* desc_out, size_t cap, size_t* out_count);
/*
 * Replaces the loop of detect_keypoints(), src/sfm.cpp:1191-1201 (detectKeypointsAndDescriptors
 * for every image of the dataset, one after the other): all n_img images in one call.  imgs[i]:
 * HOST pointers to 8-bit images of one size (width, height, pitch).  Every device stage runs once
 * per chunk of images (thousands at EuRoC size) with the image as a grid dimension; the host
 * steps in between (the sequential minimum-distance pass of goodFeaturesToTrack; atan2 / cos / sin
 * with the caller's libm) run on up to VSM_HOST_THREADS (default 8) threads.  Results are those of
 * vsm_detect_and_describe image by image, bit for bit: out_offsets[n_img + 1] (always written),
 * image i's keypoints at [out_offsets[i], out_offsets[i+1]) of corners_out[cap][2], angles_out[cap],
 * desc_out[cap][4] (each may be NULL: not wanted).  VSM_ERR_CAPACITY if cap is too small for the
 * wanted arrays (out_offsets then still holds the counts, no database is kept; n_img * num_features
 * always suffices).
 * keep_as_database != 0 (needs num_features > 0): the descriptors also stay on the device as the
 * context's database, laid out by out_offsets as after vsm_db_upload, ready for vsm_match_pairs --
 * images in, match lists out, descriptors never leave the GPU.
 * Environment: VSM_HOST_THREADS (1..64, default 8); VSM_TRACE_BATCH (set: stage times to stderr);
 * VSM_BATCH_CHUNK (images per chunk, a test hook: the default fills 8 GB of work areas).
 */
int vsm_detect_and_describe_batch(vsm_ctx* ctx, const uint8_t* const* imgs, size_t n_img, int width,
                                  int height, size_t pitch, int num_features, int rotate_features,
                                  uint32_t* out_offsets, double* corners_out, double* angles_out,
                                  uint64_t* desc_out, size_t cap, int keep_as_database);

/*
 * Vocabulary tree of visnav::BowVocabulary (include/visnav/bow_voc.h:163-247), flattened:
 * node n has children child_idx[child_off[n] .. child_off[n+1]) (Node::children), centroid
 * node_desc[n][4] (Node::descriptor), Node::id node_id[n] and, as a leaf, word_id[n] /
 * weight[n] (after createWords(), bow_voc.h:219-232).  Root = node 0.  HOST pointers, copied.
 */
int vsm_voc_upload(vsm_ctx* ctx, const uint64_t* node_desc, const uint32_t* child_off,
                   const uint32_t* child_idx, const uint32_t* node_id, const uint32_t* word_id,
                   const double* weight, size_t n_nodes);
/*
 * Replaces the per-feature loop of BowVocabulary::transform over transformFeatureToWord
 * (bow_voc.h:57-92, 112-113; called from match_bow, src/sfm.cpp:1379): word id and weight of
 * the leaf each descriptor descends to (children compared by Hamming distance, the first of
 * equally close children wins).  The BoW vector itself (accumulation in feature order, L1
 * normalisation, bow_voc.h:114-130) is a few hundred additions and stays on the host.
 * VSM_ERR_INVALID where the reference's loop would not terminate (a node none of whose children
 * is closer than 256).  Parity unpinned: the reference's vocabulary file is not in its checkout.
 */
int vsm_bow_words(vsm_ctx* ctx, const uint64_t* desc, size_t n, uint32_t* word_out, double* weight_out);

/*
 * visnav::BowDatabase (include/visnav/bow_db.h:45-168) for a whole image set.
 *
 * vsm_bowdb_upload replaces clear() + insert() (bow_db.h:49-61, 133) of images 0 .. n_img-1 IN THIS
 * ORDER: image i's BoW vector is (words[k], values[k]), k in [vec_off[i], vec_off[i+1]), as
 * BowVocabulary::transform returns it (a word at most once per image, values L1-normalised; word
 * ids < 2^26, values finite, at most 2^24 images: VSM_ERR_INVALID otherwise).  An image is known by its index; the caller
 * keeps the FrameCamId of each.  HOST pointers, copied.
 *
 * vsm_bowdb_query replaces query() (bow_db.h:63-131) for n_q BoW vectors at once (same layout):
 * the L1 distance 2 + sum over common words of (|q - v| - |q| - |v|) to every image sharing a word
 * with the query, accumulated in the order of the query's words exactly as the reference
 * accumulates it (doubles: same additions, same order, same result bits), and of those the
 * num_results nearest.  db_limit (may be NULL = all): query q only sees images [0, db_limit[q]) --
 * the state of the reference's database after that many insert() calls.  Results of query q:
 * out_n[q] = how many (<= num_results), out_idx / out_score [q * num_results + r] ascending by
 * score; unused entries -1 / 0.  The reference leaves the order of EQUAL scores to std::sort and its
 * hash map (its own test says so, test/src/test_ex3.cpp:523-525); here the lower image index comes
 * first, so at a tie across the cut another image of the same score may be chosen.
 *
 * vsm_bow_select_pairs replaces the pair selection of match_bow (src/sfm.cpp:1366-1392) over the
 * uploaded images: image k queries images 0 .. k-1 (query, then insert), and every result of
 * another frame id becomes the pair (k, result) -- first index the querying image, as the reference
 * passes them to matchDescriptors (sfm.cpp:1401-1411); ready for vsm_match_pairs.  Order: by k, then
 * by score.  VSM_ERR_CAPACITY with *n_pairs = the need if cap is too small.
 */
int vsm_bowdb_upload(vsm_ctx* ctx, const uint32_t* vec_off, const uint32_t* words, const double* values,
                     size_t n_img);
int vsm_bowdb_query(vsm_ctx* ctx, const uint32_t* q_off, const uint32_t* q_words, const double* q_values,
                    size_t n_q, const uint32_t* db_limit, size_t num_results, int32_t* out_idx,
                    double* out_score, uint32_t* out_n);
int vsm_bow_select_pairs(vsm_ctx* ctx, const int64_t* frame_id, size_t num_candidates,
                         uint32_t (*pairs)[2], size_t cap, size_t* n_pairs);

/*
 * Measurement hooks (bench.py): wall time in ms of the scan kernel launches of
 * the last batched/nn2 call, measured with HIP events on the context's stream,
 * how many launches that was and how many 256-bit comparisons they evaluated.
 */
int vsm_last_scan_stats(vsm_ctx* ctx, double* ms, uint64_t* launches, uint64_t* comparisons);
/* The same split by launch kind: 0 = full scans and the forward scans of the
 * batched path, 1 = its pruned backward scans (only the columns a forward-accepted
 * row points at are scanned backward; the cross-check reads nothing else). */
int vsm_last_scan_stats_kind(vsm_ctx* ctx, int kind, double* ms, uint64_t* launches,
                             uint64_t* comparisons);
/* Which scan kernel the last launch of that kind was: 0 = none yet, 1 = the popcount scan (v_xor + v_bcnt), 2 = the
 * matrix-core scan on v_mfma_f32_32x32x64_f8f6f4, 3 = on v_mfma_f32_16x16x128_f8f6f4 (csrc/mfma16_scan.h).  bench.py
 * names the kernel its roofline line is about by it. */
int vsm_last_scan_kernel(vsm_ctx* ctx, int kind);
/* Enable/disable the event timing above (off by default; costs two events per launch). */
int vsm_set_profiling(vsm_ctx* ctx, int enabled);
/* Kernel variant selector for experiments and for the parity tests, which drive every kernel shape
 * through it: 0 = default (what ships).  Bits are listed at choose_shape in csrc/vsmatch.hip. */
int vsm_set_variant(vsm_ctx* ctx, int variant);
/*
 * Host-only helper (no GPU, no context; for tests): the distance L in [1, 256] at which the
 * matrix-core scan starts its best / second best for the accept test of keypoints.h:295-296
 * with these parameters.  Replacing every best or second best distance >= L by L leaves
 * `int(best) < threshold && dist_2_best * best <= second` unchanged for all 0 <= best <= second
 * <= 256; tests/test_abi.py checks that exhaustively.
 */
int vsm_accept_limit(int threshold, double dist_2_best);

#ifdef __cplusplus
}
#endif
#endif /* VSMATCH_H */
