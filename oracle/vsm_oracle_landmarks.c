/*
 * vsm_oracle_landmarks.c -- CPU restatement of visnav::find_matches_landmarks
 * (/root/reference/include/visnav/vo_utils.h:95-187; called every odometry
 * frame at src/odometry.cpp:832, 887).  SURVEY.md section 8f row N1.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as vsm_oracle.c).
 *
 * Parity status: PINNED by the reference's own golden set for this function
 * (test/src/test_ex5.cpp:151-269): keypoints of image (3,0) and the landmark
 * observations from test/ex4_test_data/map_localize_camera.cereal, the
 * projected points from test/ex5_test_data/projections.cereal and the expected
 * matches from test/ex5_test_data/matches.cereal, converted to
 * tests/golden/ex5_landmarks.npz by tests/golden/make_golden_ex5.py.  The real
 * function cannot be compiled here (vo_utils.h needs Eigen, Sophus, opengv and
 * TBB, all absent), so there is no oracle/_ref for it.  Not pinned by any
 * fixture: which side of the radius gate a point lands on when its distance is
 * within one ulp of match_max_dist_2d (the gate is restated as Eigen evaluates
 * it for a fixed 2-vector: subtract, square and add without fusing, sqrt, all
 * in double).
 *
 * Flattened inputs (what a caller gathers from the reference's containers):
 *   kp_desc[nk][4], kp_xy[nk][2]      kdl.corner_descriptors / kdl.corners
 *   proj_xy[np][2]                    projected_points
 *   obs_off[np + 1], obs_desc[..][4]  for projected point p the descriptors
 *       feature_corners.at(fcid).corner_descriptors[fid] of every (fcid, fid)
 *       in landmarks.at(projected_track_ids[p]).obs (vo_utils.h:150-154)
 * Output: (key_point_index, projected_point_index) pairs, ascending in the
 * first; the caller maps the second through projected_track_ids
 * (vo_utils.h:183-184).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* built with -ffp-contract=off (oracle/Makefile); the attribute below repeats it */

static inline int ham256(const uint64_t *a, const uint64_t *b) {
  return __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
         __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
}

/* (a - b).norm() for Eigen::Vector2d, vo_utils.h:136-138 */
__attribute__((optimize("fp-contract=off"))) static double norm2d(const double *a, const double *b) {
  const double dx = a[0] - b[0], dy = a[1] - b[1];
  const double xx = dx * dx, yy = dy * dy;
  return sqrt(xx + yy);
}

size_t vsm_oracle_match_landmarks(const uint64_t *kp_desc, const double *kp_xy, size_t nk,
                                  const double *proj_xy, size_t np, const uint32_t *obs_off,
                                  const uint64_t *obs_desc, double match_max_dist_2d,
                                  int threshold, double dist_2_best, int32_t *out_pairs,
                                  size_t cap) {
  size_t count = 0;
  for (size_t k = 0; k < nk; k++) { /* vo_utils.h:116 */
    size_t min1 = 256, min2 = 256;  /* vo_utils.h:124-126 */
    int best = -1;
    for (size_t p = 0; p < np; p++) { /* vo_utils.h:128 */
      if (norm2d(kp_xy + 2 * k, proj_xy + 2 * p) > match_max_dist_2d) continue; /* 136-142 */
      size_t dmin = 256; /* vo_utils.h:149-161: min over the landmark's observations */
      for (uint32_t o = obs_off[p]; o < obs_off[p + 1]; o++) {
        const size_t d = (size_t)ham256(obs_desc + 4 * (size_t)o, kp_desc + 4 * k);
        if (d < dmin) dmin = d;
      }
      if (dmin < min1) { /* vo_utils.h:165-168 */
        min2 = min1;
        min1 = dmin;
        best = (int)p;
      } else if (dmin < min2) { /* vo_utils.h:172-173 */
        min2 = dmin;
      }
    }
    /* vo_utils.h:178-184.  best == -1 (possible when threshold > 256 and
     * dist_2_best <= 1) makes the reference read projected_track_ids[-1]:
     * undefined behaviour, restated as "no match". */
    if ((int)min1 < threshold && dist_2_best * (int)min1 <= (int)min2 && best >= 0) {
      if (count < cap) {
        out_pairs[2 * count] = (int32_t)k;
        out_pairs[2 * count + 1] = best;
      }
      count++;
    }
  }
  return count;
}
