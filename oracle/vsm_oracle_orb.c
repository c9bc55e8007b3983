/*
 * vsm_oracle_orb.c -- CPU restatement of visnav::computeAngles and
 * visnav::computeDescriptors (/root/reference/include/visnav/keypoints.h:151-201,
 * 203-237), the step that produces the matcher's input.  SURVEY.md section 8f row N2.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as vsm_oracle.c).
 *
 * Parity status: PINNED by the reference's own golden vectors for these two
 * functions (test/src/test_ex3.cpp:76-146): images test/ex3_test_data/0_{0,1}.jpg
 * (committed as tests/golden/ex3_0_{0,1}.jpg, decoded with PIL -- the decode was
 * checked to reproduce the goldens) with the corners, angles (tolerance 1e-8, the
 * reference test's own) and descriptors (exact) of kd{0,1}.json, committed as
 * tests/golden/ex3_kd{0,1}_corners.npy / _angles.npy and ex3_kd{0,1}.npy.
 * The real functions cannot be compiled here (they take pangolin::ManagedImage and
 * Eigen types; Pangolin and Eigen are absent).  Detection (keypoints.h:133-149,
 * OpenCV goodFeaturesToTrack) is not restated: corners are an input.
 *
 * Image: 8-bit, row-major, `pitch` bytes per row; img(x, y) = img[y * pitch + x]
 * (pangolin::Image::operator()(x, y), as used at keypoints.h:179-180, 227-228).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "orb_pattern.h"

/* keypoints.h:151-201.  angles[i] = atan2(m01, m10) over the disc dx^2 + dy^2 <= 225
 * clipped to the image; 0 when rotate_features is false.  The moments are sums of
 * small integers, exact in double whatever the order. */
void vsm_oracle_orb_angles(const uint8_t *img, int w, int h, size_t pitch, const double *corners_xy,
                           size_t n, int rotate_features, double *angles) {
  for (size_t i = 0; i < n; i++) {
    const int cx = (int)corners_xy[2 * i], cy = (int)corners_xy[2 * i + 1]; /* keypoints.h:158-159 */
    double angle = 0;
    if (rotate_features) {
      double m01 = 0, m10 = 0;
      for (int dx = -15; dx < 16; dx++)
        for (int dy = -15; dy < 16; dy++)
          if (dx * dx + dy * dy <= 225) { /* keypoints.h:171 */
            const int x = cx + dx, y = cy + dy;
            if (0 <= x && x < w && 0 <= y && y < h) { /* InBounds, keypoints.h:175 */
              const int v = img[(size_t)y * pitch + (size_t)x];
              m10 = m10 + dx * v; /* keypoints.h:179-180 */
              m01 = m01 + dy * v;
            }
          }
      angle = atan2(m01, m10); /* keypoints.h:195 */
    }
    angles[i] = angle;
  }
}

/* keypoints.h:203-237.  Returns 0, or -1 if a corner lies closer than EDGE_THRESHOLD (19)
 * to the border: the reference reads outside the image there (undefined behaviour;
 * detectKeypoints never produces such corners, keypoints.h:145). */
__attribute__((optimize("fp-contract=off"))) int vsm_oracle_orb_descriptors(
    const uint8_t *img, int w, int h, size_t pitch, const double *corners_xy, const double *angles,
    size_t n, uint64_t *desc) {
  for (size_t i = 0; i < n; i++) {
    const int cx = (int)corners_xy[2 * i], cy = (int)corners_xy[2 * i + 1]; /* keypoints.h:213-214 */
    if (cx < VSM_ORB_EDGE_THRESHOLD || cx >= w - VSM_ORB_EDGE_THRESHOLD || cy < VSM_ORB_EDGE_THRESHOLD ||
        cy >= h - VSM_ORB_EDGE_THRESHOLD)
      return -1;
    const double c = cos(angles[i]), s = sin(angles[i]); /* keypoints.h:219 */
    const double r00 = c, r01 = -s, r10 = s, r11 = c;
    uint64_t d[4] = {0, 0, 0, 0};
    for (int di = 0; di < 256; di++) {
      const double xa = VSM_ORB_PATTERN[di][0], ya = VSM_ORB_PATTERN[di][1];
      const double xb = VSM_ORB_PATTERN[di][2], yb = VSM_ORB_PATTERN[di][3];
      /* Rot * pa, Rot * pb (keypoints.h:224-225): row times column, products then sum */
      const double ax = r00 * xa + r01 * ya, ay = r10 * xa + r11 * ya;
      const double bx = r00 * xb + r01 * yb, by = r10 * xb + r11 * yb;
      /* img_raw(cx + round(.), cy + round(.)): double sums converted to size_t (227-228) */
      const size_t pax = (size_t)(cx + round(ax)), pay = (size_t)(cy + round(ay));
      const size_t pbx = (size_t)(cx + round(bx)), pby = (size_t)(cy + round(by));
      if (img[pay * pitch + pax] < img[pby * pitch + pbx]) d[di >> 6] |= 1ull << (di & 63);
    }
    for (int k = 0; k < 4; k++) desc[4 * i + k] = d[k];
  }
  return 0;
}
