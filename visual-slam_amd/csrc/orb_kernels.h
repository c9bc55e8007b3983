// orb_kernels.h -- gfx950 kernels of the ORB orientation + rotated-BRIEF descriptor step.
//
// Path: visnav::computeAngles and visnav::computeDescriptors,
// /root/reference/include/visnav/keypoints.h:151-201 and 203-237 (called through
// detectKeypointsAndDescriptors, keypoints.h:239-245, at src/sfm.cpp:1199,
// src/odometry.cpp:804-807).  They produce the descriptors the matcher consumes.
// Corner detection (OpenCV goodFeaturesToTrack, keypoints.h:133-149) stays on the host.
//
// Exactness: the intensity-centroid moments are integer sums (order free).  atan2, cos and
// sin are evaluated on the HOST with the same libm the reference links (between the two
// kernels), so the device does only what is exactly reproducible: integer sums, and
// double products/sums with contraction off plus round-half-away for the rotated offsets.
#pragma once
#include "orb_pattern.h"
#include "vsmatch_kernels.h"

namespace vsm {

// m10 = sum dx * I(cx + dx, cy + dy), m01 = sum dy * I(.) over the disc dx^2 + dy^2 <= 225
// clipped to the image (keypoints.h:166-193).  One wave per keypoint, 4 keypoints per
// workgroup; lanes walk the 31 x 31 window row by row (x fastest: coalesced reads).
// Batches: kp_img[k] (may be null: one image) names the image of keypoint k, at img + kp_img[k] * img_stride.
__global__ __launch_bounds__(256) void orb_moments_kernel(const uint8_t* __restrict__ img, int w, int h,
                                                          u32 pitch, const double2* __restrict__ corners,
                                                          u32 n, int2* __restrict__ moments,
                                                          const u32* __restrict__ kp_img, size_t img_stride) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 k = blockIdx.x * 4 + wave;
  if (k >= n) return;
  if (kp_img) img += kp_img[k] * img_stride;
  const double2 p = corners[k];
  const int cx = (int)p.x, cy = (int)p.y;  // keypoints.h:158-159 (truncation)
  int m10 = 0, m01 = 0;
  for (int idx = (int)lane; idx < 31 * 31; idx += 64) {
    const int dy = idx / 31 - 15, dx = idx - (dy + 15) * 31 - 15;
    const int x = cx + dx, y = cy + dy;
    if (dx * dx + dy * dy <= 225 && x >= 0 && x < w && y >= 0 && y < h) {  // keypoints.h:171, 175
      const int v = img[(size_t)y * pitch + (size_t)x];
      m10 += dx * v;
      m01 += dy * v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    m10 += __shfl_xor(m10, o);
    m01 += __shfl_xor(m01, o);
  }
  if (lane == 0) moments[k] = make_int2(m10, m01);
}

__device__ __forceinline__ double round_half_away(double x) {  // C round(), keypoints.h:227-228
  const double t = trunc(x);
  return fabs(x - t) >= 0.5 ? t + copysign(1.0, x) : t;
}

// One workgroup per keypoint, thread di = descriptor bit di (keypoints.h:221-233):
// bit = I(c + round(R a)) < I(c + round(R b)) with R = [cos -sin; sin cos]; each wave's
// ballot is one 64-bit word of the packed descriptor.  cs[k] = (cos, sin) of the angle.
// The host has checked that every corner is >= EDGE_THRESHOLD (19) from the border, so
// all reads are inside the image (|R a| <= 18.4); indices are clamped anyway.
__global__ __launch_bounds__(256) void orb_descriptor_kernel(const uint8_t* __restrict__ img, int w, int h,
                                                             u32 pitch, const double2* __restrict__ corners,
                                                             const double2* __restrict__ cs, u32 n,
                                                             u64* __restrict__ desc,
                                                             const u32* __restrict__ kp_img, size_t img_stride) {
#pragma clang fp contract(off)
  const u32 k = blockIdx.x;
  const u32 di = threadIdx.x;
  if (k >= n) return;
  if (kp_img) img += kp_img[k] * img_stride;
  const double2 p = corners[k];
  const int cx = (int)p.x, cy = (int)p.y;  // keypoints.h:213-214
  const double c = cs[k].x, s = cs[k].y;
  const double r00 = c, r01 = -s, r10 = s, r11 = c;  // keypoints.h:219
  const double xa = VSM_ORB_PATTERN[di][0], ya = VSM_ORB_PATTERN[di][1];
  const double xb = VSM_ORB_PATTERN[di][2], yb = VSM_ORB_PATTERN[di][3];
  const double t0 = r00 * xa, t1 = r01 * ya, t2 = r10 * xa, t3 = r11 * ya;
  const double u0 = r00 * xb, u1 = r01 * yb, u2 = r10 * xb, u3 = r11 * yb;
  const double ax = t0 + t1, ay = t2 + t3, bx = u0 + u1, by = u2 + u3;  // keypoints.h:224-225
  const int pax = min(max((int)((double)cx + round_half_away(ax)), 0), w - 1);
  const int pay = min(max((int)((double)cy + round_half_away(ay)), 0), h - 1);
  const int pbx = min(max((int)((double)cx + round_half_away(bx)), 0), w - 1);
  const int pby = min(max((int)((double)cy + round_half_away(by)), 0), h - 1);
  const bool bit = img[(size_t)pay * pitch + (size_t)pax] < img[(size_t)pby * pitch + (size_t)pbx];
  const u64 word = __ballot(bit);
  if ((di & 63) == 0) desc[4 * (size_t)k + (di >> 6)] = word;
}

}  // namespace vsm
