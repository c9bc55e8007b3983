// landmark_kernels.h -- gfx950 kernels of the keypoint-to-landmark matcher.
//
// Path: visnav::find_matches_landmarks, /root/reference/include/visnav/vo_utils.h:95-187
// (called on every odometry frame, src/odometry.cpp:832, 887).  Same XOR+popcount
// best/second scan as matchDescriptors, but
//   * a candidate landmark counts only if its projection lies within
//     match_max_dist_2d of the keypoint (vo_utils.h:136-142, double precision),
//   * a landmark's distance is the minimum over the descriptors of all its
//     observations (vo_utils.h:149-161),
//   * there is no backward pass / cross-check.
#pragma once
#include "vsmatch_kernels.h"

namespace vsm {

typedef const __attribute__((address_space(4))) double cdouble_t;
typedef const __attribute__((address_space(4))) u32 cu32_t;

// Grid: x = tile of 64 keypoints, y = group of 4 splits of the projected points; wave w of
// workgroup (x, y) scans split y*4 + w for the 64 keypoints of tile x (lane = keypoint) and
// writes its partial (best key, second key) to scratch[split * split_stride + keypoint];
// merge_splits_kernel folds the splits exactly (vsmatch_kernels.h, merge2).
//
// Projected points, observation offsets and observation descriptors are wave-uniform and
// come in through scalar loads.  The radius gate is evaluated per lane; the Hamming
// distances of a landmark's observations are evaluated by the whole wave when ANY lane
// passed the gate (control flow stays uniform) and masked per lane afterwards.
//
// gate_t: host-computed threshold on the SQUARED distance, the largest double x with
// sqrt(x) <= match_max_dist_2d (host sqrt is correctly rounded), so that
// `(a - b).norm() > match_max_dist_2d` == `squared > gate_t` bit for bit with no device
// sqrt.  The squared distance itself is formed as Eigen forms it for a fixed 2-vector:
// two products, one sum, nothing fused (contraction is switched off below).
__global__ __launch_bounds__(256) void landmark_scan_kernel(
    const u32x4* __restrict__ kp_desc, const double2* __restrict__ kp_xy, u32 nk,
    const double* __restrict__ proj_xy, u32 np, const u32* __restrict__ obs_off,
    const u32x4* __restrict__ obs_desc, double gate_t, u32 splits, u32 split_stride,
    uint2* __restrict__ scratch) {
#pragma clang fp contract(off)
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 split = blockIdx.y * 4 + wave;
  if (split >= splits) return;
  const u32 k = blockIdx.x * 64 + lane;
  const u32 kc = min(k, nk - 1);
  u32 q[8];
  {
    const u32x4 a = kp_desc[2 * (size_t)kc], b = kp_desc[2 * (size_t)kc + 1];
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w;
    q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
  }
  const double2 kxy = kp_xy[kc];
  const u32 per = (np + splits - 1) / splits;
  const u32 p0 = min(split * per, np), p1 = min(p0 + per, np);
  cdouble_t* pxy = (cdouble_t*)(uintptr_t)proj_xy;
  cu32_t* off = (cu32_t*)(uintptr_t)obs_off;
  cdesc_t* od = (cdesc_t*)(uintptr_t)obs_desc;

  u32 best = KEY_INIT, second = KEY_INIT;
  for (u32 p = p0; p < p1; p++) {
    const double px = pxy[2 * (size_t)p], py = pxy[2 * (size_t)p + 1];
    const double dx = kxy.x - px, dy = kxy.y - py;
    const double xx = dx * dx, yy = dy * dy;
    const double sq = xx + yy;
    const bool pass = !(sq > gate_t);  // vo_utils.h:136-142: `norm > r` -> continue
    if (__ballot(pass) == 0) continue;
    const u32 o0 = off[p], o1 = off[p + 1];
    u32 dmin = 256;  // vo_utils.h:149
    for (u32 o = o0; o < o1; o++) {
      const u32x8 t = od[o];
      u32 d = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) d += __popc(q[w] ^ t[w]);
      dmin = min(dmin, d);  // vo_utils.h:158-160
    }
    // vo_utils.h:165-175 on packed keys: strict '<', lowest projected index wins ties,
    // 256 is never recorded; a lane outside the radius contributes nothing.
    const u32 key = pass ? ((dmin << IDX_BITS) | p) : 0xffffffffu;
    second = med3_u32(best, key, second);
    best = min(best, key);
  }
  if (k < nk) scratch[(size_t)split * split_stride + k] = make_uint2(best, second);
}

// Accept test (vo_utils.h:178-179) on the merged result of every keypoint and ordered
// emission of (key_point_index, projected_point_index), ascending in the first, as the
// reference's push_back order (vo_utils.h:116, 183-184).  Single 256-thread workgroup.
// out[0..1] (as u64) = number of matches; pairs follow at out_pairs.  A keypoint with no
// candidate (index -1 in the reference, where reading projected_track_ids[-1] is undefined
// behaviour) is never emitted.
// Any workgroup size up to 1024.  `flag` (may be null): raised with system scope after everything
// else has been written (the outputs may live in mapped host memory), for a host that polls.
__global__ __launch_bounds__(1024) void landmark_emit_kernel(const uint2* __restrict__ scratch, u32 nk,
                                                             AcceptParams prm, u64* __restrict__ out_count,
                                                             int2* __restrict__ out_pairs, u64 cap_pairs,
                                                             u64* __restrict__ flag, u64 seq) {
  __shared__ uint16_t lut[260];
  __shared__ u32 wave_cnt[16];
  load_lut(lut, prm);
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  u32 done = 0;
  for (u32 base = 0; base < nk; base += blockDim.x) {
    const u32 i = base + threadIdx.x;
    u32 j = 0;
    const bool hit = i < nk && accepted(scratch[i], lut, prm.threshold, j);
    const u64 m = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = (u32)__popcll(m);
    __syncthreads();
    u32 before = done, all = 0;
    for (u32 w = 0; w < n_waves; w++) {
      if (w < wave) before += wave_cnt[w];
      all += wave_cnt[w];
    }
    if (hit) {
      const u64 pos = before + (u32)__popcll(m & ((1ull << lane) - 1ull));
      if (pos < cap_pairs) out_pairs[pos] = make_int2((int)i, (int)j);
    }
    done += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_count = done;
  if (flag) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace vsm
