// gftt_kernels.h -- gfx950 kernels of the corner detector.
//
// Path: visnav::detectKeypoints, /root/reference/include/visnav/keypoints.h:133-149 =
// cv::goodFeaturesToTrack(image, points, num_features, 0.01, 8) + the EDGE_THRESHOLD filter.
// OpenCV is a system dependency of the reference (absent here, no version pinned); the
// per-pixel part of its published algorithm (3x3 Sobel -> covariance -> 3x3 box sums ->
// minimum eigenvalue -> threshold at quality * max -> 3x3 local maxima) runs in these
// kernels; the ordering of the candidates and the sequential minimum-distance pass stay on
// the host (vsmatch.hip), as does the EDGE_THRESHOLD filter.
//
// Batches (vsm_detect_and_describe_batch): blockIdx.z is the image of a batch of equally sized
// images; image z lives at img + z * img_stride and its work areas at z * (w * h) elements (the
// candidate capacity `cap` is w * h), its (max key, count) pair at misc + 2 z.  One image: z = 0.
//
// Float arithmetic is kept exactly as the restatement in the checker does it: products and
// sums in float with contraction off, the 3x3 box sums in double, correctly rounded sqrt.
#pragma once
#include "vsmatch_kernels.h"

namespace vsm {

__device__ __forceinline__ int reflect101(int p, int n) {  // BORDER_REFLECT_101, |offset| <= 1
  if (n == 1) return 0;
  return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p);
}

// Order-preserving map float -> u32 (for atomicMax over floats of either sign).
__device__ __forceinline__ u32 float_key(float f) {
  const u32 b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(u32 k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// eig = minimum eigenvalue of the 3x3-summed covariance matrix of the image gradient
// (cv::cornerMinEigenVal for 8-bit input, block size 3, aperture 3); *max_key = float_key(max eig)
// over the image (zero-initialised by the host).  A workgroup produces a 64 x 16 tile of `eig`:
//  1. the gradient (Dx, Dy) of every pixel of the tile and its one-pixel halo, once: Sobel 3x3 scaled by
//     1 / (4 * 3 * 255) -- Dx: row filter [-1 0 1] (exact), column filter [f1 f0 f1]; Dy the
//     transpose -- with BORDER_REFLECT_101, evaluated at the REFLECTED coordinate where the halo leaves
//     the image (that is the gradient the box sum reads there); its three products Dx Dx, Dx Dy, Dy Dy
//     (float) go to LDS as doubles;
//  2. the horizontal sums of three, left to right from 0.0 (double);
//  3. the vertical sums (top + middle) + bottom (double), then a = (float)sxx / 2, b = (float)sxy,
//     c = (float)syy / 2, eig = (a + c) - sqrt((a - c)^2 + b^2) in float, correctly rounded sqrt.
// Operation for operation what the checker does per pixel (products and sums in float with contraction
// off, box sums in double in this order), so the same bits.  (Round 1's form -- a Sobel kernel writing
// the gradient image and a thread-per-pixel kernel forming all nine products of its window again, 27
// float -> double conversions and 27 double additions per pixel, 386 instructions per wave -- took
// 3.9 + 19 us per 752 x 480 image; this one kernel replaces both.)
constexpr int GFTT_TILE_W = 64, GFTT_TILE_H = 16;
__global__ __launch_bounds__(256) void gftt_sobel_mineig_kernel(const uint8_t* __restrict__ img, int w, int h, u32 pitch,
                                                                float* __restrict__ eig, u32* __restrict__ max_key,
                                                                size_t img_stride) {
#pragma clang fp contract(off)
  constexpr int PW = GFTT_TILE_W + 2, PH = GFTT_TILE_H + 2;
  __shared__ double prod[3][PH][PW];           // pxx, pxy, pyy of the tile and its halo
  __shared__ double rows[3][PH][GFTT_TILE_W];  // their horizontal sums of three
  __shared__ u32 wave_max[4];
  img += blockIdx.z * img_stride;
  eig += blockIdx.z * ((size_t)w * h);
  max_key += 2 * blockIdx.z;
  const int X0 = blockIdx.x * GFTT_TILE_W, Y0 = blockIdx.y * GFTT_TILE_H;
  const float f1 = (float)(1.0 / (4.0 * 3.0 * 255.0)), f0 = 2.0f * f1;
  for (int cell = threadIdx.x; cell < PH * PW; cell += 256) {
    const int ty = cell / PW, tx = cell - ty * PW;
    const int cx = X0 - 1 + tx, cy = Y0 - 1 + ty;
    float pxx = 0.0f, pxy = 0.0f, pyy = 0.0f;
    if (cx <= w && cy <= h) {  // cells further out serve no pixel of the image
      const int x = reflect101(cx, w), y = reflect101(cy, h);
      const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * pitch;
      const uint8_t* r1 = img + (size_t)y * pitch;
      const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * pitch;
      const float a00 = r0[xm], a01 = r0[x], a02 = r0[xp];
      const float a10 = r1[xm], a12 = r1[xp];
      const float a20 = r2[xm], a21 = r2[x], a22 = r2[xp];
      const float d0 = a02 - a00, d1 = a12 - a10, d2 = a22 - a20;
      const float dxa = d1 * f0, dxb = (d0 + d2) * f1;
      const float ta = a01 * f0, tb = (a00 + a02) * f1, ba = a21 * f0, bb = (a20 + a22) * f1;
      const float top = ta + tb, bot = ba + bb;
      const float gx = dxa + dxb, gy = bot - top;
      pxx = gx * gx;
      pxy = gx * gy;
      pyy = gy * gy;
    }
    prod[0][ty][tx] = (double)pxx;
    prod[1][ty][tx] = (double)pxy;
    prod[2][ty][tx] = (double)pyy;
  }
  __syncthreads();
  for (int cell = threadIdx.x; cell < PH * GFTT_TILE_W; cell += 256) {
    const int r = cell / GFTT_TILE_W, c = cell - r * GFTT_TILE_W;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double s = 0;
      s += prod[k][r][c];
      s += prod[k][r][c + 1];
      s += prod[k][r][c + 2];
      rows[k][r][c] = s;
    }
  }
  __syncthreads();
  u32 key = 0;
  for (int o = threadIdx.x; o < GFTT_TILE_H * GFTT_TILE_W; o += 256) {
    const int r = o / GFTT_TILE_W, c = o - r * GFTT_TILE_W;
    const int x = X0 + c, y = Y0 + r;
    if (x < w && y < h) {
      const double sxx = (rows[0][r][c] + rows[0][r + 1][c]) + rows[0][r + 2][c];
      const double sxy = (rows[1][r][c] + rows[1][r + 1][c]) + rows[1][r + 2][c];
      const double syy = (rows[2][r][c] + rows[2][r + 1][c]) + rows[2][r + 2][c];
      const float a = (float)sxx * 0.5f, b = (float)sxy, cc = (float)syy * 0.5f;
      const float amc = a - cc;
      const float t1 = amc * amc, t2 = b * b;
      const float e = (a + cc) - sqrtf(t1 + t2);
      eig[(size_t)y * w + x] = e;
      key = max(key, float_key(e));
    }
  }
  for (int o = 32; o > 0; o >>= 1) key = max(key, (u32)__shfl_xor((int)key, o));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    if (m) atomicMax(max_key, m);
  }
}

// Candidates: interior pixels whose eigenvalue survives the threshold (float)(max * quality),
// is non-zero and equals the 3x3 maximum of the thresholded map (cv::threshold TOZERO +
// cv::dilate + the `val != 0 && val == tmp` test).  Appended unordered as
// (float_key(value) << 32 | pixel index), a key whose unsigned DEScending order is OpenCV's
// candidate order: value descending, ties by the higher pixel address first (greaterThanPtr).
__global__ __launch_bounds__(256) void gftt_candidates_kernel(const float* __restrict__ eig, int w, int h,
                                                              const u32* __restrict__ max_key, double quality,
                                                              u64* __restrict__ cand, u32* __restrict__ count,
                                                              u32 cap) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  eig += blockIdx.z * ((size_t)w * h);
  max_key += 2 * blockIdx.z;
  count += 2 * blockIdx.z;
  cand += blockIdx.z * (size_t)cap;
  bool is_cand = x >= 1 && y >= 1 && x < w - 1 && y < h - 1;
  float v = 0.0f;
  if (is_cand) {
    const float thr = (float)((double)key_float(*max_key) * quality);
    v = eig[(size_t)y * w + x];
    is_cand = v > thr && v != 0.0f;
    if (is_cand) {
#pragma unroll
      for (int j = -1; j <= 1; j++)
#pragma unroll
        for (int i = -1; i <= 1; i++) {
          float u = eig[(size_t)(y + j) * w + (x + i)];
          if (!(u > thr)) u = 0.0f;
          is_cand = is_cand && !(u > v);
        }
    }
  }
  // one atomic per wave (the order of the list is arbitrary anyway; the sort that follows fixes it):
  // a counter per image takes every candidate of that image, 15 k same-address atomics otherwise
  const u64 m = __ballot(is_cand);
  if (m == 0) return;
  const u32 lane = threadIdx.x & 63;
  u32 base = 0;
  if (lane == (u32)__builtin_ctzll(m)) base = atomicAdd(count, (u32)__builtin_popcountll(m));
  base = (u32)__shfl((int)base, __builtin_ctzll(m));
  if (!is_cand) return;
  const u32 slot = base + (u32)__builtin_popcountll(m & (((u64)1 << lane) - 1));
  if (slot < cap) cand[slot] = ((u64)float_key(v) << 32) | (u32)(y * w + x);
}

// Orders the candidates by brute force, the way the matcher scans descriptors: thread i
// counts the keys greater than its own (tiles of 256 keys through LDS, broadcast reads) and
// that count is its position -- keys are unique (they carry the pixel index), so this is a
// permutation.  n^2 / 2^14 lane-iterations: 15 k candidates take ~25 us, against ~1 ms for a
// host sort; the host falls back to std::sort above GFTT_RANK_SORT_MAX.
constexpr u32 GFTT_RANK_SORT_MAX = 65536;
constexpr u32 GFTT_RANK_BLOCK = 64;  // one wave per workgroup: 15 k candidates still cover ~all 256 CUs
// (the bitonic sort of a batch, below: block size, padded size, and which images it orders)
constexpr u32 GFTT_LDS_SORT_BLOCK = 16384;
__host__ __device__ __forceinline__ u32 gftt_sort_size(u32 n) {
  u32 N = 64;
  while (N < n) N <<= 1;
  return N;
}
__host__ __device__ __forceinline__ bool gftt_lds_sorts(u32 n, u32 cap) {
  return n <= GFTT_RANK_SORT_MAX && gftt_sort_size(n) <= cap;
}
// skip_lds_sorted: images gftt_lds_sort_kernel orders (a batch) are left alone.
// host_count / host_first (single image; may be null): MAPPED host memory that receives the candidate
// count and the first host_first_n ordered candidates straight from this kernel -- what the host's
// sequential pass reads next, without a copy in between.
__global__ __launch_bounds__(GFTT_RANK_BLOCK) void gftt_rank_sort_kernel(const u64* __restrict__ cand,
                                                                         const u32* __restrict__ count, u32 cap,
                                                                         u64* __restrict__ sorted, u32 skip_lds_sorted,
                                                                         u32* __restrict__ host_count,
                                                                         u64* __restrict__ host_first, u32 host_first_n) {
  __shared__ u64 tile[256];
  count += 2 * blockIdx.z;
  cand += blockIdx.z * (size_t)cap;
  sorted += blockIdx.z * (size_t)cap;
  if (host_count && blockIdx.x == 0 && threadIdx.x == 0) *host_count = *count;
  const u32 n = min(*count, cap);
  if (n > GFTT_RANK_SORT_MAX || blockIdx.x * GFTT_RANK_BLOCK >= n) return;  // block-uniform
  if (skip_lds_sorted && gftt_lds_sorts(n, cap)) return;
  const u32 i = blockIdx.x * GFTT_RANK_BLOCK + threadIdx.x;
  const u64 mine = i < n ? cand[i] : 0;
  u32 r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  for (u32 base = 0; base < n; base += 256) {
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 256; k += GFTT_RANK_BLOCK) {
      const u32 j = base + k + threadIdx.x;
      tile[k + threadIdx.x] = j < n ? cand[j] : 0;  // 0 is below every real key
    }
    __syncthreads();
#pragma unroll 4
    for (u32 j = 0; j < 256; j += 4) {  // four independent counters: the compares do not serialise
      r0 += tile[j] > mine ? 1u : 0u;
      r1 += tile[j + 1] > mine ? 1u : 0u;
      r2 += tile[j + 2] > mine ? 1u : 0u;
      r3 += tile[j + 3] > mine ? 1u : 0u;
    }
  }
  if (i < n) {
    const u32 rank = (r0 + r1) + (r2 + r3);
    sorted[rank] = mine;
    if (host_first && rank < host_first_n) host_first[rank] = mine;
  }
}

// A batch of images: the brute-force order above is n^2 work per image whatever the batch (15 k
// candidates: 70 us of the whole chip), so a batch sorts each image's candidates with ONE workgroup
// instead -- a bitonic network over the keys padded with zeros to a power of two N, descending -- and
// the images run side by side on the CUs.  Blocks of GFTT_LDS_SORT_BLOCK keys (128 KB) are sorted in
// LDS; for N beyond one block the merges with a partner distance >= the block run through global
// memory (`sorted`, N <= cap entries), the rest of each merge again block by block in LDS.  Images
// with more than GFTT_RANK_SORT_MAX candidates are left to the host, images whose N exceeds cap to
// gftt_rank_sort_kernel (gftt_lds_sorts() says which).  blockIdx.x = image.
// the merges of partner distance < span over the LDS block whose first key has global index g0
__device__ __forceinline__ void gftt_lds_merge(u64* keys, u32 B, u32 g0, u32 k_first, u32 k_last, u32 j_first) {
  for (u32 k = k_first; k <= k_last; k <<= 1)
    for (u32 j = min(k >> 1, j_first); j > 0; j >>= 1) {
      for (u32 t = threadIdx.x; t < B / 2; t += 1024) {
        const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;  // the pair (i, i + j), bit j of i clear
        const bool descending = ((g0 + i) & k) == 0;                    // the last merge (k = N) is descending throughout
        const u64 a = keys[i], b = keys[l];
        if ((a < b) == descending) {
          keys[i] = b;
          keys[l] = a;
        }
      }
      __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void gftt_lds_sort_kernel(const u64* __restrict__ cand, const u32* __restrict__ count,
                                                             u32 cap, u64* sorted) {
  extern __shared__ u64 gftt_sort_keys[];
  u64* keys = gftt_sort_keys;
  count += 2 * blockIdx.x;
  cand += blockIdx.x * (size_t)cap;
  sorted += blockIdx.x * (size_t)cap;
  const u32 n = min(*count, cap);
  if (n == 0 || !gftt_lds_sorts(n, cap)) return;  // block-uniform
  const u32 N = gftt_sort_size(n), B = min(N, GFTT_LDS_SORT_BLOCK);
  // every block sorted on its own, directions alternating as the network wants them
  for (u32 g0 = 0; g0 < N; g0 += B) {
    for (u32 i = threadIdx.x; i < B; i += 1024) keys[i] = g0 + i < n ? cand[g0 + i] : 0;  // 0 is below every real key
    __syncthreads();
    gftt_lds_merge(keys, B, g0, 2, B, B);
    if (N == B) {
      for (u32 i = threadIdx.x; i < n; i += 1024) sorted[i] = keys[i];
      return;
    }
    for (u32 i = threadIdx.x; i < B; i += 1024) sorted[g0 + i] = keys[i];
    __syncthreads();
  }
  // the merges across blocks
  for (u32 k = 2 * B; k <= N; k <<= 1) {
    for (u32 j = k >> 1; j >= B; j >>= 1) {
      for (u32 t = threadIdx.x; t < N / 2; t += 1024) {
        const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const bool descending = (i & k) == 0;
        const u64 a = sorted[i], b = sorted[l];
        if ((a < b) == descending) {
          sorted[i] = b;
          sorted[l] = a;
        }
      }
      __syncthreads();  // global writes of this workgroup are visible to it after the barrier
    }
    for (u32 g0 = 0; g0 < N; g0 += B) {
      for (u32 i = threadIdx.x; i < B; i += 1024) keys[i] = sorted[g0 + i];
      __syncthreads();
      gftt_lds_merge(keys, B, g0, k, k, B >> 1);
      for (u32 i = threadIdx.x; i < B; i += 1024) sorted[g0 + i] = keys[i];
      __syncthreads();
    }
  }
}

}  // namespace vsm
