// mfma_scan.h -- the brute-force scan of visnav::matchDescriptors on the gfx950 matrix cores.
//
// Path: /root/reference/include/visnav/keypoints.h:256-290 (and 305-339): per query the
// nearest and second nearest train descriptor by Hamming distance over 256 bits.
//
// The Hamming distance of two bit strings is a dot product: with a -> +1/-1 per bit,
//     sum_k a_k * b_k = (#equal bits) - (#different bits) = 256 - 2 d.
// v_mfma_f32_32x32x64_f8f6f4 on FP4 (e2m1) operands multiplies a 32 x 64 tile by a 64 x 32 tile
// in 32 cycles per SIMD, so four of them give the 1024 distances of 32 train x 32 query
// descriptors in 128 cycles; the popcount path (v_xor + v_bcnt, vsmatch_kernels.h) needs
// 16 x 55.4 = 887 cycles for the same 1024 comparisons.  Every product is +1 or -1 and every sum
// an integer of magnitude <= 256, all exact in the f32 accumulator: the distances are the
// reference's, bit for bit.
//
// Operands.  Bits become FP4 codes with two VALU instructions per register:
//     code = 0x2 | (bit << 3)        (0x2 = +1.0, 0xA = -1.0 in e2m1)
//     reg s = ((w << s) & 0x88888888) | 0x22222222,  s = 0..3
// i.e. register s takes bits 3-s, 7-s, ... 31-s of the 32-bit word w.  That is a permutation of
// the 32 k-positions, and a dot product does not care as long as both operands use the same
// one.  Queries are expanded once per tile (from ~w, so that a product is +1 where the bits
// DIFFER), train descriptors once per 32 x (32 NB) block of comparisons.  The block-scaled form
// of the instruction multiplies every product by 2^-7 and the accumulator starts at the inline
// constant 4.0, so a result is D' = 4 + (2 d - 256) / 128 = 2 + d / 64: exact, and positive, so
// that its bit pattern orders like its value.
//
// Tracking.  A lane of the 32x32 result holds 16 train rows of ONE query column, so it keeps
// its own (best, second) per query block and the two lanes of a column are merged at the end.
// Per block of 1024 distances the common case is eight v_min3_i32 and a compare: only if some
// lane sees a distance at or below its current second best does the wave take the slow path
// (float keys d * 32768 + j, v_med3_f32 / v_min_f32, as vsmatch_kernels.h FILL 5).  The start
// value of (best, second) is a key of distance `d_lim`: the matcher passes the smallest
// distance that can no longer change the accept test (keypoints.h:295-296; accept_limit in
// vsmatch.hip), the raw 2-NN entry point passes 256 (everything is tracked, exactly).
// (Tried, no gain: tracking only the groups of three elements whose minimum passes the test
// (17 more VGPRs); a branch per element in the slow path (exact mode 4.9 -> 3.9 T at NB = 8); train codes 0.0 / 1.0 instead of +-1.0 with pop(q) folded into the lane's
// limit -- the bare instruction runs 10 % faster on half-zero operands (tools/mfma_rate.hip,
// power), this kernel does not: 14.0 T comparisons/s either way on 32768-pair launches.)
//
// Measured (tools/mfma_scan.hip, tools/mfma_rate.hip, profiles/r01): 13.6 T comparisons/s on the
// 5000 x 2000 batch against 2.6 T for the popcount scan; the instruction alone, back to back on
// random +-1 codes at the same two waves per SIMD, sustains 14.5 T (the chip clocks down to
// ~1.9 GHz under it), at 2.4 GHz it would be 19.66 T.
// Build with -mllvm -amdgpu-mfma-vgpr-form: the minimum tree reads the results with VALU
// instructions, from AGPRs every element would first need a v_accvgpr_read.
#pragma once
#include "vsmatch_kernels.h"

namespace vsm {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FMT = cbsz / blgp code of the operand format: 0 = FP8 e4m3 (8 registers per k-step),
// 4 = FP4 e2m1 (4 registers per k-step).
template <int FMT>
__device__ __forceinline__ i32x8 expand_bits(u32 w) {
  i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (FMT == 0) {
#pragma unroll
    for (int s = 0; s < 8; s++) v[s] = (int)(((w << s) & 0x80808080u) | 0x38383838u);
  } else {
#pragma unroll
    for (int s = 0; s < 4; s++) v[s] = (int)(((w << s) & 0x88888888u) | 0x22222222u);
  }
  return v;
}

template <int FMT>
__device__ __forceinline__ f32x16 mfma_step(const i32x8 a, const i32x8 b, const f32x16 c) {
  // block scales (e8m0): A x 2^0, B x 2^-7, so that a product is +-1/128 and the accumulator,
  // started at the inline constant 4.0, stays in [2, 6]
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT, FMT, 0, 0x7f7f7f7f, 0, 0x78787878);
}

// The accumulators start at 4.0 (an inline constant: any other start value costs 16 register
// copies per MFMA chain) and the products are scaled by 1/128, so every D' = 4 + (2 d - 256) / 128
// = 2 + d / 64 is a positive float and its bit pattern orders like the value: the minimum tree
// runs on v_min3_i32 (the compiler's own instruction, so it also places the wait states an MFMA
// result needs before a VALU read; an f32 minimum would add a canonicalising v_max_f32 per input).
constexpr float MFMA_BIAS = 4.0f;
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }

// float key (vsmatch_kernels.h FILL 5) = 2^23 + d * 32768 + j = D' * MFMA_KEY_SCALE + MFMA_KEY_BASE + j
constexpr float MFMA_KEY_SCALE = 64.0f * 32768.0f;
constexpr float MFMA_KEY_BASE = 8388608.0f - 2.0f * MFMA_KEY_SCALE;

// the largest D' that can still change a lane whose second best key is `second`, as its bit pattern
__device__ __forceinline__ int limit_of(float second) {
  const float x = (second - 8388608.0f) * (1.0f / 32768.0f);  // d + j / 32768, exact
  return __float_as_int(2.0f + (ceilf(x) - 1.0f) * (1.0f / 64.0f));
}

// One tile = 32 * NB queries of a job (lane l: column l & 31 of each block of 32, half l >> 5 of
// the k range) against the split's share of the train range, 32 rows at a time.
template <int NB, int FMT>
__device__ __forceinline__ void mfma_scan_tile(const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase,
                                               const Job& job, u32 q0, u32 lane, u32 split, u32 splits,
                                               u32 split_stride, uint2* __restrict__ scratch,
                                               const u32* __restrict__ qlist, u32 d_lim) {
  const u32 per = (job.nt + splits - 1) / splits;
  const u32 t0 = min(split * per, job.nt), t1 = min(t0 + per, job.nt);
  const u32 c = lane & 31, h = lane >> 5;

  i32x8 B[NB][4];
  u32 col[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const u32 qi = min(q0 + b * 32 + c, job.nq - 1);
    col[b] = job.qlist_off != NO_QLIST ? qlist[job.qlist_off + qi] : qi;
    const u32x4 w = qbase[2 * (size_t)(job.q_off + col[b]) + h];
    B[b][0] = expand_bits<FMT>(~w.x);
    B[b][1] = expand_bits<FMT>(~w.y);
    B[b][2] = expand_bits<FMT>(~w.z);
    B[b][3] = expand_bits<FMT>(~w.w);
  }
  const float init = 8388608.0f + 32768.0f * (float)d_lim;
  float best[NB], second[NB];
  int lim[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    best[b] = second[b] = init;
    lim[b] = limit_of(init);
  }

  if (t1 > t0) {
    const u32x4* tr = tbase + 2 * (size_t)job.t_off + h;
    const u32 last = t1 - 1;
    const float z = MFMA_BIAS;
    const f32x16 zero = {z, z, z, z, z, z, z, z, z, z, z, z, z, z, z, z};
    // Software pipeline over (32 train rows, query block) steps: the four MFMAs of the next step
    // are issued before the minimum tree of the current one, so the tree runs in their shadow.
    u32x4 w = tr[2 * (size_t)min(t0 + c, last)];
    u32x4 wn = tr[2 * (size_t)min(t0 + 32 + c, last)];
    i32x8 A[4];
    A[0] = expand_bits<FMT>(w.x);
    A[1] = expand_bits<FMT>(w.y);
    A[2] = expand_bits<FMT>(w.z);
    A[3] = expand_bits<FMT>(w.w);
    f32x16 D = zero;
#pragma unroll
    for (int m = 0; m < 4; m++) D = mfma_step<FMT>(A[m], B[0][m], D);
    i32x8 An[4] = {A[0], A[1], A[2], A[3]};
    for (u32 j = t0; j < t1; j += 32) {
#pragma unroll
      for (int b = 0; b < NB; b++) {
        f32x16 Dn = zero;
        if (b + 1 < NB) {
#pragma unroll
          for (int m = 0; m < 4; m++) Dn = mfma_step<FMT>(A[m], B[b + 1][m], Dn);
        } else {
#pragma unroll
          for (int m = 0; m < 4; m++) Dn = mfma_step<FMT>(An[m], B[0][m], Dn);
        }
        if (b == (NB >= 2 ? NB - 2 : 0)) {
          // operands of the next 32 train rows (past the end: the clamped last row again, result
          // unused), expanded one step early so that the VALU work sits behind this step's MFMAs
          w = wn;
          wn = tr[2 * (size_t)min(j + 64 + c, last)];
          An[0] = expand_bits<FMT>(w.x);
          An[1] = expand_bits<FMT>(w.y);
          An[2] = expand_bits<FMT>(w.z);
          An[3] = expand_bits<FMT>(w.w);
        }
        asm volatile("" : "+v"(Dn));  // keeps the MFMAs above the test below (and their result in VGPRs)
#define DI(k) __float_as_int(D[k])
        const int m0 = min3i(DI(0), DI(1), DI(2)), m1 = min3i(DI(3), DI(4), DI(5)), m2 = min3i(DI(6), DI(7), DI(8));
        const int m3 = min3i(DI(9), DI(10), DI(11)), m4 = min3i(DI(12), DI(13), DI(14));
        const int lo = min(min3i(m0, m1, m2), min3i(m3, m4, DI(15)));
#undef DI
        // issue order of a step: two MFMAs, then the minimum tree spread behind the other two
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(lo <= lim[b]) != 0, 0)) {
          // slow path: some lane has a candidate.  Row of element i: (i & 3) + 8 (i >> 2) + 4 h.
          const float cj = MFMA_KEY_BASE + (float)(j - t0 + 4 * h);
          float bb = best[b], ss = second[b];
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const u32 off = (u32)((i & 3) + 8 * (i >> 2));
            float key = fmaf(D[i], MFMA_KEY_SCALE, cj + (float)off);
            key = (j + 4 * h + off < t1) ? key : 16777216.0f;
            ss = __builtin_amdgcn_fmed3f(bb, key, ss);
            bb = fminf(bb, key);
          }
          best[b] = bb;
          second[b] = ss;
          lim[b] = limit_of(ss);
        }
        D = Dn;
      }
#pragma unroll
      for (int m = 0; m < 4; m++) A[m] = An[m];
    }
  }

  uint2* out = scratch + (size_t)split * split_stride + job.out_off;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    // the two halves of the wave hold the same query columns over disjoint train rows
    const float ob = __shfl_xor(best[b], 32), os = __shfl_xor(second[b], 32);
    const float mb = fminf(best[b], ob);
    const float ms = fminf(fminf(second[b], os), fmaxf(best[b], ob));
    const u32 qi = q0 + b * 32 + c;
    if (h == 0 && qi < job.nq)
      out[col[b]] = make_uint2(fkey_to_key(__float_as_uint(mb), t0), fkey_to_key(__float_as_uint(ms), t0));
  }
}

// WAVES == 1: same grid as scan_kernel_adaptive<*, 1, *> (one-wave workgroups): tiles of
// 32 * NBMAX queries and remainder tiles of half and a quarter of that, grouped or job-minor
// block order.  WAVES == 4: same grid as scan_kernel (4-wave workgroups, job-minor), used with
// NBMAX == 2 for the pruned backward pass, most of whose workgroups find nothing to do.
template <int FMT, int NBMAX, int WAVES = 1>
__global__ __launch_bounds__(64 * WAVES) void mfma_scan_kernel(const u32x4* __restrict__ qbase,
                                                               const u32x4* __restrict__ tbase,
                                                               const Job* __restrict__ jobs, u32 n_jobs, u32 splits,
                                                               u32 split_stride, uint2* __restrict__ scratch,
                                                               const u32* __restrict__ qlist, u32 tiles_per_job,
                                                               u32 d_lim) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  u32 job_id, t;
  if (WAVES == 1 && tiles_per_job) {
    const u32 per_group = 8 * tiles_per_job;
    const u32 group = blockIdx.x / per_group, r = blockIdx.x - group * per_group;
    t = r >> 3;
    job_id = group * 8 + (r & 7);
    if (job_id >= n_jobs) return;
  } else {
    const u32 wg_of_job = blockIdx.x / n_jobs;
    job_id = blockIdx.x - wg_of_job * n_jobs;
    t = wg_of_job * WAVES + wave;
  }
  const Job job = jobs[job_id];
  constexpr u32 RMAX = NBMAX / 2;  // tiles are counted in units of 64 queries, as adaptive_tiles does
  const u32 n64 = (job.nq + 63) >> 6, full = n64 / RMAX, rem = n64 & (RMAX - 1);
  if (t < full) {
    mfma_scan_tile<NBMAX, FMT>(qbase, tbase, job, t * 64 * RMAX, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim);
    return;
  }
  u32 k = t - full, q0 = full * 64 * RMAX;
  if (RMAX > 2 && (rem & 2)) {
    if (k == 0) {
      mfma_scan_tile<4, FMT>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim);
      return;
    }
    k--;
    q0 += 128;
  }
  if ((rem & 1) && k == 0)
    mfma_scan_tile<2, FMT>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim);
}

}  // namespace vsm
