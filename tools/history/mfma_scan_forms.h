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
// lane sees a distance below its current second best does the wave leave that path.  The start
// value of (best, second) is a key of distance `d_lim`: the matcher passes the smallest
// distance that can no longer change the accept test (keypoints.h:295-296; accept_limit in
// vsmatch.hip), the raw 2-NN entry point passes 256 (everything is tracked, exactly).
// Two forms of what happens off the common path, chosen by the host from how often it happens
// (count_steps below; vsmatch.hip scan_track):
//   * mfma_scan_tile   -- float keys d * 32768 + j, v_med3_f32 / v_min_f32 per element (as
//     vsmatch_kernels.h FILL 5), ~90 instructions: for data on which it is rare (iid bits);
//   * mfma_scan_tile_d -- the two smallest DISTANCES by a 12-instruction tournament, the train
//     index recovered at the end of the scan: for real ORB descriptors, where nearly every
//     step leaves the common path.  Described at the function.
// (Tried on the key form, no gain: tracking only the groups of three elements whose minimum
// passes the test (17 more VGPRs); a branch per element in the slow path (exact mode 4.9 -> 3.9 T
// at NB = 8); train codes 0.0 / 1.0 instead of +-1.0 with pop(q) folded into the lane's limit --
// the bare instruction runs 10 % faster on half-zero operands (tools/mfma_rate.hip, power),
// this kernel does not: 14.0 T comparisons/s either way on 32768-pair launches.)
//
// Measured (tools/mfma_scan.hip, tools/mfma_rate.hip, tools/valu_beside_mfma.hip, profiles/r01):
// 13.7 T comparisons/s on the 5000 x 2000 batch against 2.6 T for the popcount scan; the
// instruction alone, back to back on random +-1 codes at the same two waves per SIMD, sustains
// 14.5 T (the chip clocks down to ~1.9 GHz under it; at 2.4 GHz it would be 19.66 T), and the
// nine VALU instructions of the common path beside it cost the remaining ~10 %.
// Build with -mllvm -amdgpu-mfma-vgpr-form: the minimum tree reads the results with VALU
// instructions, from AGPRs every element would first need a v_accvgpr_read.
#pragma once
#include "scan_variants_kernels.h"
#include <type_traits>

namespace vsm {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FMT = cbsz / blgp code of the operand format: 4 = FP4 e2m1 (4 registers per k-step), what
// ships; 0 = FP8 e4m3 (8 registers per k-step, half the MFMA rate: 5.9 against 9.6 T with the
// first version of this kernel), not instantiated.
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

// The same FP4 expansion with the two masks in REGISTERS (0x88888888 in an SGPR, 0x22222222 in a
// VGPR): gfx950's three-operand encoding takes no literal, so with literal masks the compiler emits
// v_and + v_or per register (44 instructions per 32 train rows with the shifts); from registers it
// is one v_and_or_b32 (28).
__device__ __forceinline__ i32x8 expand_bits_r(u32 w, u32 m88, u32 c22) {
  i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 4; s++) v[s] = (int)(((w << s) & m88) | c22);
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

// Statistics for the host's choice between the two tracking forms (vsmatch.hip): steps (32 train
// rows x 32 queries) a wave went through and how many of them left the common path.  A sample:
// one workgroup in 64 reports (every wave reporting made two atomics on one address per wave
// and doubled the time of launches of many short waves, 0.88 -> 2.06 ms at euroc_V1's sizes).
__device__ __forceinline__ void count_steps(unsigned long long* stats, u32 lane, u32 steps, u32 slow) {
  if (stats && lane == 0 && (blockIdx.x & 63u) == 0) {
    atomicAdd(stats, (unsigned long long)steps);
    if (slow) atomicAdd(stats + 1, (unsigned long long)slow);
  }
}

// Forward scans of the batched, pruned path (unsplit, no query list): the accept test of
// keypoints.h:295-296 on the finished (best, second) of the 32 queries of a block, one bit per
// query, stored as word (first query of the block) / 32 of the pair's mask.  nominate_kernel then
// reads 4 bytes per 32 queries instead of every 8-byte result (524 MB per 32768 pairs of 2000
// descriptors, 170 us).  `acc.mask` null: not wanted.
struct AcceptMask {
  const uint16_t* lut;  // AcceptParams::min_second on the device
  int threshold;
  u32* mask;
};
__device__ __forceinline__ void store_accept_mask(const AcceptMask& acc, const Job& job, u32 lane, u32 q_first,
                                                  bool valid, u32 kb, u32 ks) {
  if (!acc.mask) return;
  const u32 d = kb >> IDX_BITS, sd = min(ks >> IDX_BITS, 256u);
  const bool hit = valid && d < 256u && (int)d < acc.threshold && sd >= (u32)acc.lut[min(d, 256u)];
  const u64 m = __ballot(hit);
  if (lane == 0) acc.mask[(job.out_off >> 6) + (q_first >> 5)] = (u32)m;  // valid only in lanes 0..31
}

// One tile = 32 * NB queries of a job (lane l: column l & 31 of each block of 32, half l >> 5 of
// the k range) against the split's share of the train range, 32 rows at a time.
template <int NB, int FMT>
__device__ __forceinline__ void mfma_scan_tile(const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase,
                                               const Job& job, u32 q0, u32 lane, u32 split, u32 splits,
                                               u32 split_stride, uint2* __restrict__ scratch,
                                               const u32* __restrict__ qlist, u32 d_lim,
                                               unsigned long long* __restrict__ stats, const AcceptMask& acc) {
  const u32 per = (job.nt + splits - 1) / splits;
  const u32 t0 = min(split * per, job.nt), t1 = min(t0 + per, job.nt);
  const u32 c = lane & 31, h = lane >> 5;
  u32 slow_steps = 0;

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
          slow_steps++;
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

  count_steps(stats, lane, (t1 - t0 + 31) / 32 * NB, slow_steps);
  uint2* out = scratch + (size_t)split * split_stride + job.out_off;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    // the two halves of the wave hold the same query columns over disjoint train rows
    const float ob = __shfl_xor(best[b], 32), os = __shfl_xor(second[b], 32);
    const float mb = fminf(best[b], ob);
    const float ms = fminf(fminf(second[b], os), fmaxf(best[b], ob));
    const u32 qi = q0 + b * 32 + c;
    const u32 kb = fkey_to_key(__float_as_uint(mb), t0), ks = fkey_to_key(__float_as_uint(ms), t0);
    if (h == 0 && qi < job.nq) out[col[b]] = make_uint2(kb, ks);
    store_accept_mask(acc, job, lane, q0 + b * 32, h == 0 && qi < job.nq, kb, ks);
  }
}

// ---- Distance-only tracking (TRACK 1) ------------------------------------------------------
// The tile above tracks float keys d * 32768 + j, ~90 instructions whenever some lane of the wave
// has a distance at or below its second best.  On iid random bits that is rare; on REAL ORB
// descriptors it is not: unrelated descriptors are 126 +- 19 bits apart (tests/golden/ex3_kd*),
// 2 % of all pairs are below the accept limit of 83 and 95 % of the 32 x 32 blocks hold one,
// so the slow path ran in nearly every step: 5.3 T comparisons/s instead of 13.8 T
// (tools/real_like_probe.py).  This tile tracks the two smallest DISTANCES only: the same
// eight-instruction minimum and test, then a 12-instruction tournament for the two smallest of the
// lane's sixteen results and five instructions to merge them into (best, second) and to remember
// the 32-row step in which the lane's best last improved.  The train INDEX of the best is
// recovered at the end of the scan: the lane recomputes the distances of the 16 rows it owns in
// that step with v_xor / v_bcnt and takes the lowest row that has the best distance -- the
// reference's "strict <, ascending scan" (keypoints.h:273) exactly: the remembered step is the
// first one that reached the final best distance, and rows are taken in ascending order.  The
// second best needs no index (keypoints.h:296 reads its distance only).
// (Tried: the tournament in every step, no test and no branch -- the same rate where every step
// is slow anyway (7.9 T in exact mode at 2000 x 2000 either way) and 9.2 T on any data: the slow
// step is bound by its ~25 three-operand VALU instructions, not by the branch.)
template <int NB, int FMT>
__device__ __forceinline__ void mfma_scan_tile_d(const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase,
                                                 const Job& job, u32 q0, u32 lane, u32 split, u32 splits,
                                                 u32 split_stride, uint2* __restrict__ scratch,
                                                 const u32* __restrict__ qlist, u32 d_lim,
                                                 unsigned long long* __restrict__ stats, const AcceptMask& acc) {
  const u32 per = (job.nt + splits - 1) / splits;
  const u32 t0 = min(split * per, job.nt), t1 = min(t0 + per, job.nt);
  const u32 c = lane & 31, h = lane >> 5;
  u32 slow_steps = 0;

  i32x8 B[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const u32 qi = min(q0 + b * 32 + c, job.nq - 1);
    const u32 colb = job.qlist_off != NO_QLIST ? qlist[job.qlist_off + qi] : qi;
    const u32x4 w = qbase[2 * (size_t)(job.q_off + colb) + h];
    B[b][0] = expand_bits<FMT>(~w.x);
    B[b][1] = expand_bits<FMT>(~w.y);
    B[b][2] = expand_bits<FMT>(~w.z);
    B[b][3] = expand_bits<FMT>(~w.w);
  }
  const float dlim = 2.0f + (float)d_lim * (1.0f / 64.0f);  // D' of distance d_lim
  // `best` carries, in the 10 lowest mantissa bits of its D' (one distance unit is 2^15 or 2^16
  // units in the last place, a range has at most 1024 steps of 32 rows), the number of the step
  // in which it was found: v_min then keeps the EARLIEST step of equal distances, which is what
  // the index recovery needs, at the price of one v_or per slow step and no register.
  float best[NB], second[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) best[b] = second[b] = dlim;

  if (t1 > t0) {
    const u32x4* tr = tbase + 2 * (size_t)job.t_off + h;
    const u32 last = t1 - 1;
    const float z = MFMA_BIAS;
    const f32x16 zero = {z, z, z, z, z, z, z, z, z, z, z, z, z, z, z, z};
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
    // the two smallest of the lane's sixteen results merged into (best, second) of block b
    auto med3 = [](int p, int q, int r) {
      return __float_as_int(__builtin_amdgcn_fmed3f(__int_as_float(p), __int_as_float(q), __int_as_float(r)));
    };
    auto tournament = [&](const int (&x)[16], const int (&lo)[5], int a1, int b1, int b, u32 step) {
      int mid[5];
#pragma unroll
      for (int k = 0; k < 5; k++) mid[k] = med3(x[3 * k], x[3 * k + 1], x[3 * k + 2]);
      const int a2 = med3(lo[0], lo[1], lo[2]), b2 = med3(lo[3], lo[4], x[15]);
      const int mm1 = min(a1, b1);
      const int sl = min3i(max(a1, b1), a2, b2);
      const int m2 = min(min3i(mid[0], mid[1], mid[2]), min3i(mid[3], mid[4], sl));
      const int ob = __float_as_int(best[b]), os = __float_as_int(second[b]);
      best[b] = __int_as_float(min(ob, mm1 | (int)step));
      second[b] = __int_as_float(min3i(max(ob & ~1023, mm1), os, m2));
    };
    // minima of the five triples, of the two halves, of all sixteen
    auto minima = [&](const int (&x)[16], int (&lo)[5], int& a1, int& b1) {
#pragma unroll
      for (int k = 0; k < 5; k++) lo[k] = min3i(x[3 * k], x[3 * k + 1], x[3 * k + 2]);
      a1 = min3i(lo[0], lo[1], lo[2]);
      b1 = min3i(lo[3], lo[4], x[15]);
      return min(a1, b1);
    };
    u32 step = 0, j = t0;
    // Full steps of 32 train rows, software pipelined as in the key form.
    for (; j + 32 <= t1; j += 32, step++) {
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
          w = wn;
          wn = tr[2 * (size_t)min(j + 64 + c, last)];
          An[0] = expand_bits<FMT>(w.x);
          An[1] = expand_bits<FMT>(w.y);
          An[2] = expand_bits<FMT>(w.z);
          An[3] = expand_bits<FMT>(w.w);
        }
        asm volatile("" : "+v"(Dn));  // keeps the MFMAs above the test below (and their result in VGPRs)
        // x[i] as integers: D' is a positive float, its bit pattern orders like its value
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __float_as_int(D[i]);
        // smallest element of the lane: five triples and the sixteenth element
        int lo[5], a1, b1;
        const int m1 = minima(x, lo, a1, b1);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(m1 < __float_as_int(second[b])) != 0, 0)) {
          // Some lane has an element below its second best: the two smallest of the sixteen by
          // a tournament (the median of a triple is its second smallest, and the second
          // smallest of all is the second smallest of the triples' minima or one of their
          // medians), then merged into (best, second); the triples' minima come from the test.
          slow_steps++;
          tournament(x, lo, a1, b1, b, step);
        }
        D = Dn;
      }
#pragma unroll
      for (int m = 0; m < 4; m++) A[m] = An[m];
    }
    // The last, partial step of a range (once per scan, not pipelined; a second, masked copy of
    // the pipelined body cost 40 registers and a wave per SIMD): D holds its block 0, A its
    // operands; rows past the end are replaced by 8.0, farther than anything.
    if (j < t1) {
      const u32 left = t1 - j - 4 * h;      // the lane's rows at offsets < left exist ...
      const bool any_row = j + 4 * h < t1;  // ... if any does
#pragma unroll
      for (int b = 0; b < NB; b++) {
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const u32 off = (u32)((i & 3) + 8 * (i >> 2));
          x[i] = (any_row && off < left) ? __float_as_int(D[i]) : 0x41000000;
        }
        int lo[5], a1, b1;
        minima(x, lo, a1, b1);
        tournament(x, lo, a1, b1, b, step);
        if (b + 1 < NB) {
          D = zero;
#pragma unroll
          for (int m = 0; m < 4; m++) D = mfma_step<FMT>(A[m], B[b + 1][m], D);
        }
      }
    }
  }

  // Epilogue.  Everything it needs besides (best, second) is recomputed from q0,
  // laundered through an empty asm so that the compiler cannot hoist it above the scan and keep
  // it in registers there (it did: 16 address registers and col[] spilled around the loop).
  u32 q0e = q0;
  asm volatile("" : "+s"(q0e));
  count_steps(stats, lane, (t1 - t0 + 31) / 32 * NB, slow_steps);
  uint2* out = scratch + (size_t)split * split_stride + job.out_off;
  const u32 sentinel = d_lim >= 256u ? KEY_INIT : ((d_lim << IDX_BITS) | t0);
#pragma unroll
  for (int b = 0; b < NB; b++) {
    // the two halves of the wave hold the same query columns over disjoint train rows
    const u32 my_step = __float_as_uint(best[b]) & 1023u;
    const float mb = __uint_as_float(__float_as_uint(best[b]) & ~1023u);  // D' of the lane's best alone
    const float ob = __shfl_xor(mb, 32), os = __shfl_xor(second[b], 32);
    const float fb = fminf(mb, ob);
    const float fs = fminf(fminf(second[b], os), fmaxf(mb, ob));
    const u32 db = (u32)((fb - 2.0f) * 64.0f), ds = (u32)((fs - 2.0f) * 64.0f);
    const u32 qi = q0e + b * 32 + c;
    const u32 qc = min(qi, job.nq - 1);
    const u32 colb = job.qlist_off != NO_QLIST ? qlist[job.qlist_off + qc] : qc;
    u32 idx = 0xffffffffu;
    const bool mine = fb < dlim && mb == fb;  // this half reached the final best distance
    if (__builtin_amdgcn_ballot_w64(mine) != 0) {
      // index recovery: the lowest of the lane's 16 rows of that step at distance db
      const u32x4* qp = qbase + 2 * (size_t)(job.q_off + colb);
      const u32x4 qa = qp[0], qb = qp[1];
      const u32 base = t0 + 32 * (mine ? my_step : 0u) + 4 * h;
      const u32x4* tp = tbase + 2 * (size_t)job.t_off;
      // two groups of four consecutive rows (2 x 128 contiguous bytes) per iteration, all sixteen
      // loads in flight together; descending, so that the lowest row is assigned last and wins
#pragma unroll 1
      for (int g = 1; g >= 0; g--) {
        u32x4 ta[8], tb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const u32 rc = min(base + 16 * (u32)g + 8 * (u32)(k >> 2) + (u32)(k & 3), t1 - 1);
          ta[k] = tp[2 * (size_t)rc];
          tb[k] = tp[2 * (size_t)rc + 1];
        }
#pragma unroll
        for (int k = 7; k >= 0; k--) {
          const u32 r = base + 16 * (u32)g + 8 * (u32)(k >> 2) + (u32)(k & 3);
          const u32 d = __popc(qa.x ^ ta[k].x) + __popc(qa.y ^ ta[k].y) + __popc(qa.z ^ ta[k].z) + __popc(qa.w ^ ta[k].w) +
                        __popc(qb.x ^ tb[k].x) + __popc(qb.y ^ tb[k].y) + __popc(qb.z ^ tb[k].z) + __popc(qb.w ^ tb[k].w);
          if (mine && r < t1 && d == db) idx = r;
        }
      }
    }
    const u32 fidx = min(idx, (u32)__shfl_xor((int)idx, 32));
    const u32 kb = fb < dlim ? ((db << IDX_BITS) | fidx) : sentinel;
    const u32 ks = fs < dlim ? (ds << IDX_BITS) : sentinel;
    if (h == 0 && qi < job.nq) out[colb] = make_uint2(kb, ks);
    store_accept_mask(acc, job, lane, q0e + b * 32, h == 0 && qi < job.nq, kb, ks);
  }
}

// ---- Group tracking (TRACK 2): no branch, eleven instructions per step, no index recovery ----
// Both forms above pay for a wave-level skip test that real descriptors defeat: a running minimum
// over the 62 steps of a 2000-row scan improves ~5 times per lane, a wave has 64 lanes, so some
// lane has news in nearly every step and the whole wave leaves the common path (92 % of the steps
// on the real-like database, 8.6 T comparisons/s).  This tile does the same work in EVERY step and
// makes that work small.
//  * The accumulators start from sixteen REGISTERS, element i at 6.0 + i ulp, instead of the
//    inline constant: D'[i] = 4 + d / 64 + i * 2^-21.  All of [4, 8) is one binade, so the bit
//    pattern of a result is 0x40800000 | d << 15 | i: the distance in bits 15-22, the element
//    number -- which orders like the train row, (i & 3) + 8 (i >> 2) + 4 h -- in bits 0-3, every
//    partial sum a multiple of 2^-21 below 8, exact in f32.  The eight v_min3 that find the
//    smallest of a lane's sixteen results now find (distance, lowest row) at once.
//  * A lane's sixteen results of a step are a GROUP; the lane keeps the two smallest group
//    minima: second = med3(best, m, second), best = min(best, m | step << 4) -- three more
//    instructions, no compare, no branch; `best` orders like (distance, step, row), i.e. like the
//    reference's strict '<' in an ascending scan (keypoints.h:273), and its train index is read off
//    its bits at the end.
//  * What this leaves out: the second smallest of ALL values is either the second smallest group
//    minimum -- tracked here -- or the second smallest value INSIDE the best's group (any other
//    group contributes at most its minimum).  The scan therefore stores an upper bound of the
//    second best distance that is exact unless the true one sits in the 32-row step of the best;
//    the consumers (refine_second, vsmatch_kernels.h) look at those 32 rows again with v_xor /
//    v_bcnt -- and only for queries the accept test would otherwise pass, a few per image pair.
// Tag bits make `best` / `second` up to 2^14 units in the last place larger than the pure D'; a
// distance unit is 2^15, so order between different distances is untouched.
constexpr int GROUP_D_BASE = 0x40800000;  // bits of 4.0 = D' of distance 0
constexpr int GROUP_D_256 = 0x41000000;   // bits of 8.0 = D' of distance 256
template <int NB, int FMT>
__device__ __forceinline__ void mfma_scan_tile_g(const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase,
                                                 const Job& job, u32 q0, u32 lane, u32 split, u32 splits,
                                                 u32 split_stride, uint2* __restrict__ scratch,
                                                 const u32* __restrict__ qlist, u32 d_lim, const AcceptMask& acc) {
  const u32 per = (job.nt + splits - 1) / splits;
  const u32 t0 = min(split * per, job.nt), t1 = min(t0 + per, job.nt);
  const u32 c = lane & 31, h = lane >> 5;

  u32 m88 = 0x88888888u, c22 = 0x22222222u;
  asm volatile("" : "+s"(m88), "+v"(c22));
  i32x8 B[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const u32 qi = min(q0 + b * 32 + c, job.nq - 1);
    const u32 colb = job.qlist_off != NO_QLIST ? qlist[job.qlist_off + qi] : qi;
    const u32x4 w = qbase[2 * (size_t)(job.q_off + colb) + h];
    B[b][0] = expand_bits_r(~w.x, m88, c22);
    B[b][1] = expand_bits_r(~w.y, m88, c22);
    B[b][2] = expand_bits_r(~w.z, m88, c22);
    B[b][3] = expand_bits_r(~w.w, m88, c22);
  }
  const int init = d_lim >= 256u ? GROUP_D_256 : GROUP_D_BASE | (int)(d_lim << 15);
  int best[NB], second[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) best[b] = second[b] = init;

  if (t1 > t0) {
    const u32x4* tr = tbase + 2 * (size_t)job.t_off + h;
    const u32 last = t1 - 1;
    // start values of the accumulators: 6.0 + i ulp, kept in registers for the whole scan
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; i++) zero[i] = __int_as_float(0x40C00000 + i);
    asm volatile("" : "+v"(zero));
    u32x4 w = tr[2 * (size_t)min(t0 + c, last)];
    u32x4 wn = tr[2 * (size_t)min(t0 + 32 + c, last)];
    i32x8 A[4];
    A[0] = expand_bits_r(w.x, m88, c22);
    A[1] = expand_bits_r(w.y, m88, c22);
    A[2] = expand_bits_r(w.z, m88, c22);
    A[3] = expand_bits_r(w.w, m88, c22);
    f32x16 D = zero;
#pragma unroll
    for (int m = 0; m < 4; m++) D = mfma_step<FMT>(A[m], B[0][m], D);
    i32x8 An[4] = {A[0], A[1], A[2], A[3]};
    auto med3 = [](int p, int q, int r) {
      return __float_as_int(__builtin_amdgcn_fmed3f(__int_as_float(p), __int_as_float(q), __int_as_float(r)));
    };
    // group minimum: five triples and the sixteenth element
    auto group_min = [](const int (&x)[16]) {
      const int m0 = min3i(x[0], x[1], x[2]), m1 = min3i(x[3], x[4], x[5]), m2 = min3i(x[6], x[7], x[8]);
      const int m3 = min3i(x[9], x[10], x[11]), m4 = min3i(x[12], x[13], x[14]);
      return min(min3i(m0, m1, m2), min3i(m3, m4, x[15]));
    };
    u32 step16 = 0, j = t0;  // step16 = 16 x the number of the 32-row step
    for (; j + 32 <= t1; j += 32, step16 += 16) {
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
          w = wn;
          wn = tr[2 * (size_t)min(j + 64 + c, last)];
          An[0] = expand_bits_r(w.x, m88, c22);
          An[1] = expand_bits_r(w.y, m88, c22);
          An[2] = expand_bits_r(w.z, m88, c22);
          An[3] = expand_bits_r(w.w, m88, c22);
        }
        asm volatile("" : "+v"(Dn));  // keeps the MFMAs above the minimum tree (and their result in VGPRs)
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __float_as_int(D[i]);
        const int m = group_min(x);
        second[b] = med3(best[b], m, second[b]);
        best[b] = min(best[b], m | (int)step16);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        D = Dn;
      }
#pragma unroll
      for (int m = 0; m < 4; m++) A[m] = An[m];
    }
    // The last, partial step of a range (once per scan, not pipelined): D holds its block 0, A
    // its operands; rows past the end are replaced by distance 256, which is never recorded.
    if (j < t1) {
      const u32 left = t1 - j - 4 * h;      // the lane's rows at offsets < left exist ...
      const bool any_row = j + 4 * h < t1;  // ... if any does
#pragma unroll
      for (int b = 0; b < NB; b++) {
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const u32 off = (u32)((i & 3) + 8 * (i >> 2));
          x[i] = (any_row && off < left) ? __float_as_int(D[i]) : GROUP_D_256;
        }
        const int m = group_min(x);
        second[b] = med3(best[b], m, second[b]);
        best[b] = min(best[b], m | (int)step16);
        if (b + 1 < NB) {
          D = zero;
#pragma unroll
          for (int m2 = 0; m2 < 4; m2++) D = mfma_step<FMT>(A[m2], B[b + 1][m2], D);
        }
      }
    }
  }

  // Epilogue (q0 laundered so that nothing of it is hoisted above the scan, see the distance form):
  // bits -> packed keys, the two halves of the wave (same queries, disjoint rows) merged exactly.
  u32 q0e = q0;
  asm volatile("" : "+s"(q0e));
  uint2* out = scratch + (size_t)split * split_stride + job.out_off;
  const u32 sentinel = d_lim >= 256u ? KEY_INIT : ((d_lim << IDX_BITS) | t0);
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const u32 bb = (u32)best[b], sb = (u32)second[b];
    const u32 i = bb & 15u;
    const u32 row = t0 + 2 * ((bb >> 4) & 1023u) * 16 + 4 * h + (i & 3u) + 8 * (i >> 2);
    const u32 kl = (int)bb < init ? ((((bb >> 15) & 255u) << IDX_BITS) | row) : sentinel;
    const u32 sl = (int)(sb & ~32767u) < init ? (((sb >> 15) & 255u) << IDX_BITS) : sentinel;
    const u32 ko = (u32)__shfl_xor((int)kl, 32), so = (u32)__shfl_xor((int)sl, 32);
    const u32 kb = min(kl, ko), ks = min(min(sl, so), max(kl, ko));
    const u32 qi = q0e + b * 32 + c;
    const u32 qc = min(qi, job.nq - 1);
    const u32 colb = job.qlist_off != NO_QLIST ? qlist[job.qlist_off + qc] : qc;
    if (h == 0 && qi < job.nq) out[colb] = make_uint2(kb, ks);
    store_accept_mask(acc, job, lane, q0e + b * 32, h == 0 && qi < job.nq, kb, ks);
  }
}

// WAVES == 1: same grid as scan_kernel_adaptive<*, 1, *> (one-wave workgroups): tiles of
// 32 * NBMAX queries and remainder tiles of half and a quarter of that, grouped or job-minor
// block order.  WAVES == 4: same grid as scan_kernel (4-wave workgroups, job-minor), used with
// NBMAX == 2 for the pruned backward pass, most of whose workgroups find nothing to do.
template <int FMT, int NBMAX, int WAVES = 1, int TRACK = 0>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2))) void mfma_scan_kernel(const u32x4* __restrict__ qbase,
                                                               const u32x4* __restrict__ tbase,
                                                               const Job* __restrict__ jobs, u32 n_jobs, u32 splits,
                                                               u32 split_stride, uint2* __restrict__ scratch,
                                                               const u32* __restrict__ qlist, u32 tiles_per_job,
                                                               u32 d_lim, unsigned long long* __restrict__ stats,
                                                               AcceptMask acc) {
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
    if (TRACK == 2) mfma_scan_tile_g<NBMAX, FMT>(qbase, tbase, job, t * 64 * RMAX, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, acc);
    else if (TRACK) mfma_scan_tile_d<NBMAX, FMT>(qbase, tbase, job, t * 64 * RMAX, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, stats, acc);
    else mfma_scan_tile<NBMAX, FMT>(qbase, tbase, job, t * 64 * RMAX, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, stats, acc);
    return;
  }
  u32 k = t - full, q0 = full * 64 * RMAX;
  if (RMAX > 2 && (rem & 2)) {
    if (k == 0) {
      if (TRACK == 2) mfma_scan_tile_g<4, FMT>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, acc);
      else if (TRACK) mfma_scan_tile_d<4, FMT>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, stats, acc);
      else mfma_scan_tile<4, FMT>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, stats, acc);
      return;
    }
    k--;
    q0 += 128;
  }
  if ((rem & 1) && k == 0) {
    if (TRACK == 2) mfma_scan_tile_g<2, FMT>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, acc);
    else if (TRACK) mfma_scan_tile_d<2, FMT>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, stats, acc);
    else mfma_scan_tile<2, FMT>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, stats, acc);
  }
}

}  // namespace vsm
