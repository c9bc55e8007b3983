// mfma_scan.h -- the brute-force scan of visnav::matchDescriptors on the gfx950 matrix cores.
//
// Path: /root/reference/include/visnav/keypoints.h:256-290 (and 305-339): per query the
// nearest and second nearest train descriptor by Hamming distance over 256 bits.
//
// The Hamming distance of two bit strings is a dot product: with a -> +1/-1 per bit,
//     sum_k a_k * b_k = (#equal bits) - (#different bits) = 256 - 2 d.
// v_mfma_f32_32x32x64_f8f6f4 on FP4 (e2m1) operands multiplies a 32 x 64 tile by a 64 x 32
// tile in 32 cycles per SIMD, so four of them give the 1024 distances of 32 train x 32 query
// descriptors in 128 cycles; the popcount path (v_xor + v_bcnt, vsmatch_kernels.h) needs
// 16 x 55.4 = 887 cycles for the same 1024 comparisons.  Every product is +1/4 or -1/4 and every sum
// a multiple of 1/4 of magnitude <= 64, all exact in the f32 accumulator: the distances are the
// reference's, bit for bit.
//
// Operands.  Bits become FP4 codes:
//     code = 0x1 | (bit << 3)        (0x1 = +0.5, 0x9 = -0.5 in e2m1)
//     reg s = ((w << s) & 0x88888888) | 0x11111111,  s = 0..3
// i.e. register s takes bits 3-s, 7-s, ... 31-s of the 32-bit word w.  That is a permutation of
// the 32 k-positions, and a dot product does not care as long as both operands use the same
// one.  Queries are expanded once per tile (from ~w, so that a product is +1 where the bits
// DIFFER), train descriptors once per 32 x (32 NB) block of comparisons.  The instruction is the one
// WITHOUT block scales (the builtin with both scale operands 0 selects it).  Rounds 1-2 used the
// block-scaled form, codes +-1.0 scaled by 2^-7 into the binade [4, 8): that form is two instructions
// to the hardware (v_mfma_ld_scale_b32 + the MFMA) and costs ~37 cycles of the pipe instead of 32 --
// back to back on random codes 14.5 T comparisons/s against 17.4 T without scales (tools/mfma_rate.hip);
// the scan went from 14.1-14.3 to 14.3-14.6 T (+2 %, same box: what bounds it now is the power the
// VALU work beside the MFMAs takes -- 1.92 GHz against 2.12 GHz of the bare instruction).
//
// Results and tracking ("group form").
//  * The accumulators start from sixteen REGISTERS, element i at 192.0 + i ulp:
//    D'[i] = 192 + (2 d - 256) / 4 + i * 2^-16 = 128 + d / 2 + i * 2^-16.  All of [128, 256) is one
//    binade, so the bit pattern of a result is 0x43000000 | d << 15 | i: the distance in bits
//    15-22, the element number -- which orders like the train row, (i & 3) + 8 (i >> 2) + 4 h --
//    in bits 0-3; every partial sum is a multiple of 2^-16 in [128, 256], exact in f32.  Eight
//    v_min3_i32 over the sixteen results of a lane give (distance, lowest row) at once.
//  * A lane of the 32 x 32 result holds 16 train rows of ONE query: a GROUP.  Per step the lane
//    keeps the two smallest group minima: second = med3(best, m, second), best = min(best,
//    m | step << 4) -- three more instructions, no compare, no branch.  `best` orders like
//    (distance, step, row), i.e. like the reference's strict '<' in an ascending scan
//    (keypoints.h:273), and its train index is read off its bits at the end.  Tag bits make
//    `best` / `second` up to 2^14 units in the last place larger than the pure D'; a distance unit
//    is 2^15, so order between different distances is untouched.
//  * What this leaves out: the second smallest of ALL values is either the second smallest group
//    minimum -- tracked here -- or the second smallest value INSIDE the best's group (any other
//    group contributes at most its minimum).  The scan therefore stores an upper bound of the
//    second best distance that is exact unless the true one sits among the best's fifteen group
//    mates; the consumers (refine_second, vsmatch_kernels.h) look at those rows again with
//    v_xor / v_bcnt -- only for queries the accept test would otherwise pass, a few per image pair.
//  * The start value of (best, second) is distance `d_lim`: the matcher passes the smallest
//    distance that can no longer change the accept test (keypoints.h:295-296; accept_limit in
//    vsmatch.hip), the raw 2-NN entry point passes 256.  Distance 256 is never recorded
//    (keypoints.h:259-262): its D' is 256.0, not below any start value.
//
// History (round 1, tools/history/mfma_scan_forms.h, timed against this form by tools/mfma_scan.hip):
// float keys d * 32768 + j behind a wave-level skip test (fast only while no lane of the wave
// sees a candidate: iid bits 12.4 T comparisons/s, real ORB descriptors 5.3 T) and a distance
// tournament with index recovery by v_bcnt at the end (11.1 / 8.4 T).  A running minimum over the
// 62 steps of a 2000-row scan improves ~5 times per lane and a wave has 64 lanes, so on real
// descriptors some lane has news in nearly every step: any wave-level test is defeated, the work
// has to be the same in every step -- and small.  This form: 12.5 T on either kind of data, 7.4 T
// on 333-descriptor images (3.9 T before).
// (Tried again in round 2, no gain: train codes 0.0 / 1.0 instead of +-1.0 -- half-zero operands
// draw less power and the bare instruction runs 10 % faster on them (tools/mfma_rate.hip), but this
// kernel 2.4 % slower, 13.55 against 13.87 T; the offset pop(q) would ride in the start distance.)
// (Tried in round 2, much slower: FP4 codes pre-expanded in HBM (128 B per descriptor, loaded
// instead of expanded: 14.75 -> 11.25 VALU instructions per step) -- the four-fold operand traffic
// from L2 costs more than the VALU work it saves: 10.1 against 13.9 T at 256-query tiles, 8.8
// against 13.3 T at 128, 4.6 against 12.0 T at 64.)
// Build with -mllvm -amdgpu-mfma-vgpr-form: the minimum tree reads the results with VALU
// instructions, from AGPRs every element would first need a v_accvgpr_read.
#pragma once
#include "vsmatch_kernels.h"

namespace vsm {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32-bit word -> four registers of FP4 codes.  The two masks come in REGISTERS (0x88888888 in an
// SGPR, 0x11111111 in a VGPR): gfx950's three-operand encoding takes no literal, so with literal
// masks the compiler emits v_and + v_or per register (44 instructions per 32 train rows with the
// shifts); from registers it is one v_and_or_b32 (28).
__device__ __forceinline__ i32x8 expand_bits_r(u32 w, u32 m88, u32 c22) {
  i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 4; s++) v[s] = (int)(((w << s) & m88) | c22);
  return v;
}

// CODES 1, the shift-free "zero-rich" train codes (round 3).  A train code need not be symmetric: with the
// QUERY code of a bit symmetric (+-g), a train bit may take any two values {mu - delta, mu + delta}: the sum
// over k of query x train is then  sum_k (+-g mu)  +  g delta (256 - 2 d)  -- a constant of the QUERY (which
// rides in the lane's start value of best / second, a lane of the result holds one query) plus the distance,
// as long as g delta is the same for every bit.  e2m1 has such pairs for every bit of a nibble WITHOUT moving
// the bit: bit 3 (the sign) of code 0x1: +0.5 / -0.5; bit 2 alone: 0 / 2.0; bit 1 alone: 0 / 1.0; bit 0 alone:
// 0 / 0.5 -- with query codes +-1.0, +-0.5, +-1.0, +-2.0 for the four bit classes every product of differing
// bits is +0.5 (or 0 and +-0.5 / +-1 ... the constant part), g delta = 0.5.  So register s of a train word is
// just  w & (0x88888888 >> s)  (| 0x11111111 for s = 0): 16 instructions per 32 train rows instead of 28, and
// three eighths of the train codes are ZERO -- less switching in the multipliers, which is what holds the clock
// down under this kernel.  Result: D' = 768 + i ulp + C_q + d - 128 with C_q = 96 - P, P = the number of one
// bits of the query outside the sign class (w & 0x77777777): everything in [544, 992], inside the binade
// [512, 1024) whose ulp is 2^-14; every partial sum is a multiple of 0.5 (+ the tag) of magnitude <= 224
// around 768: exact.  Bit pattern: Z_q + (d << 14 | tag), Z_q = 0x44000000 + ((224 - P) << 14).
__device__ __forceinline__ i32x8 expand_train_zr(u32 w, u32 m88, u32 c11, u32 m44, u32 m22, u32 m11) {
  i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  v[0] = (int)((w & m88) | c11);
  v[1] = (int)(w & m44);
  v[2] = (int)(w & m22);
  v[3] = (int)(w & m11);
  return v;
}
// the query side of CODES 1 (once per tile): sign class 0x2 | ~bit << 3 (+-1.0), then 0x1 (0.5), 0x2 (1.0),
// 0x4 (2.0) | bit << 3
__device__ __forceinline__ i32x8 expand_query_zr(u32 w) {
  i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  v[0] = (int)((~w & 0x88888888u) | 0x22222222u);
  v[1] = (int)(((w << 1) & 0x88888888u) | 0x11111111u);
  v[2] = (int)(((w << 2) & 0x88888888u) | 0x22222222u);
  v[3] = (int)(((w << 3) & 0x88888888u) | 0x44444444u);
  return v;
}

// cbsz = blgp = 4: FP4 e2m1 operands (FP8 e4m3 runs at half the rate with twice the registers:
// 5.9 against 9.6 T with the first version of this kernel).  Scale operands 0 / 0: the compiler emits
// v_mfma_f32_32x32x64_f8f6f4, no block scales, so a product is +-1/4.
__device__ __forceinline__ f32x16 mfma_step(const i32x8 a, const i32x8 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
}
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }

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

constexpr int GROUP_D_BASE = 0x43000000;  // bits of 128.0 = D' of distance 0
constexpr int GROUP_D_256 = 0x43800000;   // bits of 256.0 = D' of distance 256
constexpr int GROUP_D_START = 0x43400000; // bits of 192.0 = the accumulators' start value (+ i ulp)
constexpr int ZR_START = 0x44400000;      // CODES 1: bits of 768.0
constexpr int ZR_BASE = 0x44000000 + (224 << 14);  // CODES 1: D' of distance 0 for a query with P = 0
constexpr int NEVER = 0x7f000000;         // a result that is never below a start value (rows past the end)
// One tile = 32 * NB queries of a job (lane l: column l & 31 of each block of 32, half l >> 5 of
// the k range) against the split's share of the train range, 32 rows at a time.  Software
// pipeline over (32 train rows, query block) steps: the four MFMAs of the next step are issued
// before the minimum tree of the current one (two accumulator sets); sched_group_barriers
// interleave them, three VALU instructions behind each MFMA.
//
// A RUN: the wave goes on with the jobs j0 + 1, j0 + 2, ... < j1 for as long as they have the SAME
// query set as job j0 (consecutive pairs of an all-pairs list share their first image, sfm.cpp:1285-1290;
// so do the candidate pairs of match_bow), keeping the expanded query operands in registers: the
// wave launch, the load of the queries and their expansion (224 instructions at NB = 8) are paid once
// per run instead of once per job.  Returns the first job it did not do.  RUNS = false compiles the
// loop away (one job per call): the loop costs registers (NB 8: 237 -> 252, still two waves per
// SIMD; NB 4: 166 -> 190 and NB 2: 117 -> 140 would each lose a wave per SIMD), so only the
// 256-query-tile kernel of the batched scan has it.
template <int NB, bool RUNS, int CODES>
__device__ __forceinline__ u32 mfma_scan_run(const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase,
                                             const Job* __restrict__ jobs, u32 j0, u32 j1, u32 q0, u32 lane,
                                             u32 split, u32 splits, u32 split_stride, uint2* __restrict__ scratch,
                                             const u32* __restrict__ qlist, u32 d_lim, const AcceptMask& acc) {
  Job job = jobs[j0];
  const u32 c = lane & 31, h = lane >> 5;

  constexpr int DSH = CODES ? 14 : 15;  // the distance's place in a result's bits
  u32 m88 = 0x88888888u, c22 = 0x11111111u;  // (the name is history: codes 0x1 / 0x9 = +-0.5)
  u32 m44 = 0x44444444u, m22 = 0x22222222u, m11 = 0x11111111u;
  asm volatile("" : "+s"(m88), "+v"(c22));
  if (CODES) asm volatile("" : "+s"(m44), "+s"(m22), "+s"(m11));
  auto expand_train = [&](u32 w) { return CODES ? expand_train_zr(w, m88, c22, m44, m22, m11) : expand_bits_r(w, m88, c22); };
  i32x8 B[NB][4];
  // D' of distance 0 for the block's query of this lane: CODES 0 the same for every query; CODES 1
  // ZR_BASE - (P << 14), the P of four blocks packed into one register (they live through the whole scan)
  u32 ppack[(NB + 3) / 4] = {};
  auto zq = [&](int b) { return CODES ? ZR_BASE - (int)(((ppack[b >> 2] >> (8 * (b & 3))) & 255u) << 14) : GROUP_D_BASE; };
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const u32 qi = min(q0 + b * 32 + c, job.nq - 1);
    const u32 colb = job.qlist_off != NO_QLIST ? qlist[job.qlist_off + qi] : qi;
    const u32x4 w = qbase[2 * (size_t)(job.q_off + colb) + h];
    if (CODES) {
      B[b][0] = expand_query_zr(w.x);
      B[b][1] = expand_query_zr(w.y);
      B[b][2] = expand_query_zr(w.z);
      B[b][3] = expand_query_zr(w.w);
      const int ph = __popc(w.x & 0x77777777u) + __popc(w.y & 0x77777777u) + __popc(w.z & 0x77777777u) + __popc(w.w & 0x77777777u);
      ppack[b >> 2] |= (u32)(ph + __shfl_xor(ph, 32)) << (8 * (b & 3));
    } else {
      B[b][0] = expand_bits_r(~w.x, m88, c22);
      B[b][1] = expand_bits_r(~w.y, m88, c22);
      B[b][2] = expand_bits_r(~w.z, m88, c22);
      B[b][3] = expand_bits_r(~w.w, m88, c22);
    }
  }
  const int dl = (int)(min(d_lim, 256u) << DSH);  // start value of (best, second) above zq
  u32 j = j0;
  for (;;) {
  const u32 per = (job.nt + splits - 1) / splits;
  const u32 t0 = min(split * per, job.nt), t1 = min(t0 + per, job.nt);
  int best[NB], second[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) best[b] = second[b] = zq(b) + dl;

  if (t1 > t0) {
    // train rows: a wave-uniform base and a 32-bit byte offset per lane (row c of the step, half h),
    // clamped to the last row of the range (rows past the end are loaded again, never used)
    const char* tr = (const char*)(tbase + 2 * (size_t)job.t_off);
    const u32 last_off = (t1 - 1) * 32u + h * 16u;
    auto row = [&](u32 off) { return *(const u32x4*)(tr + min(off, last_off)); };
    u32 off = (t0 + c) * 32u + h * 16u;  // this lane's row of the current step
    // start values of the accumulators: 192.0 + i ulp, kept in registers for the whole scan
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; i++) zero[i] = __int_as_float((CODES ? ZR_START : GROUP_D_START) + i);
    asm volatile("" : "+v"(zero));
    u32x4 w = row(off);
    u32x4 wn = row(off + 1024u);  // the step after: in flight while the current one is expanded
    off += 2048u;
    i32x8 A[4];
    A[0] = expand_train(w.x);
    A[1] = expand_train(w.y);
    A[2] = expand_train(w.z);
    A[3] = expand_train(w.w);
    f32x16 D = zero;
#pragma unroll
    for (int m = 0; m < 4; m++) D = mfma_step(A[m], B[0][m], D);
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
    u32 step16 = 0, tr0 = t0;  // step16 = 16 x the number of the 32-row step, tr0 = its first train row
    for (; tr0 + 32 <= t1; tr0 += 32, step16 += 16) {
#pragma unroll
      for (int b = 0; b < NB; b++) {
        f32x16 Dn = zero;
        auto expand_next = [&]() {  // the next step's train rows; the step after goes in flight
          An[0] = expand_train(wn.x);
          An[1] = expand_train(wn.y);
          An[2] = expand_train(wn.z);
          An[3] = expand_train(wn.w);
          wn = row(off);
          off += 1024u;
        };
        if (NB == 1) expand_next();  // a one-block tile: its only block needs them now
        if (b + 1 < NB) {
#pragma unroll
          for (int m = 0; m < 4; m++) Dn = mfma_step(A[m], B[b + 1][m], Dn);
        } else {
#pragma unroll
          for (int m = 0; m < 4; m++) Dn = mfma_step(An[m], B[0][m], Dn);
        }
        if (NB >= 2 && b == NB - 2) expand_next();  // one block early, behind the MFMAs of the last block
        asm volatile("" : "+v"(Dn));  // keeps the MFMAs above the minimum tree (and their result in VGPRs)
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __float_as_int(D[i]);
        const int m = group_min(x);
        second[b] = med3(best[b], m, second[b]);
        best[b] = min(best[b], m | (int)step16);
        // issue order of a step: each of the four MFMAs followed by three of the step's ~15 VALU
        // instructions (tools/mfma_scan.hip -DVSM_PRODUCT: 2+4+1+4+1, 1+4 x 4, 1+2 x 4, 2+6 x 2 are
        // all within the run-to-run noise of +-1.5 %; this one needs the fewest registers, 237)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        D = Dn;
      }
#pragma unroll
      for (int m = 0; m < 4; m++) A[m] = An[m];
    }
    // The last, partial step of a range (once per scan, not pipelined): D holds its block 0, A
    // its operands; rows past the end are replaced by distance 256, which is never recorded.
    if (tr0 < t1) {
      const u32 left = t1 - tr0 - 4 * h;      // the lane's rows at offsets < left exist ...
      const bool any_row = tr0 + 4 * h < t1;  // ... if any does
#pragma unroll
      for (int b = 0; b < NB; b++) {
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const u32 off = (u32)((i & 3) + 8 * (i >> 2));
          x[i] = (any_row && off < left) ? __float_as_int(D[i]) : NEVER;
        }
        const int m = group_min(x);
        second[b] = med3(best[b], m, second[b]);
        best[b] = min(best[b], m | (int)step16);
        if (b + 1 < NB) {
          D = zero;
#pragma unroll
          for (int m2 = 0; m2 < 4; m2++) D = mfma_step(A[m2], B[b + 1][m2], D);
        }
      }
    }
  }

  // Epilogue.  q0 is laundered through an empty asm so that the compiler cannot hoist what it
  // needs above the scan and keep it in registers there:
  // bits -> packed keys, the two halves of the wave (same queries, disjoint rows) merged exactly.
  u32 q0e = q0;
  asm volatile("" : "+s"(q0e));
  if (CODES) {  // likewise the start values: unpacked again here, not carried through the scan
#pragma unroll
    for (int k = 0; k < (NB + 3) / 4; k++) asm volatile("" : "+v"(ppack[k]));
  }
  uint2* out = scratch + (size_t)split * split_stride + job.out_off;
  const u32 sentinel = d_lim >= 256u ? KEY_INIT : ((d_lim << IDX_BITS) | t0);
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const u32 bb = (u32)best[b], sb = (u32)second[b];
    const int zb = zq(b), init = zb + dl;
    const u32 i = bb & 15u;
    const u32 row = t0 + 2 * ((bb >> 4) & 1023u) * 16 + 4 * h + (i & 3u) + 8 * (i >> 2);
    const u32 kl = (int)bb < init ? (((((bb - (u32)zb) >> DSH) & 255u) << IDX_BITS) | row) : sentinel;
    const u32 sl = (int)(sb & ~((1u << DSH) - 1u)) < init ? ((((sb - (u32)zb) >> DSH) & 255u) << IDX_BITS) : sentinel;
    const u32 ko = (u32)__shfl_xor((int)kl, 32), so = (u32)__shfl_xor((int)sl, 32);
    const u32 kb = min(kl, ko), ks = min(min(sl, so), max(kl, ko));
    const u32 qi = q0e + b * 32 + c;
    const u32 qc = min(qi, job.nq - 1);
    const u32 colb = job.qlist_off != NO_QLIST ? qlist[job.qlist_off + qc] : qc;
    if (h == 0 && qi < job.nq) out[colb] = make_uint2(kb, ks);
    store_accept_mask(acc, job, lane, q0e + b * 32, h == 0 && qi < job.nq, kb, ks);
  }
  // the next job of the run, if it scans for the same queries
  // (fetching the next job's first rows behind this job's last step, so that they are there when
  // the epilogue is done, changed nothing measurable and took the last four registers)
  if (!RUNS || ++j >= j1) break;
  const Job nx = jobs[j];
  if (nx.q_off != job.q_off || nx.nq != job.nq || nx.qlist_off != NO_QLIST || job.qlist_off != NO_QLIST) break;
  job = nx;
  }
  return RUNS ? j : j0 + 1;
}

// The grid counts queries in units of 64, a tile in blocks of 32: when the last block of a job's last
// tile holds no query at all (2000 queries: the eighth 256-query tile has 208, six and a half blocks)
// the tile runs with one block less -- 1/64 of the MFMA work of a 2000-query job.
template <int NB, bool RUNS, int CODES>
__device__ __forceinline__ u32 mfma_scan_run_fit(const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase,
                                                 const Job* __restrict__ jobs, u32 j0, u32 j1, u32 nq, u32 q0, u32 lane,
                                                 u32 split, u32 splits, u32 split_stride, uint2* __restrict__ scratch,
                                                 const u32* __restrict__ qlist, u32 d_lim, const AcceptMask& acc) {
  if (NB > 1 && q0 + (NB - 1) * 32 >= nq)
    return mfma_scan_run<(NB > 1 ? NB - 1 : 1), RUNS, CODES>(qbase, tbase, jobs, j0, j1, q0, lane, split, splits, split_stride, scratch, qlist, d_lim, acc);
  return mfma_scan_run<NB, RUNS, CODES>(qbase, tbase, jobs, j0, j1, q0, lane, split, splits, split_stride, scratch, qlist, d_lim, acc);
}

// WAVES == 1: one-wave workgroups, the grid of scan_kernel_adaptive<*, 1, *>: tiles of 32 * NBMAX
// queries and remainder tiles of half and a quarter of that, grouped or job-minor block order
// (see there).  WAVES == 4: the grid of scan_kernel (4-wave workgroups, job-minor), used with
// NBMAX == 2 for one stereo pair and for the pruned backward pass, most of whose workgroups find
// nothing to do.  Registers: 252 (NBMAX 8 with the job loop of mfma_scan_run, 237 without: two waves
// per SIMD), 166 (4: three), 117 (2: four).
#ifndef VSM_SCAN_CODES
#define VSM_SCAN_CODES 0
#endif
template <int NBMAX, int WAVES = 1, int CODES = VSM_SCAN_CODES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2))) void mfma_scan_kernel(
    const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase, const Job* __restrict__ jobs, u32 n_jobs,
    u32 splits, u32 split_stride, uint2* __restrict__ scratch, const u32* __restrict__ qlist, u32 tiles_per_job,
    u32 d_lim, AcceptMask acc_in, u32 jobs_per_wave) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  // The accept table of the epilogue in LDS: a global load there cannot move above the stores of the block before it
  // (they might alias), so the table lookups of a tile's blocks waited for memory one after the other.
  __shared__ uint16_t s_lut[264];
  const AcceptMask acc{s_lut, acc_in.threshold, acc_in.mask};  // (always the LDS copy: read only when there is a mask)
  if (acc_in.mask) {
    for (u32 k = threadIdx.x; k < 257; k += 64 * WAVES) s_lut[k] = acc_in.lut[k];
    __syncthreads();
  }
  // a wave's jobs [j0, j1) (WAVES == 1: jobs_per_wave consecutive ones, see mfma_scan_run) and the
  // tile t it does of each
  u32 j0, j1, t;
  if (WAVES == 1 && tiles_per_job) {
    const u32 per_super = 8 * tiles_per_job;  // eight groups of jobs_per_wave jobs: one per XCD
    const u32 super = blockIdx.x / per_super, r = blockIdx.x - super * per_super;
    t = r >> 3;
    j0 = (super * 8 + (r & 7)) * jobs_per_wave;
    j1 = min(j0 + jobs_per_wave, n_jobs);
  } else if (WAVES == 1) {
    const u32 n_groups = (n_jobs + jobs_per_wave - 1) / jobs_per_wave;
    t = blockIdx.x / n_groups;
    j0 = (blockIdx.x - t * n_groups) * jobs_per_wave;
    j1 = min(j0 + jobs_per_wave, n_jobs);
  } else {
    const u32 wg_of_job = blockIdx.x / n_jobs;
    j0 = blockIdx.x - wg_of_job * n_jobs;
    j1 = j0 + 1;
    t = wg_of_job * WAVES + wave;
  }
  constexpr u32 RMAX = NBMAX / 2;  // tiles are counted in units of 64 queries, as adaptive_tiles does
  constexpr bool RUNS = WAVES == 1 && NBMAX == 8;  // else jobs_per_wave is 1 (the host sees to it)
  // tile t of job j (and of the jobs after it that share its queries); returns the next job
  auto tile_of = [&](u32 j) -> u32 {
    const u32 nq = jobs[j].nq;
    const u32 n64 = (nq + 63) >> 6, full = n64 / RMAX, rem = n64 & (RMAX - 1);
    if (t < full) {
      return mfma_scan_run_fit<NBMAX, RUNS, CODES>(qbase, tbase, jobs, j, j1, nq, t * 64 * RMAX, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, acc);
    }
    u32 k = t - full, q0 = full * 64 * RMAX;
    if (RMAX > 2 && (rem & 2)) {
      if (k == 0)
        return mfma_scan_run_fit<4, RUNS, CODES>(qbase, tbase, jobs, j, j1, nq, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, acc);
      k--;
      q0 += 128;
    }
    if ((rem & 1) && k == 0)
      return mfma_scan_run_fit<2, RUNS, CODES>(qbase, tbase, jobs, j, j1, nq, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist, d_lim, acc);
    return j + 1;  // this job has no tile t
  };
  if (RUNS) {
    for (u32 j = j0; j < j1;) j = tile_of(j);
  } else if (j0 < j1) {
    tile_of(j0);
  }
}

}  // namespace vsm
