// mfma16_scan.h -- the forward scan of the batched matcher on v_mfma_f32_16x16x128_f8f6f4 (round 3).
//
// Path: /root/reference/include/visnav/keypoints.h:256-290, as mfma_scan.h -- same operands (the shift-free
// zero-rich codes, CODES 1 there), same tracking ("group form"), same results in the scratch array, bit for bit;
// what differs is the shape of the instruction.
//
// Why.  Under the scan the chip is power limited: every form of the 32x32x64 kernel sits at MFMA-pipe-busy x clock
// = 1.57-1.60 GHz-equivalents (0.91 busy at 1.75 GHz; profiles/r03/scan_shapes_pmc.txt).  The 16x16x128 shape does
// the same multiply-adds with half the accumulator traffic (256 sums of 128 products per instruction instead of 1024
// of 64) and holds a higher clock for the same work: bare, on the scan's codes, 17.5 T comparisons/s at 2.33 GHz
// against 16.9 T at 2.07 GHz (profiles/r03/mfma_rate_codes_both_sides.txt); in the scan 0.87 busy at 1.91 GHz.
// What it costs: twice as many MFMA issues per comparison, so the VALU work beside them is no longer free -- a step
// of 1024 comparisons (8 MFMAs) has room for ~12 VALU instructions where 4 MFMAs of 32x32 leave room for ~21
// (profiles/r03/valu_beside_mfma_by_shape.txt).  Hence 224-query tiles (14 blocks of 16 queries: the 32 expansion
// instructions of a superstep are spread over 14 blocks), buffer loads (no address arithmetic), no loop over a run
// of jobs (it costs the registers the fourteenth block needs; it was worth 0.8 %).
//
// Geometry.  One MFMA: 16 train rows x 16 queries x 128 bits.  Lane l = (c = l & 15, g = l >> 4): as INPUT lane
// (c, g) gives the 32 bits of word 2 g + m (m = 0, 1: the two MFMAs of k = 256) of train row c / query c of the
// tile -- one 8-byte load per lane and tile; as OUTPUT it holds column (query) c, rows 4 g + j, j = 0..3.
// A SUPERSTEP = 64 train rows = four tiles t: tile t's row 4 g + j is superstep row 32 (g >> 1) + 4 (g & 1) + 8 t + j,
// so that lane (c, g) holds, over the four tiles, exactly the sixteen rows ONE lane of the 32x32 result holds in one
// of its 32-row steps -- step 2 S + (g >> 1), half h = g & 1, element i = 4 t + j (mfma_scan.h): the GROUPS are
// the same, the tags in the accumulators' start values are the same, and so are `best`, `second` and what the
// consumers (refine_second, the group-minimum table; vsmatch_kernels.h) assume about them.  The four lanes of a
// query are merged at the end as the two halves are there.
#pragma once
#include <type_traits>

#include "mfma_scan.h"

namespace vsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16_step(const i32x8 a, const i32x8 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
}

// One tile = 16 * NBQ queries of a job (NBQ even) against the split's share of its train range (rows [t0, t1); groups,
// steps and tags count from t0, as in mfma_scan.h).
template <int NBQ>
__device__ __forceinline__ void mfma16_scan_tile(const char* __restrict__ qbytes, const char* __restrict__ tbytes, const Job job,
                                                 u32 q0, u32 lane, u32 split, u32 splits, u32 split_stride,
                                                 uint2* __restrict__ scratch, u32 d_lim,
                                                 const AcceptMask& acc, const uint16_t* lut) {
  static_assert(NBQ >= 2 && NBQ % 2 == 0, "pairs of blocks share a word of the accept mask");
  const u32 c = lane & 15, g = lane >> 4, h = g & 1;
  u32 m88 = 0x88888888u, c11 = 0x11111111u, m44 = 0x44444444u, m22 = 0x22222222u, m11 = 0x11111111u;
  asm volatile("" : "+s"(m88), "+v"(c11), "+s"(m44), "+s"(m22), "+s"(m11));
  auto expand_train = [&](u32 w) { return expand_train_zr(w, m88, c11, m44, m22, m11); };

  // The split's share of the train range and its first rows, before anything else: the loads are in flight while the
  // queries are loaded and expanded.  Train rows by buffer loads: the lane's row / byte offset is constant, the tile's
  // offset the instruction's immediate, the superstep's rides in the VGPR offset (one v_add per superstep; the range
  // check is then on the whole offset).  Rows past the end of the range read as zero and never count (see the tail).
  const u32 per = (job.nt + splits - 1) / splits;
  const u32 t0 = min(split * per, job.nt), t1 = min(t0 + per, job.nt);
  const u32 nt = t1 - t0;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(tbytes + ((size_t)job.t_off + t0) * 32), 0, nt * 32u, 0x00020000);
  u32 voff = (32u * (c >> 3) + 4u * ((c >> 2) & 1u) + (c & 3u)) * 32u + g * 8u;
  auto load_super = [&](uint2 (&w)[4]) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const i32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 256u * t, 0, 0);
      w[t] = make_uint2((u32)v[0], (u32)v[1]);
    }
    voff += 2048u;
  };
  uint2 w[4], wn[4];
  const u32 sfull = nt / 64, left = nt - sfull * 64u;  // full supersteps; rows of the tail (below)
  // the tail's rows: loaded behind the first block of the last full superstep, into the registers of the
  // prefetch that superstep does not need
  const u32 voff_tail = (16u * (c >> 3) + 4u * ((c >> 2) & 1u) + (c & 3u)) * 32u + g * 8u + sfull * 2048u;
  auto load_tail = [&]() {
#pragma unroll
    for (int k = 0; k < 4; k++) {  // k = 2 * half-step + tile
      const i32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff_tail + 1024u * (k >> 1) + 256u * (k & 1), 0, 0);
      wn[k] = make_uint2((u32)v[0], (u32)v[1]);
    }
  };
  if (sfull) {
    load_super(w);
    load_super(wn);
  } else {
    load_tail();
  }

  i32x8 B[NBQ][2];
  u32 ppack[(NBQ + 3) / 4] = {};  // P of mfma_scan.h (CODES 1) per block, see zq
  auto zq = [&](int b) { return ZR_BASE - (int)(((ppack[b >> 2] >> (8 * (b & 3))) & 255u) << 14); };
  {
    // query codes (expand_query_zr) with the masks in registers: a shift and one v_and_or_b32 per register
    u32 q22 = 0x22222222u, q44 = 0x44444444u;
    asm volatile("" : "+v"(q22), "+v"(q44));
    auto expand_query = [&](u32 w) {
      i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      v[0] = (int)((~w & m88) | q22);
      v[1] = (int)(((w << 1) & m88) | c11);
      v[2] = (int)(((w << 2) & m88) | q22);
      v[3] = (int)(((w << 3) & m88) | q44);
      return v;
    };
#pragma unroll
    for (int b = 0; b < NBQ; b++) {
      const u32 qi = min(q0 + b * 16 + c, job.nq - 1);
      const uint2 w = *(const uint2*)(qbytes + (size_t)(job.q_off + qi) * 32 + g * 8);
      B[b][0] = expand_query(w.x);
      B[b][1] = expand_query(w.y);
      // this lane's share of P (at most 48 of 192): the four lanes of a query are added up below, four blocks at once
      ppack[b >> 2] |= (u32)(__popc(w.x & 0x77777777u) + __popc(w.y & 0x77777777u)) << (8 * (b & 3));
    }
#pragma unroll
    for (int k = 0; k < (NBQ + 3) / 4; k++) {
      ppack[k] += (u32)__shfl_xor((int)ppack[k], 16);
      ppack[k] += (u32)__shfl_xor((int)ppack[k], 32);
    }
  }
  const int dl = (int)(min(d_lim, 256u) << 14);  // start value of (best, second) above zq
  int best[NBQ], second[NBQ];
#pragma unroll
  for (int b = 0; b < NBQ; b++) best[b] = second[b] = zq(b) + dl;

  if (nt) {
    auto expand_super = [&](const uint2 (&w)[4], i32x8 (&X)[4][2]) {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        X[t][0] = expand_train(w[t].x);
        X[t][1] = expand_train(w[t].y);
      }
    };
    f32x4 zero[4];  // start values of the accumulators: 768.0 + i ulp, i = 4 t + j
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int j = 0; j < 4; j++) zero[t][j] = __int_as_float(ZR_START + 4 * t + j);
      asm volatile("" : "+v"(zero[t]));
    }
    // the eight MFMAs of (superstep operands X, query block b): the k halves of a tile BACK TO BACK -- the second
    // takes the first's result as its accumulator straight from the pipe: +2.3 to +3.5 % over the two halves two
    // MFMAs apart (as first written, for the latency of the dependence), +2 % over four apart, same box, 1333 ... 2500
    // descriptors per image (profiles/r03/scan16_mfma_order.txt)
    auto block_mfmas = [&](const i32x8 (&X)[4][2], int b, const f32x4 (&Z)[4], f32x4 (&R)[4]) {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        R[t] = mfma16_step(X[t][0], B[b][0], Z[t]);
        R[t] = mfma16_step(X[t][1], B[b][1], R[t]);
      }
    };
    auto med3 = [](int p, int q, int r) {
      return __float_as_int(__builtin_amdgcn_fmed3f(__int_as_float(p), __int_as_float(q), __int_as_float(r)));
    };
    auto group_min = [](const int (&x)[16]) {
      const int m0 = min3i(x[0], x[1], x[2]), m1 = min3i(x[3], x[4], x[5]), m2 = min3i(x[6], x[7], x[8]);
      const int m3 = min3i(x[9], x[10], x[11]), m4 = min3i(x[12], x[13], x[14]);
      return min(min3i(m0, m1, m2), min3i(m3, m4, x[15]));
    };
    i32x8 A[4][2], An[4][2];
    int step16 = (int)(16u * (g >> 1));  // 16 x the 32-row step this lane's group belongs to: 2 S + (g >> 1)
    if (sfull) {
      expand_super(w, A);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        An[t][0] = A[t][0];
        An[t][1] = A[t][1];
      }
      f32x4 D[4];
      block_mfmas(A, 0, zero, D);
      // One superstep, software-pipelined over its blocks: the MFMAs of the next block (of the next superstep's block
      // 0: LAST = false) are issued before the minimum tree of the current one.
      auto run_super = [&](auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
        for (int b = 0; b < NBQ; b++) {
          f32x4 Dn[4];
          if (b + 1 < NBQ) block_mfmas(A, b + 1, zero, Dn);
          else if (!LAST) block_mfmas(An, 0, zero, Dn);
          if (!LAST && b == NBQ - 2) {  // the next superstep's operands one block early; the one after goes in flight
            expand_super(wn, An);
            load_super(wn);
          }
          if (LAST && b == 0 && left) load_tail();
          if (b + 1 < NBQ || !LAST) {
#pragma unroll
            for (int t = 0; t < 4; t++) asm volatile("" : "+v"(Dn[t]));  // keeps the MFMAs above the minimum tree
          }
          int x[16];
#pragma unroll
          for (int i = 0; i < 16; i++) x[i] = __float_as_int(D[i >> 2][i & 3]);
          const int m = group_min(x);
          second[b] = med3(best[b], m, second[b]);
          best[b] = min(best[b], m | step16);
#pragma unroll
          for (int n = 0; n < 8; n++) {  // (one MFMA, two others; with the last three MFMAs one apart instead, or the
            // first three, or every other one: the same within the run-to-run spread; the pairs of a tile ADJACENT,
            // two MFMAs then four others: the gain of the back-to-back order is gone)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          if (b + 1 < NBQ || !LAST) {
#pragma unroll
            for (int t = 0; t < 4; t++) D[t] = Dn[t];
          }
        }
        if (!LAST) {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            A[t][0] = An[t][0];
            A[t][1] = An[t][1];
          }
          step16 += 32;
        }
      };
      for (u32 S = 0; S + 1 < sfull; S++) run_super(std::false_type{});
      run_super(std::true_type{});
    }
    // The tail: the `left` < 64 rows behind the last full superstep, in HALF-STEPS of 32 rows, two tiles each: tile
    // tau's row 4 g + j is row 8 (2 (g >> 1) + tau) + 4 (g & 1) + j of the half-step, so that a lane's eight rows are
    // half of one group of the 32x32 geometry (its 32-row step 2 sfull + half-step, half g & 1, elements
    // i = 4 (2 (g >> 1) + tau) + j): the tracking bounds `second` by the minima of SUBSETS of the groups the consumers
    // look through again -- at least as tight.  Rows past the end: their accumulators start from a value that is
    // never below a start value of best (their operands are zero: the result is that value).
    if (left) {
      const u32 halves = (left + 31) / 32;
      for (u32 hs = 0; hs < halves; hs++) {
        const uint2 w0 = hs ? wn[2] : wn[0], w1 = hs ? wn[3] : wn[1];
        i32x8 T[2][2];
        T[0][0] = expand_train(w0.x);
        T[0][1] = expand_train(w0.y);
        T[1][0] = expand_train(w1.x);
        T[1][1] = expand_train(w1.y);
        f32x4 zs[2];
#pragma unroll
        for (int tau = 0; tau < 2; tau++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const u32 r = hs * 32u + 8u * (2u * (g >> 1) + tau) + 4u * h + j;
            zs[tau][j] = r < left ? __int_as_float(ZR_START + (int)(4u * (2u * (g >> 1) + tau)) + j) : __int_as_float(NEVER);
          }
        const int st16 = (int)(16u * (2u * sfull + hs));
        auto half_mfmas = [&](int b, f32x4 (&R)[2]) {
          R[0] = mfma16_step(T[0][0], B[b][0], zs[0]);
          R[0] = mfma16_step(T[0][1], B[b][1], R[0]);
          R[1] = mfma16_step(T[1][0], B[b][0], zs[1]);
          R[1] = mfma16_step(T[1][1], B[b][1], R[1]);
        };
        f32x4 E[2];
        half_mfmas(0, E);
#pragma unroll
        for (int b = 0; b < NBQ; b++) {
          f32x4 En[2];
          if (b + 1 < NBQ) {
            half_mfmas(b + 1, En);
            asm volatile("" : "+v"(En[0]), "+v"(En[1]));
          }
          int x[8];
#pragma unroll
          for (int i = 0; i < 8; i++) x[i] = __float_as_int(E[i >> 2][i & 3]);
          const int m = min(min3i(min3i(x[0], x[1], x[2]), min3i(x[3], x[4], x[5]), x[6]), x[7]);
          second[b] = med3(best[b], m, second[b]);
          best[b] = min(best[b], m | st16);
          if (b + 1 < NBQ) {
            E[0] = En[0];
            E[1] = En[1];
          }
        }
      }
    }
  }

  // Epilogue (as mfma_scan.h): bits -> packed keys, the four lanes of a query (same queries, disjoint rows) merged
  // exactly; q0 and the start values are laundered so that nothing of this is carried through the scan.
  u32 q0e = q0;
  asm volatile("" : "+s"(q0e));
#pragma unroll
  for (int k = 0; k < (NBQ + 3) / 4; k++) asm volatile("" : "+v"(ppack[k]));
  uint2* out = scratch + (size_t)split * split_stride + job.out_off;
  const u32 sentinel = d_lim >= 256u ? KEY_INIT : ((d_lim << IDX_BITS) | t0);
  u32 mask_word = 0;
#pragma unroll
  for (int b = 0; b < NBQ; b++) {
    const u32 bb = (u32)best[b], sb = (u32)second[b];
    const int zb = zq(b), init = zb + dl;
    const u32 i = bb & 15u;
    const u32 row = t0 + 32u * ((bb >> 4) & 1023u) + 4u * h + (i & 3u) + 8u * (i >> 2);
    u32 kb = (int)bb < init ? (((((bb - (u32)zb) >> 14) & 255u) << IDX_BITS) | row) : sentinel;
    u32 ks = (int)(sb & ~16383u) < init ? ((((sb - (u32)zb) >> 14) & 255u) << IDX_BITS) : sentinel;
#pragma unroll
    for (int sh = 16; sh <= 32; sh *= 2) {
      const u32 ko = (u32)__shfl_xor((int)kb, sh), so = (u32)__shfl_xor((int)ks, sh);
      ks = min(min(ks, so), max(kb, ko));
      kb = min(kb, ko);
    }
    const u32 qi = q0e + b * 16 + c;
    const bool valid = g == 0 && qi < job.nq;
    if (valid) out[qi] = make_uint2(kb, ks);
    if (acc.mask) {  // keypoints.h:295-296 on the finished result, one bit per query (store_accept_mask, mfma_scan.h);
      // the accept table comes from LDS (see the kernel): a global load here cannot move above the stores of the block
      // before it, the table lookups of a tile's fourteen blocks would wait for memory one after the other
      const u32 d = kb >> IDX_BITS, sd = min(ks >> IDX_BITS, 256u);
      const bool hit = valid && d < 256u && (int)d < acc.threshold && sd >= (u32)lut[min(d, 256u)];
      const u32 m = (u32)__ballot(hit) & 0xffffu;
      if (b & 1) {
        const u32 q_first = q0e + (b - 1) * 16;
        if (lane == 0 && q_first < job.nq) acc.mask[(job.out_off >> 6) + (q_first >> 5)] = mask_word | (m << 16);
      } else {
        mask_word = m;
      }
    }
  }
}

constexpr u32 MFMA16_NBQ = 14;  // most blocks of 16 queries per tile: 224 queries
// Tiles of a job of nq queries: the fewest tiles of at most 14 blocks, BALANCED -- sizes are even (two blocks share a
// word of the accept mask) and differ by at most two blocks: 2000 queries = 9 tiles of 14 blocks, 2048 = 4 of 14 + 6
// of 12 (not 9 of 14 and one of 2: a short tile spreads the expansion of the train rows over few blocks).
struct Mfma16Tiling {
  u32 tiles, small, big_tiles;  // big_tiles tiles of small + 2 blocks first, then tiles of `small` blocks
};
__host__ __device__ inline Mfma16Tiling mfma16_tiling(u32 nq) {
  const u32 be = 2 * ((nq + 31) / 32);  // blocks, even
  Mfma16Tiling tl;
  tl.tiles = (be + MFMA16_NBQ - 1) / MFMA16_NBQ;
  tl.small = tl.tiles ? 2 * (be / (2 * tl.tiles)) : 0;
  tl.big_tiles = tl.tiles ? (be - tl.tiles * tl.small) / 2 : 0;
  return tl;
}
__host__ __device__ inline u32 mfma16_tiles(u32 nq) { return (2 * ((nq + 31) / 32) + MFMA16_NBQ - 1) / MFMA16_NBQ; }

// One-wave workgroups; the grid of mfma_scan_kernel<*, 1>: tiles_per_job != 0: eight jobs side by side (one per XCD),
// tile-major within them; else job-minor.  blockIdx.y = train split.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void mfma16_scan_kernel(
    const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase, const Job* __restrict__ jobs, u32 n_jobs,
    u32 splits, u32 split_stride, uint2* __restrict__ scratch, u32 tiles_per_job, u32 d_lim, AcceptMask acc) {
  const u32 lane = threadIdx.x & 63;
  __shared__ uint16_t lut[264];  // AcceptParams::min_second, for the epilogue (one-wave workgroup: no barrier needed beyond the wait)
  u32 j, t;
  if (tiles_per_job) {
    const u32 per_super = 8 * tiles_per_job;
    const u32 super = blockIdx.x / per_super, r = blockIdx.x - super * per_super;
    t = r >> 3;
    j = super * 8 + (r & 7);
  } else {
    t = blockIdx.x / n_jobs;
    j = blockIdx.x - t * n_jobs;
  }
  if (j >= n_jobs) return;
  const Job job = jobs[j];
  const Mfma16Tiling tl = mfma16_tiling(job.nq);
  if (t >= tl.tiles) return;
  if (acc.mask) {  // (the five loads of a lane issued together)
    uint16_t v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = acc.lut[min(lane + 64 * k, 256u)];
#pragma unroll
    for (int k = 0; k < 5; k++) lut[min(lane + 64 * k, 256u)] = v[k];
    __syncthreads();
  }
  const u32 blocks = t < tl.big_tiles ? tl.small + 2 : tl.small;
  const u32 q0 = 16 * (t < tl.big_tiles ? t * (tl.small + 2) : tl.big_tiles * (tl.small + 2) + (t - tl.big_tiles) * tl.small);
  const char* q = (const char*)qbase;
  const char* tr = (const char*)tbase;
  switch (blocks) {
    case 14: mfma16_scan_tile<14>(q, tr, job, q0, lane, blockIdx.y, splits, split_stride, scratch, d_lim, acc, lut); break;
    case 12: mfma16_scan_tile<12>(q, tr, job, q0, lane, blockIdx.y, splits, split_stride, scratch, d_lim, acc, lut); break;
    case 10: mfma16_scan_tile<10>(q, tr, job, q0, lane, blockIdx.y, splits, split_stride, scratch, d_lim, acc, lut); break;
    case 8: mfma16_scan_tile<8>(q, tr, job, q0, lane, blockIdx.y, splits, split_stride, scratch, d_lim, acc, lut); break;
    case 6: mfma16_scan_tile<6>(q, tr, job, q0, lane, blockIdx.y, splits, split_stride, scratch, d_lim, acc, lut); break;
    case 4: mfma16_scan_tile<4>(q, tr, job, q0, lane, blockIdx.y, splits, split_stride, scratch, d_lim, acc, lut); break;
    default: mfma16_scan_tile<2>(q, tr, job, q0, lane, blockIdx.y, splits, split_stride, scratch, d_lim, acc, lut); break;
  }
}

}  // namespace vsm
