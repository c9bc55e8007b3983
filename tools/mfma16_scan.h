// tools/mfma16_scan.h -- EXPERIMENT (round 3): the scan of csrc/mfma_scan.h on v_mfma_f32_16x16x128_f8f6f4.
//
// Why: the bare instruction on random codes sustains 17.0-18.3 T comparisons/s on the whole chip at 2.15-2.35 GHz where
// the 32x32x64 shape holds 14.4-16.7 T at 1.83-2.14 GHz (tools/mfma_rate2.hip, tools/valu_beside_mfma.hip: the chip is
// power limited under FP4 MFMAs and the 16x16 shape accumulates into half as many f32 sums per multiply).  What it costs:
// twice as many MFMA issues per comparison, so the VALU work beside them is no longer free (a step of 1024 comparisons
// takes max(~138, 88 + 4 K) cycles of a SIMD with K VALU instructions: 32x32 max(~138, 44 + 4 K)).
//
// Geometry.  One MFMA: 16 train rows x 16 queries x 128 bits; lane l = (c = l & 15, g = l >> 4): as INPUT lane (c, g)
// gives the 32 bits of word 2 g + m (m = 0, 1: the two MFMAs of k = 256) of train row c / query c of the tile, one
// 8-byte load per lane and tile; as OUTPUT it holds column (query) c, rows 4 g + j, j = 0..3.  A SUPERSTEP = 64 train
// rows = four tiles: lane (c, g) then holds 16 results of ONE query -- its GROUP: rows 16 t + 4 g + j -- and the
// tracking of csrc/mfma_scan.h applies unchanged (tags i = 4 t + j in the accumulators' start values, group minimum by
// eight v_min3, med3 / or / min).  The four lanes of a query are merged at the end.
// Codes: the shift-free zero-rich ones of csrc/mfma_scan.h (CODES 1).
#pragma once
#include "mfma_scan.h"

namespace vsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16_step(const i32x8 a, const i32x8 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
}

// One tile = 16 * NBQ queries of a job against its whole train range (unsplit).
template <int NBQ>
__device__ __forceinline__ void mfma16_scan_tile(const char* __restrict__ qbytes, const char* __restrict__ tbytes, const Job job,
                                                 u32 q0, u32 lane, uint2* __restrict__ scratch, u32 d_lim) {
  const u32 c = lane & 15, g = lane >> 4;
  u32 m88 = 0x88888888u, c11 = 0x11111111u, m44 = 0x44444444u, m22 = 0x22222222u, m11 = 0x11111111u;
  asm volatile("" : "+s"(m88), "+v"(c11), "+s"(m44), "+s"(m22), "+s"(m11));
  auto expand_train = [&](u32 w) { return expand_train_zr(w, m88, c11, m44, m22, m11); };

  i32x8 B[NBQ][2];
  u32 ppack[(NBQ + 3) / 4] = {};
  auto zq = [&](int b) { return ZR_BASE - (int)(((ppack[b >> 2] >> (8 * (b & 3))) & 255u) << 14); };
#pragma unroll
  for (int b = 0; b < NBQ; b++) {
    const u32 qi = min(q0 + b * 16 + c, job.nq - 1);
    const uint2 w = *(const uint2*)(qbytes + (size_t)(job.q_off + qi) * 32 + g * 8);
    B[b][0] = expand_query_zr(w.x);
    B[b][1] = expand_query_zr(w.y);
    int ph = __popc(w.x & 0x77777777u) + __popc(w.y & 0x77777777u);
    ph += __shfl_xor(ph, 16);
    ph += __shfl_xor(ph, 32);
    ppack[b >> 2] |= (u32)ph << (8 * (b & 3));
  }
  const int dl = (int)(min(d_lim, 256u) << 14);
  int best[NBQ], second[NBQ];
#pragma unroll
  for (int b = 0; b < NBQ; b++) best[b] = second[b] = zq(b) + dl;

  const u32 nt = job.nt;
  if (nt) {
    const char* tr = tbytes + (size_t)job.t_off * 32;
    const u32 last_off = (nt - 1) * 32u + g * 8u;
    auto row2 = [&](u32 off) { return *(const uint2*)(tr + min(off, last_off)); };
    u32 off = c * 32u + g * 8u;  // this lane's row of tile 0 of the current superstep
    f32x4 zero[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int j = 0; j < 4; j++) zero[t][j] = __int_as_float(ZR_START + 4 * t + j);
#pragma unroll
    for (int t = 0; t < 4; t++) asm volatile("" : "+v"(zero[t]));
    uint2 w[4], wn[4];
#pragma unroll
    for (int t = 0; t < 4; t++) w[t] = row2(off + t * 512u);
#pragma unroll
    for (int t = 0; t < 4; t++) wn[t] = row2(off + 2048u + t * 512u);
    off += 4096u;
    i32x8 A[4][2], An[4][2];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      A[t][0] = expand_train(w[t].x);
      A[t][1] = expand_train(w[t].y);
      An[t][0] = A[t][0];
      An[t][1] = A[t][1];
    }
    // the eight MFMAs of (superstep operands X, query block b): tiles in pairs, the k halves of a pair interleaved so
    // that no MFMA is issued directly behind the one it depends on
    auto block_mfmas = [&](const i32x8 (&X)[4][2], int b, f32x4 (&R)[4]) {
      R[0] = mfma16_step(X[0][0], B[b][0], zero[0]);
      R[1] = mfma16_step(X[1][0], B[b][0], zero[1]);
      R[0] = mfma16_step(X[0][1], B[b][1], R[0]);
      R[1] = mfma16_step(X[1][1], B[b][1], R[1]);
      R[2] = mfma16_step(X[2][0], B[b][0], zero[2]);
      R[3] = mfma16_step(X[3][0], B[b][0], zero[3]);
      R[2] = mfma16_step(X[2][1], B[b][1], R[2]);
      R[3] = mfma16_step(X[3][1], B[b][1], R[3]);
    };
    auto med3 = [](int p, int q, int r) {
      return __float_as_int(__builtin_amdgcn_fmed3f(__int_as_float(p), __int_as_float(q), __int_as_float(r)));
    };
    auto group_min = [](const int (&x)[16]) {
      const int m0 = min3i(x[0], x[1], x[2]), m1 = min3i(x[3], x[4], x[5]), m2 = min3i(x[6], x[7], x[8]);
      const int m3 = min3i(x[9], x[10], x[11]), m4 = min3i(x[12], x[13], x[14]);
      return min(min3i(m0, m1, m2), min3i(m3, m4, x[15]));
    };
    f32x4 D[4];
    block_mfmas(A, 0, D);
    u32 step16 = 0, tr0 = 0;
    for (; tr0 + 64 <= nt; tr0 += 64, step16 += 16) {
#pragma unroll
      for (int b = 0; b < NBQ; b++) {
        f32x4 Dn[4];
        if (b + 1 < NBQ) block_mfmas(A, b + 1, Dn);
        else block_mfmas(An, 0, Dn);
        if (b == (NBQ >= 2 ? NBQ - 2 : 0)) {  // the next superstep's operands, one block early; the one after goes in flight
#pragma unroll
          for (int t = 0; t < 4; t++) {
            An[t][0] = expand_train(wn[t].x);
            An[t][1] = expand_train(wn[t].y);
            wn[t] = row2(off + t * 512u);
          }
          off += 2048u;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) asm volatile("" : "+v"(Dn[t]));
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __float_as_int(D[i >> 2][i & 3]);
        const int m = group_min(x);
        second[b] = med3(best[b], m, second[b]);
        best[b] = min(best[b], m | (int)step16);
#pragma unroll
        for (int n = 0; n < 8; n++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) D[t] = Dn[t];
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        A[t][0] = An[t][0];
        A[t][1] = An[t][1];
      }
    }
    if (tr0 < nt) {  // the last, partial superstep: D holds its block 0, A its operands
      const u32 left = nt - tr0;
#pragma unroll
      for (int b = 0; b < NBQ; b++) {
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const u32 ro = 16u * (u32)(i >> 2) + 4u * g + (u32)(i & 3);
          x[i] = ro < left ? __float_as_int(D[i >> 2][i & 3]) : NEVER;
        }
        const int m = group_min(x);
        second[b] = med3(best[b], m, second[b]);
        best[b] = min(best[b], m | (int)step16);
        if (b + 1 < NBQ) block_mfmas(A, b + 1, D);
      }
    }
  }

  u32 q0e = q0;
  asm volatile("" : "+s"(q0e));
#pragma unroll
  for (int k = 0; k < (NBQ + 3) / 4; k++) asm volatile("" : "+v"(ppack[k]));
  uint2* out = scratch + job.out_off;
  const u32 sentinel = d_lim >= 256u ? KEY_INIT : (d_lim << IDX_BITS);
#pragma unroll
  for (int b = 0; b < NBQ; b++) {
    const u32 bb = (u32)best[b], sb = (u32)second[b];
    const int zb = zq(b), init = zb + dl;
    const u32 i = bb & 15u;
    const u32 row = 64u * ((bb >> 4) & 1023u) + 16u * (i >> 2) + 4u * g + (i & 3u);
    u32 kb = (int)bb < init ? (((((bb - (u32)zb) >> 14) & 255u) << IDX_BITS) | row) : sentinel;
    u32 ks = (int)(sb & ~16383u) < init ? ((((sb - (u32)zb) >> 14) & 255u) << IDX_BITS) : sentinel;
#pragma unroll
    for (int sh = 16; sh <= 32; sh *= 2) {  // the four lanes of a query: same queries, disjoint rows
      const u32 ko = (u32)__shfl_xor((int)kb, sh), so = (u32)__shfl_xor((int)ks, sh);
      ks = min(min(ks, so), max(kb, ko));
      kb = min(kb, ko);
    }
    const u32 qi = q0e + b * 16 + c;
    if (g == 0 && qi < job.nq) out[qi] = make_uint2(kb, ks);
  }
}

template <int NBQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void mfma16_scan_kernel(
    const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase, const Job* __restrict__ jobs, u32 n_jobs,
    uint2* __restrict__ scratch, u32 tiles_per_job, u32 d_lim) {
  const u32 lane = threadIdx.x & 63;
  const u32 per_super = 8 * tiles_per_job;  // eight jobs side by side: one per XCD
  const u32 super = blockIdx.x / per_super, r = blockIdx.x - super * per_super;
  const u32 t = r >> 3, j = super * 8 + (r & 7);
  if (j >= n_jobs) return;
  const Job job = jobs[j];
  const u32 q0 = t * 16 * NBQ;
  if (q0 >= job.nq) return;
  mfma16_scan_tile<NBQ>((const char*)qbase, (const char*)tbase, job, q0, lane, scratch, d_lim);
}

}  // namespace vsm
