// tools/mfma16_scan.h -- EXPERIMENTS (round 3) behind csrc/mfma16_scan.h: the scan of csrc/mfma_scan.h on
// v_mfma_f32_16x16x128_f8f6f4, first form (v1: groups of its own geometry) and v3 (32-row units, 32-row groups).
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
#include "../visual-slam_amd/csrc/mfma16_scan.h"  // the product kernel (csrc): typedefs, mfma16_step

namespace vsm {



// One tile = 16 * NBQ queries of a job against its whole train range (unsplit).
template <int NBQ>
__device__ __forceinline__ void mfma16v1_scan_tile(const char* __restrict__ qbytes, const char* __restrict__ tbytes, const Job job,
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
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void mfma16v1_scan_kernel(
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
  mfma16v1_scan_tile<NBQ>((const char*)qbase, (const char*)tbase, job, q0, lane, scratch, d_lim);
}

}  // namespace vsm

// ---------------------------------------------------------------------------------------------------------------
// v3 (round 3): the same scan with less VALU work beside the MFMAs -- the 16x16 shape issues twice as many MFMAs per
// comparison, so a step of 1024 comparisons (8 MFMAs) has room for ~12 VALU instructions, not ~21.
//  * a UNIT is 32 train rows x 16 queries: two tiles, four MFMAs, eight results per lane; a GROUP is the 32 rows a lane
//    sees in a CHUNK of 128 rows (four units): the minimum tree of a unit takes the minimum of the units before it as
//    its ninth input, so best / second are updated once per chunk: 4 + 4 + 4 + 4 + 3 = 19 instructions per 2048
//    comparisons (9.5 per step) instead of 22 -- and the operands of a unit are 16 registers, not 32;
//  * train rows come in by buffer loads: the row / byte offset of a lane is constant, the tile offset is the
//    instruction's immediate -- no clamp (rows past the end of the range read as zero and never count: their
//    accumulators start from a value above every start value of best);
//  * rows of a chunk: r = 32 v + 8 g + 4 t + j -- unit v, lane group g, tile t, element j: lane (c, g) of the result
//    holds, of every unit, the rows [8 g, 8 g + 8): tag i = 8 v + 4 t + j in the start values of the accumulators
//    (32 registers), chunk number S in bits 5-13.  The last chunk runs only the units that hold rows.
// (v2, one commit earlier: units of 64 rows -- 64 operand registers; with the queries in AGPRs, which at two waves per
// SIMD share the 256 registers of a wave with the VGPRs, it spilled.)
namespace vsm {


template <int NBQ>
__device__ __forceinline__ void mfma16v3_scan_tile(const char* __restrict__ qbytes, const char* __restrict__ tbytes, const Job job,
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
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(tbytes + (size_t)job.t_off * 32), 0, nt * 32u, 0x00020000);
    // A-row c of tile t of a unit: unit row 8 (c >> 2) + 4 t + (c & 3); this lane's eight bytes of it
    u32 voff = (8u * (c >> 2) + (c & 3u)) * 32u + g * 8u;
    // the two tiles of the next unit to load (the unit's offset rides in the VGPR offset, one v_add per unit: the range
    // check is then on the whole offset)
    auto load_unit = [&](uint2 (&w)[2]) {
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const i32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 128u * t, 0, 0);
        w[t] = make_uint2((u32)v[0], (u32)v[1]);
      }
      voff += 1024u;
    };
    auto expand_unit = [&](const uint2 (&w)[2], i32x8 (&X)[2][2]) {
#pragma unroll
      for (int t = 0; t < 2; t++) {
        X[t][0] = expand_train(w[t].x);
        X[t][1] = expand_train(w[t].y);
      }
    };
    f32x4 zero[4][2];  // start values: 768.0 + tag ulp
#pragma unroll
    for (int v = 0; v < 4; v++)
#pragma unroll
      for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int j = 0; j < 4; j++) zero[v][t][j] = __int_as_float(ZR_START + 8 * v + 4 * t + j);
        asm volatile("" : "+v"(zero[v][t]));
      }
    auto unit_mfmas = [&](const i32x8 (&X)[2][2], int b, const f32x4 (&Z)[2], f32x4 (&R)[2]) {
      R[0] = mfma16_step(X[0][0], B[b][0], Z[0]);
      R[1] = mfma16_step(X[1][0], B[b][0], Z[1]);
      R[0] = mfma16_step(X[0][1], B[b][1], R[0]);
      R[1] = mfma16_step(X[1][1], B[b][1], R[1]);
    };
    auto med3 = [](int p, int q, int r) {
      return __float_as_int(__builtin_amdgcn_fmed3f(__int_as_float(p), __int_as_float(q), __int_as_float(r)));
    };
    int carry[NBQ];
    // Software pipeline over units x blocks: the MFMAs of the next (unit, block) are issued before the tree of the
    // current one.  A chunk = units 0..3; `units` < 4 only in the last chunk of a range.
    uint2 wa[2], wb[2];
    i32x8 A0[2][2], A1[2][2];
    load_unit(wa);
    load_unit(wb);
    expand_unit(wa, A0);
    load_unit(wa);
    f32x4 D[2];
    u32 S = 0;
    // one unit v of a chunk (operands X; the next unit's go to Xn from w, and the unit after that into w)
    // NEXT: 0 = nothing follows (the scan's last unit), 1 = unit v + 1 of this chunk follows, 2 = unit 0 of the next chunk
    auto run_unit = [&](auto vtag, auto nexttag, const f32x4 (&Z)[4][2], const i32x8 (&X)[2][2], i32x8 (&Xn)[2][2], uint2 (&w)[2]) {
      constexpr int v = decltype(vtag)::value, NEXT = decltype(nexttag)::value;
#pragma unroll
      for (int b = 0; b < NBQ; b++) {
        f32x4 Dn[2];
        if (b + 1 < NBQ) unit_mfmas(X, b + 1, Z[v], Dn);
        else if (NEXT == 1) unit_mfmas(Xn, 0, Z[(v + 1) & 3], Dn);
        else if (NEXT == 2) unit_mfmas(Xn, 0, zero[0], Dn);
        if (NEXT && b == (NBQ >= 2 ? NBQ - 2 : 0)) {  // the next unit's operands one block early
          expand_unit(w, Xn);
          load_unit(w);
        }
        if (b + 1 < NBQ || NEXT) {
#pragma unroll
          for (int t = 0; t < 2; t++) asm volatile("" : "+v"(Dn[t]));
        }
        int x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = __float_as_int(D[i >> 2][i & 3]);
        const int m0 = min3i(x[0], x[1], x[2]), m1 = min3i(x[3], x[4], x[5]);
        if (v == 0) {
          carry[b] = min(min3i(m0, m1, x[6]), x[7]);
        } else {
          const int m = min3i(min3i(m0, m1, x[6]), x[7], carry[b]);
          if (v < 3) {
            carry[b] = m;
          } else {
            second[b] = med3(best[b], m, second[b]);
            best[b] = min(best[b], m | (int)(S << 5));
          }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        if (b + 1 < NBQ || NEXT) {
          D[0] = Dn[0];
          D[1] = Dn[1];
        }
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    unit_mfmas(A0, 0, zero[0], D);
    const u32 chunks = (nt + 127) / 128;
    for (; S + 1 < chunks; S++) {
      run_unit(I0{}, I1{}, zero, A0, A1, wb);
      run_unit(I1{}, I1{}, zero, A1, A0, wa);
      run_unit(I2{}, I1{}, zero, A0, A1, wb);
      run_unit(I3{}, I2{}, zero, A1, A0, wa);
    }
    // the last chunk: `units` units hold rows; rows past the end of the range start high (NEVER: above every start
    // value of best), units past the end are left out -- the update of best / second then comes behind the last unit run
    const u32 left = nt - S * 128u, units = (left + 31) / 32;
    if (left < 128u) {
#pragma unroll
      for (int v = 0; v < 4; v++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int j = 0; j < 4; j++) zero[v][t][j] = (32u * v + 8u * g + 4u * t + j) < left ? zero[v][t][j] : __int_as_float(NEVER);
      unit_mfmas(A0, 0, zero[0], D);  // block 0 of unit 0 was started from the full start values: again
    }
    auto finish = [&]() {  // fewer than four units: the carry is the chunk's group minimum
#pragma unroll
      for (int b = 0; b < NBQ; b++) {
        second[b] = med3(best[b], carry[b], second[b]);
        best[b] = min(best[b], carry[b] | (int)(S << 5));
      }
    };
    if (units == 4) {
      run_unit(I0{}, I1{}, zero, A0, A1, wb);
      run_unit(I1{}, I1{}, zero, A1, A0, wa);
      run_unit(I2{}, I1{}, zero, A0, A1, wb);
      run_unit(I3{}, I0{}, zero, A1, A0, wa);
    } else if (units == 3) {
      run_unit(I0{}, I1{}, zero, A0, A1, wb);
      run_unit(I1{}, I1{}, zero, A1, A0, wa);
      run_unit(I2{}, I0{}, zero, A0, A1, wb);
      finish();
    } else if (units == 2) {
      run_unit(I0{}, I1{}, zero, A0, A1, wb);
      run_unit(I1{}, I0{}, zero, A1, A0, wa);
      finish();
    } else {
      run_unit(I0{}, I0{}, zero, A0, A1, wb);
      finish();
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
    const u32 i = bb & 31u;
    const u32 row = 128u * ((bb >> 5) & 511u) + 32u * (i >> 3) + 8u * g + (i & 7u);
    u32 kb = (int)bb < init ? (((((bb - (u32)zb) >> 14) & 255u) << IDX_BITS) | row) : sentinel;
    u32 ks = (int)(sb & ~16383u) < init ? ((((sb - (u32)zb) >> 14) & 255u) << IDX_BITS) : sentinel;
#pragma unroll
    for (int sh = 16; sh <= 32; sh *= 2) {
      const u32 ko = (u32)__shfl_xor((int)kb, sh), so = (u32)__shfl_xor((int)ks, sh);
      ks = min(min(ks, so), max(kb, ko));
      kb = min(kb, ko);
    }
    const u32 qi = q0e + b * 16 + c;
    if (g == 0 && qi < job.nq) out[qi] = make_uint2(kb, ks);
  }
}

template <int NBQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void mfma16v3_scan_kernel(
    const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase, const Job* __restrict__ jobs, u32 n_jobs,
    uint2* __restrict__ scratch, u32 tiles_per_job, u32 d_lim) {
  const u32 lane = threadIdx.x & 63;
  const u32 per_super = 8 * tiles_per_job;
  const u32 super = blockIdx.x / per_super, r = blockIdx.x - super * per_super;
  const u32 t = r >> 3, j = super * 8 + (r & 7);
  if (j >= n_jobs) return;
  const Job job = jobs[j];
  const u32 q0 = t * 16 * NBQ;
  if (q0 >= job.nq) return;
  mfma16v3_scan_tile<NBQ>((const char*)qbase, (const char*)tbase, job, q0, lane, scratch, d_lim);
}

// refine_second for the v3 geometry: the best's group = the rows r of its chunk of 128 with the same (r >> 3) & 3
__device__ __forceinline__ u32 refine_second_v3(const u32x4* __restrict__ q, const u32x4* __restrict__ tr, u32 nt, uint2 v) {
  const u32 d = v.x >> IDX_BITS, idx = v.x & IDX_MASK;
  u32 s = min(v.y >> IDX_BITS, 256u);
  if (d >= 256u || s <= d || idx >= nt) return s;
  const u32 b0 = (idx & ~127u) + (idx & 24u);
  const u32x4 qa = q[0], qb = q[1];
  for (u32 k = 0; k < 32; k++) {
    const u32 r = b0 + 32u * (k >> 3) + (k & 7u);
    if (r >= nt || r == idx) continue;
    const u32x4 ta = tr[2 * (size_t)r], tb = tr[2 * (size_t)r + 1];
    const u32 dr = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                   __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
    s = min(s, dr);
  }
  return s;
}

}  // namespace vsm
