// vsmatch_kernels.h -- gfx950 (MI355X, wave64) kernels of the descriptor matcher.
//
// Path: visnav::matchDescriptors, /root/reference/include/visnav/keypoints.h:247-361.
//
//   scan_kernel      one direction of the brute-force scan (keypoints.h:256-290 and
//                    305-339): per query the packed (distance, lowest index) of the
//                    nearest train descriptor and the second smallest distance.
//   count_kernel /   accept test (keypoints.h:295-296, 344-345) on both directions,
//   fill_kernel      cross-check (354-360) and ordered compaction of the match list.
//   finalize_nn2     unpacks the scan result for the raw 2-NN entry point.
//
// This file: the popcount form of the scan, integer VALU work only (v_xor_b32 +
// v_bcnt_u32_b32 with its free accumulate) -- the formulation BASELINE.json's north_star
// describes, 0.92 of the integer-VALU roofline, kept as the reference point bench.py reports
// (`popcount_path`) and reachable through vsm_set_variant -- and the consumer kernels.  The
// shipped scan is the matrix-core form in mfma_scan.h (same jobs, same scratch layout, same
// results, 5x faster: a Hamming distance is also a +-1 dot product).  The earlier popcount
// variants (no filler, s_nop filler, float keys without the skip test, 512-query tiles,
// LDS-staged train tiles) live in tools/history/scan_variants_kernels.h with their harness.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vsm {

using u32 = uint32_t;
using u64 = uint64_t;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x8 __attribute__((ext_vector_type(8)));

// Packed key = (distance << 23) | train index.  Distances are 0..256 (9 bits),
// so the key is a u32 whose unsigned order is (distance, index) lexicographic:
// v_min_u32 on keys = "smaller distance, ties to the lower index", which is what
// the reference's strict '<' in an ascending scan gives (keypoints.h:273).
constexpr u32 IDX_BITS = 23;
constexpr u32 IDX_MASK = (1u << IDX_BITS) - 1u;
// Start value of best and second: distance 256, index 0.  A real key with
// distance 256 is >= this, so "256 is never recorded" (keypoints.h:261-262, 273).
constexpr u32 KEY_INIT = 256u << IDX_BITS;

// One scan job: queries [q_off, q_off+nq) of qbase against train [t_off, t_off+nt)
// of tbase; results go to scratch[out_off + query].  32 bytes so that a wave
// fetches it with one s_load_dwordx8.
// qlist_off != NO_QLIST: the job's k-th query is descriptor q_off + qlist[qlist_off + k]
// and its result goes to scratch[out_off + qlist[qlist_off + k]] (pruned backward pass).
constexpr u32 NO_QLIST = 0xffffffffu;
struct __attribute__((aligned(32))) Job {
  u32 q_off, nq, t_off, nt, out_off, qlist_off, pad1, pad2;
};

// Host-evaluated accept test.  min_second[b] = smallest s in [0, 256] with
// dist_2_best * b <= s evaluated in double exactly as keypoints.h:296 does
// (257 = no such s); the test is monotone in s, so `second >= min_second[best]`
// is the same predicate with no floating point on the device.
struct AcceptParams {
  int threshold;
  uint16_t min_second[258];
};

__device__ __forceinline__ u32 med3_u32(u32 a, u32 b, u32 c) {
  u32 r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// popcount(x) + acc in one instruction (the accumulate operand of v_bcnt is free).
// Written as asm because the compiler otherwise re-associates the eight counts
// into v_bcnt(x, 0) + v_add3_u32 trees, three instructions more per comparison.
// A v_mov is glued behind each v_bcnt (tools/valu_rate2: on gfx950 a v_bcnt occupies the SIMD
// for ~4.6 cycles but a following v_mov / s_nop / f32 add, mul or fma of the same wave issues in
// its shadow, and the v_xor after that then runs at the 2.3-cycle rate instead of 4.6).
__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc) {
  u32 r, junk;
  asm("v_bcnt_u32_b32 %0, %2, %3\n\tv_mov_b32 %1, %2" : "=v"(r), "=&v"(junk) : "v"(x), "v"(acc));
  return r;
}
__device__ __forceinline__ u32 bcnt0(u32 x) {
  u32 r, junk;
  asm("v_bcnt_u32_b32 %0, %2, 0\n\tv_mov_b32 %1, %2" : "=v"(r), "=&v"(junk) : "v"(x));
  return r;
}

// One train descriptor (wave-uniform, in SGPRs) against the R query descriptors each lane holds
// in VGPRs.  Invariant best <= second, so the new second is the median of (best, key, second).
//
// FILL == 2: integer keys, 8 xor + 8 bcnt (each with its v_mov filler) + v_lshl_or + v_med3_u32 +
// v_min_u32 per comparison.  Any train range.
//
// FILL == 8: float keys for train ranges of <= 32768 descriptors.  The popcount chain is seeded
// with 0x4B000000, the bit pattern of 2^23, so its result read as f32 is exactly 2^23 + d.  Then
//     df  = chain - 2^23                       (v_add_f32, exact)
//     key = df * 32768 + (2^23 + j)            (v_fma_f32, exact: < 2^24 for d <= 255)
// is an f32 whose order is (distance, index) lexicographic.  Both instructions are of the class
// that issues in the shadow of a v_bcnt, and they ride in the filler slots of the NEXT
// comparison's chain (behind its first two v_bcnt; v_mov behind the other six), so the key costs
// no issue slot.  v_med3_f32 / v_min_f32 run only if some lane's key is below its second best
// (v_cmp_lt_f32 + s_cbranch: exact, they would return their old values otherwise).  Distance 256
// gives key >= 2^24 = the start value: never recorded.  `Pend` carries the last chain of one call
// (its popcount result and the bit pattern of 2^23 + j) into the next call; scan_tile flushes it
// after the last train descriptor.
constexpr u32 FKEY_MAGIC = 0x4B000000u;   // 2^23 as f32
constexpr u32 FKEY_INIT = 0x4B800000u;    // 2^24 as f32 = distance 256, index 0
constexpr u32 FKEY_MAX_SET = 32768;       // also the most rows per split of the matrix-core scan (1024 steps)
// f32 key -> packed integer key; `base` = first train index of the scanned range
// (float keys index relative to it so that any range of <= 32768 descriptors works).
__device__ __forceinline__ u32 fkey_to_key(u32 fbits, u32 base) {
  const u32 k = (u32)(__uint_as_float(fbits) - 8388608.0f);  // d * 32768 + j, exact
  const u32 d = k >> 15;
  return d >= 256u ? KEY_INIT : ((d << IDX_BITS) | ((k & 32767u) + base));
}
struct Pend {
  u32 d, jf;
};
constexpr u32 PEND_INIT_D = FKEY_MAGIC + 256u;  // distance 256: a key >= FKEY_INIT, never recorded

template <int R, int FILL>
__device__ __forceinline__ void update(const u32 (&q)[R][8], const u32x8 t, u32 j, u32 (&best)[R],
                                       u32 (&second)[R], Pend& pend) {
  static_assert(FILL == 2 || FILL == 8, "popcount scan: integer keys (2) or float keys with the skip test (8)");
  if (FILL == 8) {
    const u32 magic = FKEY_MAGIC;
    const float scale = 32768.0f;
    const u32 jf = FKEY_MAGIC + j;
    u32 dprev = pend.d;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int pr = (r + R - 1) % R;             // the comparison whose tracking rides along
      const u32 jfp = r == 0 ? pend.jf : jf;      // ... and the train index it was made with
      u32 d, df, key, nb, ns;
      asm("v_bcnt_u32_b32 %0, %2, %3\n\tv_add_f32 %1, 0xcb000000, %4"
          : "=&v"(d), "=v"(df) : "v"(q[r][0] ^ t[0]), "v"(magic), "v"(dprev));
      asm("v_bcnt_u32_b32 %0, %2, %0\n\tv_fma_f32 %1, %3, %4, %5"
          : "+v"(d), "=v"(key) : "v"(q[r][1] ^ t[1]), "v"(df), "v"(scale), "s"(jfp));
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(__uint_as_float(key) < __uint_as_float(second[pr])) != 0, 0)) {
        asm("v_med3_f32 %0, %2, %3, %4\n\tv_min_f32 %1, %2, %3"
            : "=&v"(ns), "=&v"(nb) : "v"(best[pr]), "v"(key), "v"(second[pr]));
        best[pr] = nb;
        second[pr] = ns;
      }
#pragma unroll
      for (int w = 2; w < 8; w++) d = bcnt_acc(q[r][w] ^ t[w], d);
      dprev = d;
    }
    pend.d = dprev;
    pend.jf = jf;
    return;
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    u32 d = bcnt0(q[r][0] ^ t[0]);
#pragma unroll
    for (int w = 1; w < 8; w++) d = bcnt_acc(q[r][w] ^ t[w], d);
    const u32 key = (d << IDX_BITS) | j;
    second[r] = med3_u32(best[r], key, second[r]);
    best[r] = min(best[r], key);
  }
}

// Train descriptors are read through the constant address space so that the
// wave-uniform loads become s_load_dwordx8 (scalar data cache, read only).
typedef const __attribute__((address_space(4))) u32x8 cdesc_t;

// Grid: x = workgroup-of-job * n_jobs + job (job-minor, so that the workgroups the
// dispatcher deals round-robin to the 8 XCDs mix full and empty tiles of ragged
// jobs evenly: with job-major order and 8 workgroups per job, whole XCDs got
// nothing but empty tiles and the pruned backward pass ran at half speed),
// y = train split.  A wave is
// the unit of work: it owns 64*R consecutive queries of one job (lane l holds
// queries q0 + r*64 + l, so every query load is a coalesced 2 KiB) and scans
// its split of the train range.  Waves never communicate; there is no LDS and
// no barrier in this kernel.
// One tile = 64*R queries of a job (lane l holds queries q0 + r*64 + l) against the split's
// share of the job's train range.
template <int R, int FILL>
__device__ __forceinline__ void scan_tile(const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase,
                                          const Job& job, u32 q0, u32 lane, u32 split, u32 splits,
                                          u32 split_stride, uint2* __restrict__ scratch,
                                          const u32* __restrict__ qlist) {
  const u32 per = (job.nt + splits - 1) / splits;
  const u32 t0 = min(split * per, job.nt), t1 = min(t0 + per, job.nt);

  u32 q[R][8], col[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const u32 qi = min(q0 + r * 64 + lane, job.nq - 1);
    col[r] = job.qlist_off != NO_QLIST ? qlist[job.qlist_off + qi] : qi;
    const u32x4* p = qbase + 2 * (size_t)(job.q_off + col[r]);
    const u32x4 a = p[0], b = p[1];
    q[r][0] = a.x; q[r][1] = a.y; q[r][2] = a.z; q[r][3] = a.w;
    q[r][4] = b.x; q[r][5] = b.y; q[r][6] = b.z; q[r][7] = b.w;
  }
  u32 best[R], second[R];
#pragma unroll
  for (int r = 0; r < R; r++) best[r] = second[r] = (FILL == 8 ? FKEY_INIT : KEY_INIT);
  Pend pend = {PEND_INIT_D, FKEY_MAGIC};

  if (t1 > t0) {
    cdesc_t* tr = (cdesc_t*)(uintptr_t)tbase + job.t_off;
    const u32 last = t1 - 1;
    const u32 jb = FILL == 8 ? t0 : 0u;  // float keys count from the start of the range
    u32 j = t0;
    // Software-pipelined: two groups of two descriptors in SGPRs; the loads of
    // one group are in flight while the other is consumed.  Indices past the
    // range are clamped to `last` (loaded, never used).
    u32x8 a0 = tr[min(j, last)], a1 = tr[min(j + 1, last)];
    for (; j + 4 <= t1; j += 4) {
      const u32x8 b0 = tr[j + 2], b1 = tr[j + 3];
      update<R, FILL>(q, a0, j - jb, best, second, pend);
      update<R, FILL>(q, a1, j + 1 - jb, best, second, pend);
      a0 = tr[min(j + 4, last)];
      a1 = tr[min(j + 5, last)];
      update<R, FILL>(q, b0, j + 2 - jb, best, second, pend);
      update<R, FILL>(q, b1, j + 3 - jb, best, second, pend);
    }
    for (; j < t1; ++j) update<R, FILL>(q, tr[j], j - jb, best, second, pend);
  }

  if (FILL == 8) {  // the last chain's tracking is still pending
    const float key = fmaf(__uint_as_float(pend.d) - 8388608.0f, 32768.0f, __uint_as_float(pend.jf));
    const float b = __uint_as_float(best[R - 1]), sc = __uint_as_float(second[R - 1]);
    second[R - 1] = __float_as_uint(__builtin_amdgcn_fmed3f(b, key, sc));
    best[R - 1] = __float_as_uint(fminf(b, key));
  }
  uint2* out = scratch + (size_t)split * split_stride + job.out_off;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const u32 qi = q0 + r * 64 + lane;
    if (FILL == 8) {
      best[r] = fkey_to_key(best[r], t0);
      second[r] = fkey_to_key(second[r], t0);
    }
    if (qi < job.nq) out[col[r]] = make_uint2(best[r], second[r]);
  }
}

template <int R, int FILL>
__global__ __launch_bounds__(256) void scan_kernel(const u32x4* __restrict__ qbase,
                                                   const u32x4* __restrict__ tbase,
                                                   const Job* __restrict__ jobs, u32 n_jobs,
                                                   u32 splits, u32 split_stride,
                                                   uint2* __restrict__ scratch,
                                                   const u32* __restrict__ qlist) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 wg_of_job = blockIdx.x / n_jobs;
  const u32 job_id = blockIdx.x - wg_of_job * n_jobs;
  const Job job = jobs[job_id];
  const u32 q0 = (wg_of_job * 4 + wave) * 64 * R;
  if (q0 >= job.nq) return;
  scan_tile<R, FILL>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist);
}

// Ragged jobs (real images have 100..1400 corners, not a multiple of the tile): the job's queries
// are cut into full tiles of 64 * RMAX and, for the remainder, one tile per set bit of
// ceil(rest / 64) -- 128, 64 queries -- so that a wave never carries more than 63 idle query
// slots.  adaptive_tiles(nq, rmax) on the host gives the number of tiles a job of nq queries has.
__host__ __device__ inline u32 adaptive_tiles(u32 nq, u32 rmax) {  // rmax = 8, 4, 2 or 1
  const u32 n64 = (nq + 63) >> 6, rem = n64 & (rmax - 1);
  return n64 / rmax + ((rem >> 2) & 1) + ((rem >> 1) & 1) + (rem & 1);
}
// One-wave workgroups (the tiles of a ragged job run 1-4 x as long as each other, and a 4-wave
// workgroup can only be placed when a slot is free on every SIMD of a CU at once).  Two block orders:
//  * grouped (tiles_per_job > 0; used when every job has the same number of tiles):
//    blockIdx.x = (group of 8 jobs, tile, job in group).  Workgroups are dealt round-robin to
//    the 8 XCDs, so all tiles of a job land on the same XCD within 8 * tiles consecutive
//    dispatches and its L2 serves the train set to every tile after the first (FETCH_SIZE
//    3.4 -> 0.9 GB per 16384-pair launch against job-minor order).
//  * job-minor (tiles_per_job == 0; ragged jobs): blockIdx.x = tile * n_jobs + job.  The idle
//    workgroups of jobs with fewer tiles then all sit at the end of the grid; interleaved with
//    the live ones (grouped order) they cost a ragged batch 2.5x (measured, tools/ragged_probe).
// RMAX: queries per lane of the full tiles (4 -> 256-query tiles, 2 -> 128, 1 -> 64).
template <int FILL, int RMAX>
__global__ __launch_bounds__(64) void scan_kernel_adaptive(const u32x4* __restrict__ qbase,
                                                           const u32x4* __restrict__ tbase,
                                                           const Job* __restrict__ jobs, u32 n_jobs,
                                                           u32 splits, u32 split_stride,
                                                           uint2* __restrict__ scratch,
                                                           const u32* __restrict__ qlist,
                                                           u32 tiles_per_job) {
  static_assert(RMAX == 4 || RMAX == 2 || RMAX == 1, "256-, 128- or 64-query full tiles");
  const u32 lane = threadIdx.x & 63;
  u32 job_id, t;
  if (tiles_per_job) {
    const u32 per_group = 8 * tiles_per_job;
    const u32 group = blockIdx.x / per_group, r = blockIdx.x - group * per_group;
    t = r >> 3;
    job_id = group * 8 + (r & 7);
    if (job_id >= n_jobs) return;
  } else {
    t = blockIdx.x / n_jobs;
    job_id = blockIdx.x - t * n_jobs;
  }
  const Job job = jobs[job_id];
  const u32 n64 = (job.nq + 63) >> 6, full = n64 / RMAX, rem = n64 & (RMAX - 1);
  if (t < full) {
    scan_tile<RMAX, FILL>(qbase, tbase, job, t * 64 * RMAX, lane, blockIdx.y, splits, split_stride, scratch, qlist);
    return;
  }
  // remainder tiles in the order 128, 64 (those whose bit is set in rem)
  u32 k = t - full, q0 = full * 64 * RMAX;
  if (RMAX > 2 && (rem & 2)) {
    if (k == 0) {
      scan_tile<2, FILL>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist);
      return;
    }
    k--;
    q0 += 128;
  }
  if ((rem & 1) && k == 0)
    scan_tile<1, FILL>(qbase, tbase, job, q0, lane, blockIdx.y, splits, split_stride, scratch, qlist);
}

// Exact merge of two partial (best key, second key) results over disjoint
// train ranges: the two smallest of the union.
__device__ __forceinline__ uint2 merge2(uint2 a, uint2 b) {
  return make_uint2(min(a.x, b.x), min(min(a.y, b.y), max(a.x, b.x)));
}

__device__ __forceinline__ uint2 load_merged(const uint2* __restrict__ scratch, u32 off, u32 i,
                                             u32 splits, u32 split_stride) {
  uint2 v = scratch[off + i];
  for (u32 s = 1; s < splits; s++) v = merge2(v, scratch[(size_t)s * split_stride + off + i]);
  return v;
}

// Folds the partial results of train splits 1..splits-1 into split 0, in place, one
// thread per scratch entry (entries no scan wrote are garbage in, garbage out and
// never read).  After it the consumer kernels run with splits = 1; without it a
// single pair split 46 ways spent 175 us in count/fill's serial merge loops.
__global__ __launch_bounds__(256) void merge_splits_kernel(uint2* __restrict__ scratch, u32 n,
                                                           u32 splits, u32 split_stride) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint2 v = scratch[i];
#pragma unroll 8
  for (u32 s = 1; s < splits; s++) v = merge2(v, scratch[(size_t)s * split_stride + i]);
  scratch[i] = v;
}

// The matrix-core scan (mfma_scan.h) stores an UPPER BOUND of the second best distance: exact
// unless the true second best sits in the best's own GROUP -- the sixteen train rows one lane of
// the 32 x 32 result block holds in one step of the scan: rows b0 + 4 hh + (k & 3) + 8 (k >> 2),
// k = 0..15, with b0 = t0 + 32 * ((idx - t0) / 32), hh = ((idx - b0) >> 2) & 1 and t0 the start
// of the train split that holds idx.  This looks at the group's other fifteen rows again: the
// exact second best is the minimum of the stored one and of their distances (keypoints.h:273-289:
// best and second are the two smallest of the multiset; every row outside the group is covered
// by its own group's minimum, which the scan tracked).  `q` = the query's descriptor, `tr` = first
// descriptor of the job's train range, `per` = rows per train split (= nt when unsplit).  Called
// by the consumers only for results the accept test would otherwise pass (a few per image pair)
// and by the raw 2-NN output for every query.  Idempotent, and harmless on an exact result.
// Cost on a DENSE batch (real-like database: 700 listed rows per image pair): four 128-byte lines per
// result, 7.2 GB per 19900 pairs when every listed row is refined (1.05 ms of nominate_kernel's 1.3).
// The lines come from beyond L2: the forward scan of the whole launch (4 GB streamed) lies between a
// job's scan and its consumer kernels, so its train image has left the 4 MB L2s long before.  What
// helped: refining a forward result only after the cross-check (count_list_kernel), when a third of
// the listed rows are gone -- real-like batch 10.5 -> 9.7 ms (A/B on one box).  Tried without gain
// (tools/prof_real_like*.sh, round 2):
//  * sixteen lanes per result, four results in flight per group: the same 1.3 ms;
//  * eight lanes per result, each loading one 16-byte piece of a 128-byte chunk (8 lines per load
//    instruction instead of 64), four results per team in flight: nominate 1.4, count_list 2.1 ms --
//    the line count is the same, so the L1's lookup rate was not the limit;
//  * a second instantiation for jobs listing >= nt / 8 rows that copies the train image into LDS
//    first (64 KB per job instead of 350 KB from HBM): nominate 1.2, count_list 0.9 ms -- 20 bank
//    conflict cycles per ds_read_b128 (64 lanes, random 32-byte rows) and two workgroups per CU.
__device__ __forceinline__ u32 refine_second(const u32x4* __restrict__ q, const u32x4* __restrict__ tr, u32 nt,
                                             u32 per, uint2 v) {
  const u32 d = v.x >> IDX_BITS, idx = v.x & IDX_MASK;
  u32 s = min(v.y >> IDX_BITS, 256u);
  if (d >= 256u || s <= d || idx >= nt) return s;  // nothing found / already a tie
  const u32 t0 = idx / per * per, t1 = min(t0 + per, nt);
  const u32 b0 = t0 + ((idx - t0) & ~31u) + ((idx - t0) & 4u);
  const u32x4 qa = q[0], qb = q[1];
#pragma unroll
  for (int g = 0; g < 2; g++) {
    u32x4 ta[8], tb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {  // eight loads in flight; rows past the end are clamped, then ignored
      const u32 r = min(b0 + (u32)((k & 3) + 8 * (2 * g + (k >> 2))), t1 - 1);
      ta[k] = tr[2 * (size_t)r];
      tb[k] = tr[2 * (size_t)r + 1];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 r = b0 + (u32)((k & 3) + 8 * (2 * g + (k >> 2)));
      const u32 dr = __popc(qa.x ^ ta[k].x) + __popc(qa.y ^ ta[k].y) + __popc(qa.z ^ ta[k].z) + __popc(qa.w ^ ta[k].w) +
                     __popc(qb.x ^ tb[k].x) + __popc(qb.y ^ tb[k].y) + __popc(qb.z ^ tb[k].z) + __popc(qb.w ^ tb[k].w);
      if (r < t1 && r != idx) s = min(s, dr);
    }
  }
  return s;
}

// A certificate that refine_second cannot change an accept decision, so that the consumer kernels
// of a dense batch need not read the fifteen group mates at all: gmin[t] = the smallest distance
// from train row t to the other rows of ITS OWN group (rows of the same image; a fixed function of
// the row number when the scan is unsplit: t0 = 0).  For a query q whose best row is t at distance d,
// every group mate t' has dist(q, t') >= dist(t, t') - dist(q, t) >= gmin[t] - d (triangle
// inequality of the Hamming distance), so if gmin[t] >= min_second[d] + d no group mate is closer
// than min_second[d] and the exact second best passes the ratio test exactly when the scan's upper
// bound did.  One byte per descriptor (clamped to 255: still a lower bound), computed once per
// database (vsm_db_upload / vsm_db_attach); a job's table is 2 KB, so the lookups stay in L1 where
// the refinement reads four 128-byte lines from beyond L2 per result.  Unrelated ORB descriptors
// are ~126 bits apart and true matches a few dozen: the test settles ~95 % of the results of a
// dense batch.  One thread per database row.
__global__ __launch_bounds__(256) void group_min_kernel(const u32x4* __restrict__ desc, const u32* __restrict__ img_off,
                                                        u32 n_img, u32 total, uint8_t* __restrict__ gmin) {
  const u32 r = blockIdx.x * 256 + threadIdx.x;
  if (r >= total) return;
  u32 lo = 0, hi = n_img;  // img_off[lo] <= r < img_off[hi]
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (img_off[mid] <= r) lo = mid;
    else hi = mid;
  }
  const u32 o = img_off[lo], nt = img_off[lo + 1] - o, t = r - o;
  const u32 b0 = (t & ~31u) + (t & 4u);  // refine_second's group with t0 = 0
  const u32x4 qa = desc[2 * (size_t)r], qb = desc[2 * (size_t)r + 1];
  u32 g = 255u;
#pragma unroll
  for (u32 k = 0; k < 16; k++) {
    const u32 row = b0 + (k & 3u) + 8u * (k >> 2);
    if (row < nt && row != t) {
      const u32x4 ta = desc[2 * (size_t)(o + row)], tb = desc[2 * (size_t)(o + row) + 1];
      g = min(g, (u32)(__popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                       __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w)));
    }
  }
  gmin[r] = (uint8_t)g;
}
// true: the scan's upper bound of the second best already decides the accept test of result v
// against train range t_off (see group_min_kernel); gmin may be null (no table: never certified)
__device__ __forceinline__ bool second_certified(const uint8_t* __restrict__ gmin, u32 t_off, uint2 v,
                                                 const uint16_t* lut) {
  const u32 d = min(v.x >> IDX_BITS, 256u);
  return gmin && (u32)gmin[t_off + (v.x & IDX_MASK)] >= (u32)lut[d] + d;
}

// keypoints.h:295-296: int(best) < threshold && dist_2_best*best <= second, plus
// "best_match == -1 can never be emitted" (distance 256).
__device__ __forceinline__ bool accepted(uint2 v, const uint16_t* lut, int threshold, u32& idx) {
  const u32 d = v.x >> IDX_BITS;
  const u32 s = min(v.y >> IDX_BITS, 256u);
  idx = v.x & IDX_MASK;
  return d < 256u && (int)d < threshold && s >= (u32)lut[d];
}

// Pair p of an n-pair chunk = jobs[p] (forward: set 1 queries) and jobs[n + p]
// (backward).  Thread i decides whether (i, best1[i]) survives the cross-check.
__device__ __forceinline__ bool cross_check(const uint2* __restrict__ scratch, const Job& f,
                                            const Job& b, u32 i, u32 splits, u32 split_stride,
                                            const uint16_t* lut, int threshold, u32& j) {
  if (i >= f.nq) return false;
  if (!accepted(load_merged(scratch, f.out_off, i, splits, split_stride), lut, threshold, j))
    return false;
  u32 back;
  if (!accepted(load_merged(scratch, b.out_off, j, splits, split_stride), lut, threshold, back))
    return false;
  return back == i;
}

__device__ __forceinline__ void load_lut(uint16_t* lut, const AcceptParams& prm) {
  for (u32 k = threadIdx.x; k < 257; k += blockDim.x) lut[k] = prm.min_second[k];
  __syncthreads();
}
// the same for the 256-thread consumer kernels, without the barrier (the caller has one before its first look-up)
__device__ __forceinline__ void load_lut_256(uint16_t* lut, const AcceptParams& prm) {
  lut[threadIdx.x] = prm.min_second[threadIdx.x];
  if (threadIdx.x == 0) lut[256] = prm.min_second[256];
}

// Unpruned paths (one stereo pair; batches whose scans are split or whose images exceed the
// nominate bitmap): folds the partial results of the train splits into split 0 like
// merge_splits_kernel and, for results of the matrix-core scan that pass the accept test
// (`refine` != 0), makes the second best exact (refine_second) -- in place, so the consumer
// kernels after it read finished results with splits = 1.  One thread per scratch entry of the
// chunk: entry e = pair e / (2 stride), direction (e / stride) & 1, query e % stride.
__global__ __launch_bounds__(256) void merge_refine_kernel(const Job* __restrict__ jobs, u32 n_pairs, u32 stride,
                                                           uint2* __restrict__ scratch, u32 splits, u32 split_stride,
                                                           const u32x4* __restrict__ desc_1, const u32x4* __restrict__ desc_2,
                                                           AcceptParams prm, u32 refine) {
  __shared__ uint16_t lut[260];
  load_lut(lut, prm);
  const u32 e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_pairs * 2 * stride) return;
  const u32 p = e / (2 * stride), rem = e - p * 2 * stride, dir = rem >= stride ? 1u : 0u, q = rem - dir * stride;
  const Job job = jobs[dir ? n_pairs + p : p];
  if (q >= job.nq) return;
  uint2 v = load_merged(scratch, 0, e, splits, split_stride);
  u32 idx;
  if (refine && accepted(v, lut, prm.threshold, idx)) {
    const u32 col = q;  // unpruned jobs have no query list
    const u32 per = (job.nt + splits - 1) / splits;
    // forward jobs take their queries from desc_1 and their train rows from desc_2, backward jobs the
    // other way round (one database: the same pointer twice)
    const u32x4* dq = dir ? desc_2 : desc_1;
    const u32x4* dt = dir ? desc_1 : desc_2;
    v.y = refine_second(dq + 2 * (size_t)(job.q_off + col), dt + 2 * (size_t)job.t_off, job.nt, per, v) << IDX_BITS;
  }
  scratch[e] = v;
}

// counts[p] = number of matches of pair p.  One 256-thread workgroup per pair.
__global__ __launch_bounds__(256) void count_kernel(const Job* __restrict__ jobs,
                                                    const uint2* __restrict__ scratch, u32 splits,
                                                    u32 split_stride, AcceptParams prm,
                                                    u32* __restrict__ counts) {
  __shared__ uint16_t lut[260];
  __shared__ u32 total;
  if (threadIdx.x == 0) total = 0;
  load_lut(lut, prm);
  const Job f = jobs[blockIdx.x], b = jobs[gridDim.x + blockIdx.x];
  u32 mine = 0;
  for (u32 i = threadIdx.x; i < f.nq; i += blockDim.x) {
    u32 j;
    mine += cross_check(scratch, f, b, i, splits, split_stride, lut, prm.threshold, j) ? 1u : 0u;
  }
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

constexpr u32 COMPS_SLOTS = 64, COMPS_SLOT_STRIDE = 16;  // nominate_kernel's comparison counters; stride in u64: 128 bytes

// Exclusive scan of the chunk's counts into global u64 offsets, continuing from
// *running (the total of the previous chunks); writes offsets[first + k] and
// finally *running = offsets[first + n].  Single 1024-thread workgroup.  With `slots`: also moves what
// nominate_kernel added to its COMPS_SLOTS counters into *comps and clears them.
__global__ __launch_bounds__(1024) void scan_counts_kernel(const u32* __restrict__ counts, u32 n,
                                                           u64* __restrict__ offsets, u64 first,
                                                           u64* __restrict__ running, u64* __restrict__ slots,
                                                           u64* __restrict__ comps) {
  if (slots && threadIdx.x < 64) {
    static_assert(COMPS_SLOTS == 64, "one slot per lane of the first wave");
    u64 v = slots[COMPS_SLOT_STRIDE * threadIdx.x];
    slots[COMPS_SLOT_STRIDE * threadIdx.x] = 0;
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if (threadIdx.x == 0) *comps += v;
  }
  // 4096 counts per round: every thread takes four consecutive ones (coalesced 16-byte loads).
  __shared__ u64 wave_sum[16];
  __shared__ u64 carry;
  if (threadIdx.x == 0) carry = *running;
  __syncthreads();
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (u32 base = 0; base < n; base += 4096) {
    const u32 k = base + 4 * threadIdx.x;
    u32 v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = k + e < n ? counts[k + e] : 0u;
    const u64 sum = (u64)v[0] + v[1] + v[2] + v[3];
    u64 inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
      const u64 up = __shfl_up(inc, o);
      if ((int)lane >= o) inc += up;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    u64 before = carry;
    for (u32 w = 0; w < wave; w++) before += wave_sum[w];
    u64 run = before + inc - sum;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (k + e < n) offsets[first + k + e] = run;
      run += v[e];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    offsets[first + n] = carry;
    *running = carry;
  }
}

// Writes pair p's matches, ascending in the first index (the reference's order,
// keypoints.h:354-360), to out_pairs[offsets[first + p] ...); entries past
// cap_pairs are dropped (the caller sees the needed total and gets
// VSM_ERR_CAPACITY).  One 256-thread workgroup per pair.
__global__ __launch_bounds__(256) void fill_kernel(const Job* __restrict__ jobs,
                                                   const uint2* __restrict__ scratch, u32 splits,
                                                   u32 split_stride, AcceptParams prm,
                                                   const u64* __restrict__ offsets, u64 first,
                                                   int2* __restrict__ out_pairs, u64 cap_pairs) {
  __shared__ uint16_t lut[260];
  __shared__ u32 wave_cnt[4];
  load_lut(lut, prm);
  const Job f = jobs[blockIdx.x], b = jobs[gridDim.x + blockIdx.x];
  const u64 o0 = offsets[first + blockIdx.x];
  if (offsets[first + blockIdx.x + 1] == o0) return;  // uniform: nothing to write
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 done = 0;
  for (u32 base = 0; base < f.nq; base += 256) {
    const u32 i = base + threadIdx.x;
    u32 j = 0;
    const bool hit = cross_check(scratch, f, b, i, splits, split_stride, lut, prm.threshold, j);
    const u64 m = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = (u32)__popcll(m);
    __syncthreads();
    u32 before = done, all = 0;
    for (u32 w = 0; w < 4; w++) {
      if (w < wave) before += wave_cnt[w];
      all += wave_cnt[w];
    }
    if (hit) {
      const u64 pos = o0 + before + (u32)__popcll(m & ((1ull << lane) - 1ull));
      if (pos < cap_pairs) out_pairs[pos] = make_int2((int)i, (int)j);
    }
    done += all;
    __syncthreads();
  }
}

// The single-pair form of merge_refine_kernel (one stereo pair is latency bound: 3072 entries x 23
// splits gave the one-thread-per-entry kernel 12 workgroups and ten dependent microseconds).  Sixteen
// lanes per entry: lane l folds splits l, l + 16, ..., a 16-lane butterfly of merge2 (associative and
// commutative) gives every lane the merged result, and if it passes the accept test each lane
// takes ONE of the best's sixteen group rows (see refine_second) and a 16-lane minimum makes the
// second best exact.  Lane 0 writes split 0.
__global__ __launch_bounds__(256) void merge_refine_wide_kernel(const Job* __restrict__ jobs, u32 stride,
                                                                uint2* __restrict__ scratch, u32 splits, u32 split_stride,
                                                                const u32x4* __restrict__ desc_1,
                                                                const u32x4* __restrict__ desc_2, AcceptParams prm,
                                                                u32 refine) {
  __shared__ uint16_t lut[260];
  load_lut(lut, prm);
  const u32 e = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const u32 dir = e >= stride ? 1u : 0u, q = e - dir * stride;
  const Job job = jobs[min(dir, 1u)];
  const bool live = e < 2 * stride && q < job.nq;  // (whole 16-lane groups agree)
  uint2 v = make_uint2(0xffffffffu, 0xffffffffu);
  if (live)
    for (u32 s = l; s < splits; s += 16) {
      const uint2 w = scratch[(size_t)s * split_stride + e];
      v = s == l ? w : merge2(v, w);
    }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const uint2 w = make_uint2((u32)__shfl_xor((int)v.x, o, 16), (u32)__shfl_xor((int)v.y, o, 16));
    v = merge2(v, w);
  }
  u32 idx;
  if (live && refine && accepted(v, lut, prm.threshold, idx)) {
    const u32 d = v.x >> IDX_BITS;
    u32 s2 = min(v.y >> IDX_BITS, 256u);
    if (s2 > d) {  // uniform over the 16 lanes
      const u32 per = (job.nt + splits - 1) / splits;
      const u32 t0 = idx / per * per, t1 = min(t0 + per, job.nt);
      const u32 r = t0 + ((idx - t0) & ~31u) + ((idx - t0) & 4u) + (l & 3u) + 8u * (l >> 2);
      const u32x4* dq = (dir ? desc_2 : desc_1) + 2 * (size_t)(job.q_off + q);
      const u32x4* dt = (dir ? desc_1 : desc_2) + 2 * ((size_t)job.t_off + min(r, t1 - 1));
      const u32x4 qa = dq[0], qb = dq[1], ta = dt[0], tb = dt[1];
      u32 dr = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
               __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
      dr = (r < t1 && r != idx) ? dr : 0x1ffu;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) dr = min(dr, (u32)__shfl_xor((int)dr, o, 16));
      v.y = min(s2, dr) << IDX_BITS;
    }
  }
  if (live && l == 0) scratch[e] = v;
}

// Single pair (vsm_match_pair, the odometry / match_stereo call): accept test, cross-check and
// ordered emission in ONE single-workgroup kernel instead of count + scan_counts + fill (three
// launches, ~22 us of a ~80 us call).  out_count[count_slot] = number of matches; pairs from out_pairs[0].
// jobs[0] = forward job, jobs[1] = backward job; scratch holds the merged results (splits folded).
// Any workgroup size up to 1024.  `flag` (may be null): after everything else has been written --
// the outputs may live in mapped host memory -- *flag = seq with system scope, for a host that
// polls instead of waiting on the stream.
__global__ __launch_bounds__(1024) void pair_emit_kernel(const Job* __restrict__ jobs,
                                                         const uint2* __restrict__ scratch, u32 split_stride,
                                                         AcceptParams prm, u64* __restrict__ out_count,
                                                         int2* __restrict__ out_pairs, u64 cap_pairs, u32 count_slot,
                                                         u64* __restrict__ flag, u64 seq) {
  __shared__ uint16_t lut[260];
  __shared__ u32 wave_cnt[16];
  load_lut(lut, prm);
  const Job f = jobs[0], b = jobs[1];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  u32 done = 0;
  for (u32 base = 0; base < f.nq; base += blockDim.x) {
    const u32 i = base + threadIdx.x;
    u32 j = 0;
    const bool hit = cross_check(scratch, f, b, i, 1u, split_stride, lut, prm.threshold, j);
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
  if (threadIdx.x == 0) {
    // count_slot 1: the offsets pair {0, count} vsm_match_pair reads back; 0: the count alone
    if (count_slot) out_count[0] = 0;
    out_count[count_slot] = done;
  }
  if (flag) {
    __threadfence_system();  // this thread's pairs (and the count) before the flag, as the host sees them
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Host -> device staging of one stereo pair by a kernel that reads mapped host memory (96 KB):
// the copy stays in the compute queue, where an SDMA copy costs a cross-engine dependency
// (~6 us of a 55 us call).
__global__ __launch_bounds__(256) void stage_in_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32 n16) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = __builtin_nontemporal_load(src + i);
}

// One or two jobs handed over as kernel arguments (entry points that return before their work
// has run must not stage them in host memory they reuse; entry points that may not allocate have
// no other way).
__global__ void set_jobs_kernel(Job* __restrict__ dst, Job a, Job b, u32 n) {
  if (threadIdx.x == 0) {
    dst[0] = a;
    if (n > 1) dst[1] = b;
  }
}

// jobs[p], jobs[n_pairs + p] for pair p = (pairs[p].x, pairs[p].y) of the database.
// A pair naming an image >= n_img raises *bad and becomes an empty job pair.
__global__ void plan_pairs_kernel(const uint2* __restrict__ pairs, u32 n_pairs,
                                  const u32* __restrict__ img_off, u32 n_img, u32 stride,
                                  Job* __restrict__ jobs, u32* __restrict__ bad,
                                  u64* __restrict__ comps, u32 dirs) {
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const uint2 pr = pairs[p];
  Job f = {0, 0, 0, 0, p * 2 * stride, NO_QLIST, 0, 0}, b = {0, 0, 0, 0, p * 2 * stride + stride, NO_QLIST, 0, 0};
  if (pr.x < n_img && pr.y < n_img) {
    const u32 o1 = img_off[pr.x], n1 = img_off[pr.x + 1] - o1;
    const u32 o2 = img_off[pr.y], n2 = img_off[pr.y + 1] - o2;
    f.q_off = o1; f.nq = n1; f.t_off = o2; f.nt = n2;
    b.q_off = o2; b.nq = n2; b.t_off = o1; b.nt = n1;
    // comparisons the scan jobs evaluate: both directions (keypoints.h:256-352), or
    // the forward one only when the backward pass is pruned (nominate_kernel adds its own)
    if (n1 && n2) atomicAdd((unsigned long long*)comps, (unsigned long long)dirs * n1 * n2);
  } else {
    atomicOr(bad, 1u);
  }
  jobs[p] = f;
  jobs[n_pairs + p] = b;
}

// Pruned backward pass.  The cross-check (keypoints.h:354-360) only ever looks at
// backward results of columns j = best1[i] of forward-ACCEPTED rows i, so only
// those columns need a backward scan.  Per pair (one 256-thread workgroup):
//  1. the forward-accepted rows i, ascending, go to flist[p*stride ...) (with the key of their best) -- from the accept bits the
//     matrix-core forward scan left (mfma_scan.h store_accept_mask: 4 bytes per 32 rows instead
//     of every 8-byte result) or, after a popcount scan, from the results themselves;
//     after a matrix-core scan "accepted" is by the scan's upper bound of the second best, which can
//     only list too many rows, never too few: count_list_kernel makes the second best exact for the
//     rows that survive the cross-check and drops those that fail then;
//  2. the columns best1[i] of the listed rows are marked in an LDS bitmap, compacted ascending
//     into qlist[p*stride ...) and the backward job is rewritten to scan just that list.
// jobs[p].pad1 = number of listed rows: count_list_kernel / fill_list_kernel then read those few
// rows instead of every row's result.  Exact: every backward result the consumer kernels read is
// computed.  Needs stride <= NOMINATE_MAX_STRIDE (LDS bitmap) and splits == 1.
constexpr u32 NOMINATE_MAX_STRIDE = 65536;
constexpr u32 NOMINATE_LDS_LIST = 2048;
__global__ __launch_bounds__(256) void nominate_kernel(Job* __restrict__ jobs, u32 n_pairs,
                                                       uint2* __restrict__ scratch,
                                                       AcceptParams prm, u32 stride,
                                                       u32* __restrict__ qlist,
                                                       uint2* __restrict__ flist,
                                                       u64* __restrict__ comps,
                                                       const u32* __restrict__ accmask) {
  __shared__ uint16_t lut[260];
  __shared__ u32 bitmap[NOMINATE_MAX_STRIDE / 32];
  __shared__ u32 wave_cnt[4];
  // the first NOMINATE_LDS_LIST listed rows are also kept in LDS between steps 1 and 2: reading them
  // back from global memory right after writing them cost a round trip per workgroup (215 us per
  // 32768 sparse pairs against 80 us)
  __shared__ u32 slist[NOMINATE_LDS_LIST];
  // (n_pairs is an argument: with gridDim.x the index of the backward job, and so the job, came through vector loads --
  // a workgroup of these kernels is a chain of dependent loads and little else, every link ~1.5 us under load)
  if (!accmask) load_lut_256(lut, prm);
  const u32 p = blockIdx.x;
  Job f = jobs[p];
  Job b = jobs[n_pairs + p];
  const u32 n2 = b.nq, words = (n2 + 31) / 32;
  for (u32 w = threadIdx.x; w < words; w += 256) bitmap[w] = 0;
  __syncthreads();
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2* list = flist + (size_t)p * stride;  // entries (row i, key of its best: distance << IDX_BITS | column), see step 2
  // position of this thread's `cnt` items among the workgroup's, after `acc` earlier ones; returns
  // the new total (two barriers)
  auto place = [&](u32 cnt, u32 acc, u32& pos) {
    u32 inc = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      const u32 up = __shfl_up(inc, o);
      if ((int)lane >= o) inc += up;
    }
    if (lane == 63) wave_cnt[wave] = inc;
    __syncthreads();
    u32 before = acc, all = 0;
    for (u32 k = 0; k < 4; k++) {
      if (k < wave) before += wave_cnt[k];
      all += wave_cnt[k];
    }
    pos = before + inc - cnt;
    __syncthreads();
    return acc + all;
  };
  u32 acc = 0;
  if (accmask) {
    // a thread takes a 32-row word of accept bits
    const u32 nw = (f.nq + 31) / 32;
    const u32* mw = accmask + (f.out_off >> 6);
    for (u32 base = 0; base < nw; base += 256) {
      const u32 w = base + threadIdx.x;
      u32 bits = w < nw ? mw[w] : 0u;
      u32 pos;
      acc = place((u32)__popc(bits), acc, pos);
      while (bits) {
        const u32 i = w * 32 + (u32)__builtin_ctz(bits);
        bits &= bits - 1;
        if (pos < NOMINATE_LDS_LIST) slist[pos] = i;
        else list[pos].x = i;
        pos++;
      }
    }
  } else {
    for (u32 base = 0; base < f.nq; base += 256) {
      const u32 i = base + threadIdx.x;
      u32 j = 0, pos;
      const bool hit = i < f.nq && accepted(scratch[f.out_off + i], lut, prm.threshold, j);
      acc = place(hit ? 1u : 0u, acc, pos);
      if (hit) {
        if (pos < NOMINATE_LDS_LIST) slist[pos] = i;
        else list[pos].x = i;
      }
    }
  }
  __syncthreads();  // the list is read below by other threads
  // mark the columns of the listed rows; the list entry takes the key along, so that the consumer kernels
  // read (row, column, distance) in one coalesced 8-byte load instead of the row's result at a random place
  // (two entries per thread and turn, both keys requested before either is used)
  for (u32 k0 = threadIdx.x; k0 < acc; k0 += 512) {
    const u32 k1 = k0 + 256;
    const bool two = k1 < acc;
    const u32 i0 = k0 < NOMINATE_LDS_LIST ? slist[k0] : list[k0].x;
    const u32 i1 = !two ? i0 : k1 < NOMINATE_LDS_LIST ? slist[k1] : list[k1].x;
    const u32 key0 = scratch[f.out_off + i0].x;
    const u32 key1 = scratch[f.out_off + i1].x;
    const u32 j0 = key0 & IDX_MASK, j1 = key1 & IDX_MASK;
    atomicOr(&bitmap[j0 >> 5], 1u << (j0 & 31));
    list[k0] = make_uint2(i0, key0);
    if (two) {
      atomicOr(&bitmap[j1 >> 5], 1u << (j1 & 31));
      list[k1] = make_uint2(i1, key1);
    }
  }
  __syncthreads();
  u32 done = 0;
  for (u32 base = 0; base < words; base += 256) {  // a thread takes a 32-column word of the bitmap
    const u32 w = base + threadIdx.x;
    u32 bits = w < words ? bitmap[w] : 0u;
    u32 pos;
    done = place((u32)__popc(bits), done, pos);
    while (bits) {
      qlist[(size_t)p * stride + pos++] = w * 32 + (u32)__builtin_ctz(bits);
      bits &= bits - 1;
    }
  }
  if (threadIdx.x == 0) {
    f.pad1 = acc;
    jobs[p] = f;
    b.nq = done;
    b.qlist_off = p * stride;
    jobs[n_pairs + p] = b;
    // one of COMPS_SLOTS counters, 128 bytes apart (scan_counts_kernel sums them): adds of every workgroup to ONE
    // word are served one after the other, ~11 ns each -- 19900 pairs with nominated columns held this kernel
    // for 220 of its 268 us
    if (done && b.nt) atomicAdd((unsigned long long*)comps + COMPS_SLOT_STRIDE * (p % COMPS_SLOTS), (unsigned long long)done * b.nt);
  }
}

// count_kernel / fill_kernel over the forward-accepted rows only (pruned path, unsplit scans).
//
// count_list_kernel: cross-check (keypoints.h:354-360) of the listed rows of pair p -- row i = flist[k].x with its
// best column j: the backward result of j must be accepted and point back at i -- then, for results of the
// matrix-core scan (`refine` bit 0: forward, bit 1: backward), the exact second best of the rows that are left
// (nominate_kernel lists rows by the scan's upper bound; an upper bound can only list too many rows, never too
// few; a third of the listed rows of a dense batch are gone after the cross-check) -- most are settled by the
// group table without reading a descriptor (second_certified).  The matches (i, j) are written, in order, over
// the first results of the pair's own FORWARD scratch range, which nothing reads afterwards: entry k of the list
// names a row i >= k and goes to a place <= k, a turn of the loop reads before it writes, and later turns read rows
// beyond every place written so far.  fill_list_kernel then only copies them to their place in the output:
// per listed row one random 8-byte read (the backward result) and two table bytes instead of the six random reads of
// the rounds before (forward result in three kernels, backward result twice): nominate + count_list + fill_list
// 0.91 -> 0.64 ms on the real-like batch (fill_list 253 -> 27 us, count_list 390 -> 348 us).
#ifndef COUNT_LIST_ENTRIES
#define COUNT_LIST_ENTRIES 2
#endif
__global__ __launch_bounds__(256, 8) void count_list_kernel(const Job* __restrict__ jobs, u32 n_pairs,
                                                         uint2* __restrict__ scratch,
                                                         const uint2* __restrict__ flist, u32 stride,
                                                         AcceptParams prm, u32* __restrict__ counts,
                                                         const u32x4* __restrict__ desc, u32 refine,
                                                         const uint8_t* __restrict__ gmin) {
  // The results the table does not settle (~5 % on a dense batch) go to a queue in LDS and are refined by TEAMS of 8
  // lanes, two group rows per lane (as merge_refine_wide_kernel does): with refine_second called where it is needed,
  // nearly every wave had a lane that needed it and ran its sixteen-row loop for that lane alone -- 890 VALU
  // instructions per wave, most of them in these loops (rocprofv3 SQ_INSTS_VALU, real-like batch).
  // What is left is a chain of dependent loads (a workgroup's ~250 VALU instructions per turn are nothing beside
  // it), so the chain is kept short: both jobs through scalar loads (n_pairs is an argument; with gridDim.x the
  // backward job came through two vector loads), the backward result as ONE 8-byte load (the accept test's
  // short-circuit had split it in two dependent ones), both table bytes requested WITH the backward result (the
  // backward one sits at row i if the entry is a match at all) and the next turn's list entry requested a turn ahead:
  // per turn list -> backward result -> its second word -> table byte -> descriptors became
  // {backward result, table bytes, next entry} -> descriptors.
  // (Eight / two entries per thread with their loads issued together were measured too, before that: 379 / 360 us
  // against 348.)
  // E entries per thread and turn (k = base + 256 e + thread), all their loads requested together: a workgroup's time is
  // its number of turns times a chain of two round trips; measured again after the chains were shortened (E = 2 against 1).
  constexpr u32 E = COUNT_LIST_ENTRIES;
  __shared__ uint16_t lut[260];
  __shared__ u32 wave_cnt[E][4];
  __shared__ u32 n_tasks;
  __shared__ u32 task_q[512 * E], task_key[512 * E], task_who[512 * E];  // who: (entry of the turn) << 1 | direction
  __shared__ u32 reject[256 * E];
  load_lut_256(lut, prm);
  const Job f = jobs[blockIdx.x], b = jobs[n_pairs + blockIdx.x];
  const uint2* list = flist + (size_t)blockIdx.x * stride;
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool table = gmin != nullptr;
  u32 done = 0;
  uint2 e_next[E];
#pragma unroll
  for (u32 e = 0; e < E; e++) e_next[e] = 256 * e + threadIdx.x < f.pad1 ? list[256 * e + threadIdx.x] : make_uint2(0u, 0u);
  for (u32 base = 0; base < f.pad1; base += 256 * E) {
    if (threadIdx.x == 0) n_tasks = 0;
#pragma unroll
    for (u32 e = 0; e < E; e++) reject[256 * e + threadIdx.x] = 0;
    __syncthreads();  // (and the accept table is there)
    u32 i[E], j[E], key[E], gf[E], gb[E];
    unsigned long long raw[E];
    bool hit[E];
#pragma unroll
    for (u32 e = 0; e < E; e++) {
      const u32 k = base + 256 * e + threadIdx.x;
      i[e] = e_next[e].x;
      key[e] = e_next[e].y;
      j[e] = key[e] & IDX_MASK;
      raw[e] = 0;
      gf[e] = gb[e] = 0;
      if (k < f.pad1) {
        raw[e] = *reinterpret_cast<const unsigned long long*>(scratch + b.out_off + j[e]);
        if (table) {
          gf[e] = gmin[f.t_off + j[e]];
          gb[e] = gmin[b.t_off + i[e]];
        }
        if (k + 256 * E < f.pad1) e_next[e] = list[k + 256 * E];
      }
    }
#pragma unroll
    for (u32 e = 0; e < E; e++) {
      const u32 k = base + 256 * e + threadIdx.x;
      hit[e] = false;
      if (k < f.pad1) {
        const uint2 vb = make_uint2((u32)raw[e], (u32)(raw[e] >> 32));
        u32 back;
        hit[e] = accepted(vb, lut, prm.threshold, back) && back == i[e];
        const u32 df = min(key[e] >> IDX_BITS, 256u), db = min(vb.x >> IDX_BITS, 256u);
        if (hit[e] && (refine & 1u) && !(table && gf[e] >= (u32)lut[df] + df)) {  // second_certified, forward
          const u32 t = atomicAdd(&n_tasks, 1u);
          task_q[t] = f.q_off + i[e];
          task_key[t] = key[e];
          task_who[t] = (256 * e + threadIdx.x) << 1;
        }
        if (hit[e] && (refine & 2u) && !(table && gb[e] >= (u32)lut[db] + db)) {  // backward: its best is row i
          const u32 t = atomicAdd(&n_tasks, 1u);
          task_q[t] = b.q_off + j[e];
          task_key[t] = vb.x;
          task_who[t] = (256 * e + threadIdx.x) << 1 | 1u;
        }
      }
    }
    __syncthreads();
    // a team per task: the fifteen group mates of the best (refine_second: rows b0 + (r & 3) + 8 (r >> 2) of the
    // best's 32-row step, r = 0 .. 15, unsplit scan), lane l of the team takes r = l and l + 8; a closer one than
    // the accept test's least second best rejects the entry (the stored second best passed the test already)
    const u32 nt_ = n_tasks, team = threadIdx.x >> 3, l = threadIdx.x & 7;
    for (u32 t = team; t < nt_; t += 32) {
      const u32 who = task_who[t], tkey = task_key[t], idx = tkey & IDX_MASK, d = tkey >> IDX_BITS;
      const u32 t_off = (who & 1u) ? b.t_off : f.t_off, ntr = (who & 1u) ? b.nt : f.nt;
      const u32 r0 = (idx & ~31u) + (idx & 4u) + (l & 3u) + 8u * (l >> 2), r1 = r0 + 16u;
      const u32x4* dq = desc + 2 * (size_t)task_q[t];
      const u32x4* d0 = desc + 2 * ((size_t)t_off + min(r0, ntr - 1));
      const u32x4* d1 = desc + 2 * ((size_t)t_off + min(r1, ntr - 1));
      const u32x4 qa = dq[0], qb = dq[1], ta = d0[0], tb = d0[1], ua = d1[0], ub = d1[1];
      u32 dr0 = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
      u32 dr1 = __popc(qa.x ^ ua.x) + __popc(qa.y ^ ua.y) + __popc(qa.z ^ ua.z) + __popc(qa.w ^ ua.w) +
                __popc(qb.x ^ ub.x) + __popc(qb.y ^ ub.y) + __popc(qb.z ^ ub.z) + __popc(qb.w ^ ub.w);
      dr0 = (r0 < ntr && r0 != idx) ? dr0 : 0x1ffu;
      dr1 = (r1 < ntr && r1 != idx) ? dr1 : 0x1ffu;
      u32 dr = min(dr0, dr1);
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) dr = min(dr, (u32)__shfl_xor((int)dr, o, 8));
      if (l == 0 && dr < (u32)lut[min(d, 256u)]) reject[who >> 1] = 1;
    }
    __syncthreads();
    u64 m[E];
#pragma unroll
    for (u32 e = 0; e < E; e++) {
      hit[e] = hit[e] && !reject[256 * e + threadIdx.x];
      m[e] = __ballot(hit[e]);
      if (lane == 0) wave_cnt[e][wave] = (u32)__popcll(m[e]);
    }
    __syncthreads();  // every read of this turn is through
#pragma unroll
    for (u32 e = 0; e < E; e++) {  // in order: the entries e of every thread before the entries e + 1
      u32 before = done, all = 0;
      for (u32 w = 0; w < 4; w++) {
        if (w < wave) before += wave_cnt[e][w];
        all += wave_cnt[e][w];
      }
      if (hit[e]) scratch[f.out_off + before + (u32)__popcll(m[e] & ((1ull << lane) - 1ull))] = make_uint2(i[e], j[e]);
      done += all;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = done;
}

// The matches count_list_kernel left at the head of the pair's forward scratch range, to their place in the output.
__global__ __launch_bounds__(256) void fill_list_kernel(const Job* __restrict__ jobs,
                                                        const uint2* __restrict__ scratch,
                                                        const u64* __restrict__ offsets,
                                                        u64 first, int2* __restrict__ out_pairs, u64 cap_pairs) {
  const u64 o0 = offsets[first + blockIdx.x];
  const u32 n = (u32)(offsets[first + blockIdx.x + 1] - o0);
  const uint2* src = scratch + jobs[blockIdx.x].out_off;
  for (u32 k = threadIdx.x; k < n; k += 256) {
    const uint2 e = src[k];
    if (o0 + k < cap_pairs) out_pairs[o0 + k] = make_int2((int)e.x, (int)e.y);
  }
}

// Unpacks scan results (keypoints.h:259-262 conventions: idx -1 / 256 / 256
// when nothing closer than 256 exists); after a matrix-core scan (`refine`) the second best of
// every query is made exact first.
__global__ void finalize_nn2_kernel(const uint2* __restrict__ scratch, u32 nq, u32 splits,
                                    u32 split_stride, int* __restrict__ best_idx,
                                    int* __restrict__ best_dist, int* __restrict__ second_dist,
                                    const u32x4* __restrict__ query, const u32x4* __restrict__ train, u32 nt,
                                    u32 refine) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const uint2 v = load_merged(scratch, 0, i, splits, split_stride);
  const u32 d = v.x >> IDX_BITS;
  u32 s = min(v.y >> IDX_BITS, 256u);
  if (refine) s = refine_second(query + 2 * (size_t)i, train, nt, (nt + splits - 1) / splits, v);
  best_idx[i] = d < 256u ? (int)(v.x & IDX_MASK) : -1;
  best_dist[i] = (int)min(d, 256u);
  second_dist[i] = (int)s;
}

}  // namespace vsm
