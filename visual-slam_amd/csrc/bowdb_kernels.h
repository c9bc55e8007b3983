// bowdb_kernels.h -- gfx950 kernel of the BoW database query.
//
// Path: visnav::BowDatabase::query, /root/reference/include/visnav/bow_db.h:63-131, as match_bow
// calls it for every image (src/sfm.cpp:1373-1392): the L1 distance of one L1-normalised BoW vector
// to every database image it shares a word with, accumulated over an inverted index
//   first common word:  ((2 + |q - v|) - |q|) - |v|        (bow_db.h:87-89)
//   every further one:  s + ((|q - v| - |q|) - |v|)        (bow_db.h:92-94)
// in the order of the QUERY's words (double additions: the order is part of the result), then the
// num_results smallest (bow_db.h:114-127).
//
// One wave = one query.  The words of a query are taken one after the other (the order above); the 64
// lanes share the entries of a word's list, which name distinct images, so no two lanes touch the same
// score within a word.  Scores live in a row of n_img doubles: in LDS (databases up to 16384 images;
// a wave's LDS operations complete in order, so a word costs one LDS round trip) or else in HBM.  The
// inverted lists of 64 words at a time are fetched into an LDS staging area first (many loads in
// flight), so that HBM latency is paid per batch, not per word: a single 940-word query against 5000
// images went from 7.6 ms (row in HBM, lists fetched word by word) to the figure in DESIGN.md.
// A database image is an index (insertion order); `limit` = how many of the first images a query
// sees, which is how match_bow's query-then-insert loop becomes one launch: image k is queried with
// limit k.  Results: rank among the images present by (score, image index) -- the reference's
// std::sort leaves the order of equal scores open (test_ex3.cpp:523-525), here the earlier image
// comes first.  For num_results <= 64 only candidates at or below the num_results-th smallest of
// the 64 lanes' minima are ranked (an upper bound of the cut: those minima are distinct candidates).
// HBM/LDS-bound integer and double work, no MFMA.
#pragma once
#include "vsmatch_kernels.h"

namespace vsm {

constexpr double BOWDB_ABSENT = 1e300;  // scores are in [-eps, 2]

__device__ __forceinline__ double bowdb_readlane(double x, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane),
                          __builtin_amdgcn_readlane(__double2loint(x), lane));
}

constexpr u32 BOWDB_STAGE = 2048;  // inverted-list entries staged in LDS at a time
// LDS: st_v[STAGE] f64 | qa[64] f64 | st_f[STAGE] u32 | P[68] u32 | qb[64] u32 | row[n_img] f64 (LDS form)
constexpr u32 BOWDB_LDS_FIXED = BOWDB_STAGE * 8 + 64 * 8 + BOWDB_STAGE * 4 + 68 * 4 + 64 * 4;
constexpr u32 BOWDB_LDS_MAX_IMAGES = 16384;
static_assert(BOWDB_LDS_FIXED % 8 == 0, "the score row must be 8-byte aligned");

__device__ __forceinline__ u32 bowdb_word_of(const u32* P, u32 X) {  // largest j in 0..63 with P[j] <= X
  u32 lo = 0;
#pragma unroll
  for (u32 step = 32; step; step >>= 1)
    if (P[lo + step] <= X) lo += step;  // P[64] = total > X never passes for lo + step == 64
  return lo;
}

// grid = queries of this launch; block = 64.  rows (HBM form only) / cand_* / surv_*: one stretch of
// n_img per query of the launch.
template <bool LDSROW>
__global__ __launch_bounds__(64) void bowdb_query_kernel(
    const u32* __restrict__ q_off, const u32* __restrict__ q_words, const double* __restrict__ q_vals, u32 q0,
    const u32* __restrict__ limit, u32 limit_iota, const u32* __restrict__ inv_start, u32 n_dense,
    const u32* __restrict__ inv_img, const double* __restrict__ inv_val, u32 n_img, double* rows,
    u32* __restrict__ cand_idx, double* __restrict__ cand_score, u32* __restrict__ surv_idx,
    double* __restrict__ surv_score, u32 num_results, int* __restrict__ out_idx, double* __restrict__ out_score,
    u32* __restrict__ out_n) {
  extern __shared__ __align__(16) unsigned char bowdb_smem[];
  double* st_v = (double*)bowdb_smem;
  double* qa = st_v + BOWDB_STAGE;
  u32* st_f = (u32*)(qa + 64);
  u32* P = st_f + BOWDB_STAGE;
  u32* qb = P + 68;
  const u32 lane = threadIdx.x;
  const u32 q = q0 + blockIdx.x;
  u32 lim = limit_iota ? q : (limit ? limit[q] : n_img);
  if (lim > n_img) lim = n_img;
  double* row = LDSROW ? (double*)(qb + 64) : rows + (size_t)blockIdx.x * n_img;
  u32* cidx = cand_idx + (size_t)blockIdx.x * n_img;
  double* cscore = cand_score + (size_t)blockIdx.x * n_img;
  for (u32 f = lane; f < lim; f += 64) row[f] = BOWDB_ABSENT;
  __syncthreads();

  const u32 kb = q_off[q], ke = q_off[q + 1];
  for (u32 k0 = kb; k0 < ke; k0 += 64) {
    // 64 words: list bounds with one load per lane, running sum of the list lengths
    u32 b = 0, len = 0;
    double a = 0;
    if (k0 + lane < ke) {
      const u32 w = q_words[k0 + lane];
      a = q_vals[k0 + lane];
      if (w < n_dense) {
        b = inv_start[w];
        len = inv_start[w + 1] - b;  // 0: a word no database image has (bow_db.h:81-83)
      }
    }
    u32 incl = len;
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
      const u32 t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    P[lane] = incl - len;
    if (lane == 63) P[64] = incl;
    qa[lane] = a;
    qb[lane] = b;
    const u32 total = __builtin_amdgcn_readlane(incl, 63);
    __syncthreads();
    for (u32 X0 = 0; X0 < total; X0 += BOWDB_STAGE) {
      const u32 nS = min(BOWDB_STAGE, total - X0);
      for (u32 x = lane; x < nS; x += 64) {  // the next nS entries, in word order, into LDS
        const u32 j = bowdb_word_of(P, X0 + x);
        const u32 i = qb[j] + (X0 + x - P[j]);
        st_f[x] = inv_img[i];
        st_v[x] = inv_val[i];
      }
      __syncthreads();
      const u32 j_lo = bowdb_word_of(P, X0), j_hi = bowdb_word_of(P, X0 + nS - 1);
      for (u32 j = j_lo; j <= j_hi; j++) {  // word by word: the order of the additions
        const u32 p0 = max(P[j], X0), p1 = min(P[j + 1], X0 + nS);
        if (p0 >= p1) continue;  // uniform
        const double aj = qa[j];
        for (u32 x = p0 - X0 + lane; x < p1 - X0; x += 64) {
          const u32 f = st_f[x];
          if (f < lim) {
            const double v = st_v[x];
            const double s = row[f];
            row[f] = (s == BOWDB_ABSENT) ? 2 + fabs(aj - v) - fabs(aj) - fabs(v) : s + (fabs(aj - v) - fabs(aj) - fabs(v));
          }
        }
        __syncthreads();  // the next word may reach the same image from another lane
      }
    }
    __syncthreads();
  }

  // the images present, in image order
  u32 count = 0;
  for (u32 f0 = 0; f0 < lim; f0 += 64) {
    const u32 f = f0 + lane;
    const double s = f < lim ? row[f] : BOWDB_ABSENT;
    const bool present = s != BOWDB_ABSENT;
    const unsigned long long mask = __ballot(present);
    if (present) {
      const u32 pos = count + __popcll(mask & ((1ull << lane) - 1));
      cidx[pos] = f;
      cscore[pos] = s;
    }
    count += __popcll(mask);
  }
  __syncthreads();
  const u32 keep = min(num_results, count);
  if (lane == 0) out_n[q] = keep;
  if (keep == 0) return;

  const u32* ridx = cidx;  // what gets ranked: every candidate, or the survivors of the bound
  const double* rscore = cscore;
  u32 rcount = count;
  if (keep <= 64 && count > 256) {
    double mine = BOWDB_ABSENT;  // lanes 0 .. min(count, 64) - 1 all hold a candidate
    for (u32 i = lane; i < count; i += 64) mine = fmin(mine, cscore[i]);
    u32 rank = 0;
    for (u32 l = 0; l < 64; l++) {
      const double t = bowdb_readlane(mine, l);
      rank += (t < mine || (t == mine && l < lane)) ? 1u : 0u;
    }
    const unsigned long long at = __ballot(rank == keep - 1);  // exactly one lane
    const double bound = bowdb_readlane(mine, __ffsll((long long)at) - 1);
    u32* sidx = surv_idx + (size_t)blockIdx.x * n_img;
    double* sscore = surv_score + (size_t)blockIdx.x * n_img;
    u32 n_s = 0;
    for (u32 i0 = 0; i0 < count; i0 += 64) {
      const u32 i = i0 + lane;
      const double s = i < count ? cscore[i] : BOWDB_ABSENT;
      const bool in = s <= bound;
      const unsigned long long mask = __ballot(in);
      if (in) {
        const u32 pos = n_s + __popcll(mask & ((1ull << lane) - 1));
        sidx[pos] = cidx[i];
        sscore[pos] = s;
      }
      n_s += __popcll(mask);
    }
    __syncthreads();
    ridx = sidx;
    rscore = sscore;
    rcount = n_s;
  }

  // rank by (score, position); positions are in image order
  for (u32 i0 = 0; i0 < rcount; i0 += 64) {
    const u32 i = i0 + lane;
    const double s = i < rcount ? rscore[i] : 0.0;
    u32 rank = 0;
    for (u32 j0 = 0; j0 < rcount; j0 += 64) {
      const double t = j0 + lane < rcount ? rscore[j0 + lane] : BOWDB_ABSENT;
      const u32 nj = min(64u, rcount - j0);
      for (u32 jj = 0; jj < nj; jj++) {
        const double tj = bowdb_readlane(t, jj);
        rank += (tj < s || (tj == s && j0 + jj < i)) ? 1u : 0u;
      }
    }
    if (i < rcount && rank < keep) {
      out_idx[(size_t)q * num_results + rank] = (int)ridx[i];
      out_score[(size_t)q * num_results + rank] = s;
    }
  }
}

}  // namespace vsm
