/*
 * vsm_oracle.c -- CPU restatement of visnav::matchDescriptors.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library,
 * the C++ shim, the Python host module) may link, load or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  This restatement is checked (tests/test_oracle.py)
 *   - against the reference's own golden vector test/ex3_test_data/
 *     {kd0,kd1,matches}.json (test_ex3.cpp:148-185), committed as
 *     tests/golden/ex3_*.npy, and
 *   - against outputs of the real reference function (oracle/_ref, built by
 *     oracle/Makefile from /root/reference/include/visnav/keypoints.h:247-361)
 *     on a parameter sweep, edge cases and planted-random sets, committed as
 *     tests/golden/ref_cases.json by tests/golden/make_golden.py.
 *
 * Descriptor layout: one descriptor = 4 little-endian uint64 words; bit k of
 * the reference's std::bitset<256> (keypoints.h:53) is bit (k % 64) of word
 * k / 64, i.e. exactly the libstdc++ in-memory image of the bitset.
 *
 * Plain C11, no dependencies.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define VSM_WORDS 4

/* popcount(a ^ b) over 256 bits: std::bitset::operator^ + count(),
 * keypoints.h:269-273 / 318-322. */
static inline int ham256(const uint64_t *a, const uint64_t *b) {
  return __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
         __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
}

/*
 * One direction of the scan, keypoints.h:256-290 (forward) and, with the
 * roles swapped, 305-339 (backward).  For every query (ascending) scan all
 * train descriptors (ascending) keeping the smallest and the second smallest
 * distance; strict '<' so the lowest train index wins ties; both start at 256
 * and the index at -1 (keypoints.h:259-262), so a distance of 256 is never
 * recorded.
 */
void vsm_oracle_nn2(const uint64_t *query, size_t nq, const uint64_t *train,
                    size_t nt, int32_t *best_idx, int32_t *best_dist,
                    int32_t *second_dist) {
  for (size_t q = 0; q < nq; q++) {
    int idx = -1;
    size_t best = 256, second = 256;
    const uint64_t *qd = query + VSM_WORDS * q;
    for (size_t t = 0; t < nt; t++) {
      size_t d = (size_t)ham256(qd, train + VSM_WORDS * t);
      if (d < best) { /* keypoints.h:273-278 */
        second = best;
        idx = (int)t;
        best = d;
      } else if (d < second) { /* keypoints.h:283-285 */
        second = d;
      }
    }
    best_idx[q] = idx;
    best_dist[q] = (int32_t)best;
    second_dist[q] = (int32_t)second;
  }
}

/* Accept test, keypoints.h:295-296 / 344-345: evaluated in double. */
static inline int accept(int32_t best, int32_t second, int threshold,
                         double dist_2_best) {
  return (int)best < threshold && dist_2_best * (int)best <= (int)second;
}

/*
 * Full matcher, keypoints.h:247-361.  Returns the number of matches the
 * reference would hold in `matches`; the first min(count, cap) of them are
 * written to out_pairs as (first, second) int32 pairs, in reference order.
 * Returns (size_t)-1 on allocation failure.
 */
size_t vsm_oracle_match(const uint64_t *d1, size_t n1, const uint64_t *d2,
                        size_t n2, int threshold, double dist_2_best,
                        int32_t *out_pairs, size_t cap) {
  size_t n = n1 > n2 ? n1 : n2;
  int32_t *idx = (int32_t *)malloc(3 * (n + 1) * sizeof(int32_t));
  int32_t *m1 = (int32_t *)malloc(2 * (n1 + 1) * sizeof(int32_t));
  int32_t *m2 = (int32_t *)malloc(2 * (n2 + 1) * sizeof(int32_t));
  if (!idx || !m1 || !m2) {
    free(idx);
    free(m1);
    free(m2);
    return (size_t)-1;
  }
  int32_t *bd = idx + (n + 1), *sd = bd + (n + 1);
  size_t c1 = 0, c2 = 0, count = 0;

  /* forward pass, keypoints.h:256-303 -> matches1 = (descriptor1, best) */
  vsm_oracle_nn2(d1, n1, d2, n2, idx, bd, sd);
  for (size_t i = 0; i < n1; i++)
    if (accept(bd[i], sd[i], threshold, dist_2_best)) {
      m1[2 * c1] = (int32_t)i;
      m1[2 * c1 + 1] = idx[i];
      c1++;
    }

  /* backward pass, keypoints.h:305-352 -> matches2 = (best, descriptor2) */
  vsm_oracle_nn2(d2, n2, d1, n1, idx, bd, sd);
  for (size_t j = 0; j < n2; j++)
    if (accept(bd[j], sd[j], threshold, dist_2_best)) {
      m2[2 * c2] = idx[j];
      m2[2 * c2 + 1] = (int32_t)j;
      c2++;
    }

  /* cross-check, keypoints.h:354-360: the reference's nested loop, kept as a
   * nested loop so the CPU baseline does the reference's work. */
  for (size_t a = 0; a < c1; a++)
    for (size_t b = 0; b < c2; b++)
      if (m1[2 * a] == m2[2 * b] && m1[2 * a + 1] == m2[2 * b + 1]) {
        if (count < cap) {
          out_pairs[2 * count] = m1[2 * a];
          out_pairs[2 * count + 1] = m1[2 * a + 1];
        }
        count++;
      }

  free(idx);
  free(m1);
  free(m2);
  return count;
}

/*
 * The caller pattern of src/sfm.cpp:1295-1321 (match_all) for one pair of a
 * descriptor database: offsets[i]..offsets[i+1] are image i's descriptors.
 * Kept here so bench.py can time "pairs" with one foreign call per pair.
 */
size_t vsm_oracle_match_db_pair(const uint64_t *db, const uint32_t *offsets,
                                uint32_t img1, uint32_t img2, int threshold,
                                double dist_2_best, int32_t *out_pairs,
                                size_t cap) {
  return vsm_oracle_match(db + (size_t)VSM_WORDS * offsets[img1],
                          offsets[img1 + 1] - offsets[img1],
                          db + (size_t)VSM_WORDS * offsets[img2],
                          offsets[img2 + 1] - offsets[img2], threshold,
                          dist_2_best, out_pairs, cap);
}
