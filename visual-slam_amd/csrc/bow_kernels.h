// bow_kernels.h -- gfx950 kernel of the vocabulary-tree descent.
//
// Path: visnav::BowVocabulary::transformFeatureToWord, /root/reference/include/visnav/
// bow_voc.h:57-92 (called per feature by transform(), bow_voc.h:94-131, from match_bow,
// src/sfm.cpp:1379).  One lane = one feature: at every level the Hamming distances to the
// node's children (k = 10 in the reference's vocabulary), strict '<' so the first of equally
// close children wins, down to a leaf; output the leaf's word id and weight.  The BoW vector
// (accumulation in feature order, L1 normalisation, bow_voc.h:107-130) is built on the host.
#pragma once
#include "vsmatch_kernels.h"

namespace vsm {

// flags[0] |= 1 if some feature's walk would not terminate in the reference (no child closer
// than 256: child_id keeps its value and the loop spins); that lane stops where it is.
__global__ __launch_bounds__(256) void bow_words_kernel(const u32x4* __restrict__ desc, u32 n,
                                                        const u32x4* __restrict__ node_desc,
                                                        const u32* __restrict__ child_off,
                                                        const u32* __restrict__ child_idx,
                                                        const u32* __restrict__ node_id,
                                                        const u32* __restrict__ word_id,
                                                        const double* __restrict__ weight, u32 n_nodes,
                                                        u32* __restrict__ out_word, double* __restrict__ out_weight,
                                                        u32* __restrict__ flags) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u32x4 a = desc[2 * (size_t)i], b = desc[2 * (size_t)i + 1];
  u32 parent = 0, child = 0;
  for (u32 depth = 0; depth <= n_nodes; depth++) {  // a tree walk visits each node at most once
    const u32 c0 = child_off[parent], c1 = child_off[parent + 1];
    if (c1 <= c0) break;  // leaf
    u32 dist = 256;
    for (u32 c = c0; c < c1; c++) {
      const u32 node = child_idx[c];
      const u32x4 p = node_desc[2 * (size_t)node], q = node_desc[2 * (size_t)node + 1];
      const u32 d = __popc(a.x ^ p.x) + __popc(a.y ^ p.y) + __popc(a.z ^ p.z) + __popc(a.w ^ p.w) +
                    __popc(b.x ^ q.x) + __popc(b.y ^ q.y) + __popc(b.z ^ q.z) + __popc(b.w ^ q.w);
      if (d < dist) {
        dist = d;
        child = node_id[node];
      }
    }
    if (child == parent) {
      atomicOr(flags, 1u);
      break;
    }
    parent = child;
  }
  out_word[i] = word_id[parent];
  out_weight[i] = weight[parent];
}

}  // namespace vsm
