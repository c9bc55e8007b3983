/*
 * vsm_oracle_bow.c -- CPU restatement of visnav::BowVocabulary::transformFeatureToWord
 * (/root/reference/include/visnav/bow_voc.h:57-92): descent of the vocabulary tree by
 * Hamming distance.  SURVEY.md section 8f row N4.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as vsm_oracle.c).
 *
 * Parity status: PARITY UNPINNED.  The reference's goldens for this function
 * (test/ex3_test_data/bow_res.json, bow_res2.json; test_ex3.cpp:355-445) are outputs under
 * its vocabulary data/ORBvoc.cereal, a large blob that is missing from the reference checkout
 * (.MISSING_LARGE_BLOBS), so nothing the reference holds can be reproduced here.  The
 * restatement is checked only against a second, independent writing in numpy on synthetic
 * vocabularies (tests/test_bow_oracle.py).
 *
 * The tree as the caller flattens BowVocabulary::m_nodes: node n has children
 * child_idx[child_off[n] .. child_off[n+1]) (Node::children), centroid node_desc[n]
 * (Node::descriptor), Node::id node_id[n], and as a leaf word_id[n] / weight[n].
 */
#include <stddef.h>
#include <stdint.h>

static inline int ham256(const uint64_t *a, const uint64_t *b) {
  return __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
         __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
}

/* Returns 0, or -1 if the reference's loop would not terminate for some feature: at a node
 * none of whose children is closer than 256, child_id keeps its old value (bow_voc.h:75, 82-87)
 * and the walk stays where it is forever. */
int vsm_oracle_bow_words(const uint64_t *desc, size_t n, const uint64_t *node_desc, const uint32_t *child_off,
                         const uint32_t *child_idx, const uint32_t *node_id, const uint32_t *word_id,
                         const double *weight, uint32_t *out_word, double *out_weight) {
  for (size_t i = 0; i < n; i++) {
    uint32_t parent = 0, child = 0; /* bow_voc.h:74-75 */
    while (child_off[parent + 1] > child_off[parent]) { /* !isLeaf() */
      size_t dist = 256;
      for (uint32_t c = child_off[parent]; c < child_off[parent + 1]; c++) {
        const size_t d = (size_t)ham256(node_desc + 4 * (size_t)child_idx[c], desc + 4 * i);
        if (d < dist) { /* strict '<': the first of equally close children wins */
          dist = d;
          child = node_id[child_idx[c]]; /* m_nodes[child_node].id */
        }
      }
      if (child == parent) return -1;
      parent = child;
    }
    out_word[i] = word_id[parent];
    out_weight[i] = weight[parent];
  }
  return 0;
}
