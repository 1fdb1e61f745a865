// DBoW2 vocabulary-tree descent (TemplatedVocabulary::transform, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1207-1258) on
// the device: 16 lanes per feature, each lane takes the Hamming distance (FORB::distance, FORB.cpp:81-101) to a subset of the
// current node's children, the group keeps the first minimum (`d < best_d`, children visited in order), and steps down until
// a leaf.  A 6-level, k = 10 tree is 60 distances per feature: latency-trivial, so no tiling beyond the 16-lane groups.
#include "common.hpp"

namespace uvo {

struct VocTree {
  const int32_t* child_start;
  const int32_t* children;
  const uint8_t* desc;
  const int32_t* word_id;
  const double* weight;
  int L;
};

__global__ __launch_bounds__(256) void k_bow_descend(VocTree V, const uint8_t* __restrict__ feat, int n, int levelsup,
                                                     int32_t* __restrict__ out_word, double* __restrict__ out_weight,
                                                     int32_t* __restrict__ out_node) {
  const int g = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const bool live = g < n;
  const uint4* F = reinterpret_cast<const uint4*>(feat + (int64_t)(live ? g : 0) * 32);
  const uint4 fa = F[0], fb = F[1];
  const int nid_level = V.L - levelsup;
  int nid = 0, final_id = 0, current_level = 0;
  bool nid_set = nid_level <= 0;  // :1220 root
  for (;;) {
    const int b = V.child_start[final_id], e = V.child_start[final_id + 1];
    if (b == e) break;  // leaf (:1254 isLeaf)
    ++current_level;
    uint32_t best = 0xffffffffu;  // distance << 16 | position among the children: the first minimum has the smallest key
    for (int c = b + sub; c < e; c += 16) {
      const int id = V.children[c];
      const uint4* D = reinterpret_cast<const uint4*>(V.desc + (int64_t)id * 32);
      const uint4 da = D[0], db = D[1];
      const uint32_t d = __popc(fa.x ^ da.x) + __popc(fa.y ^ da.y) + __popc(fa.z ^ da.z) + __popc(fa.w ^ da.w) + __popc(fb.x ^ db.x) +
                         __popc(fb.y ^ db.y) + __popc(fb.z ^ db.z) + __popc(fb.w ^ db.w);
      const uint32_t key = (d << 16) | (uint32_t)(c - b);
      best = key < best ? key : best;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)best, off, 64);
      best = o < best ? o : best;
    }
    final_id = V.children[b + (int)(best & 0xffffu)];
    if (current_level == nid_level) nid = final_id, nid_set = true;
  }
  if (!nid_set) nid = final_id;  // descent ended above nid_level: the reference leaves *nid unset; declared: the leaf
  if (live && sub == 0) {
    if (out_word) out_word[g] = V.word_id[final_id];
    if (out_weight) out_weight[g] = V.weight[final_id];
    if (out_node) out_node[g] = nid;
  }
}

void launch_bow_descend(hipStream_t s, const int32_t* d_child_start, const int32_t* d_children, const uint8_t* d_desc, const int32_t* d_word_id,
                        const double* d_weight, int L, const uint8_t* d_feat, int n, int levelsup, int32_t* d_word, double* d_w, int32_t* d_node) {
  VocTree V{d_child_start, d_children, d_desc, d_word_id, d_weight, L};
  hipLaunchKernelGGL(k_bow_descend, dim3((n * 16 + 255) / 256), dim3(256), 0, s, V, d_feat, n, levelsup, d_word, d_w, d_node);
}

}  // namespace uvo
