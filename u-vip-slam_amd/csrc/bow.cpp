// Host side of uvo_vocabulary_* / uvo_bow_transform (include/uvo/uvo.h): vocabulary upload, per-feature tree descent on the
// device (bow.hip), BowVector / FeatureVector assembly in the reference's insertion order
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1125-1188, BowVector.cpp:34-83, FeatureVector.cpp:31-45).
#include <cmath>
#include <map>
#include <vector>

#include "common.hpp"

namespace uvo {
void launch_bow_descend(hipStream_t s, const int32_t* d_child_start, const int32_t* d_children, const uint8_t* d_desc, const int32_t* d_word_id,
                        const double* d_weight, int L, const uint8_t* d_feat, int n, int levelsup, int32_t* d_word, double* d_w, int32_t* d_node);
}
using namespace uvo;

struct uvo_vocabulary {
  int device = 0, n_nodes = 0, L = 0, weighting = 0, normalize = 0;
  hipStream_t stream = nullptr;
  int32_t *d_child_start = nullptr, *d_children = nullptr, *d_word_id = nullptr;
  uint8_t* d_desc = nullptr;
  double* d_weight = nullptr;
  // per-call staging
  uint8_t* d_feat = nullptr;
  int32_t *d_word = nullptr, *d_node = nullptr;
  double* d_w = nullptr;
  int cap = 0;
};

template <class T>
static int v_alloc(T** p, size_t n) {
  hipError_t e = hipMalloc((void**)p, (n ? n : 1) * sizeof(T));
  if (e != hipSuccess) {
    hip_err_set(e, "hipMalloc");
    return e == hipErrorOutOfMemory ? UVO_E_NOMEM : UVO_E_HIP;
  }
  return UVO_OK;
}

extern "C" {

void uvo_vocabulary_destroy(uvo_vocabulary* v) {
  if (!v) return;
  hipSetDevice(v->device);
  if (v->stream) hipStreamSynchronize(v->stream);
  void* ptrs[] = {v->d_child_start, v->d_children, v->d_word_id, v->d_desc, v->d_weight, v->d_feat, v->d_word, v->d_node, v->d_w};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (v->stream) hipStreamDestroy(v->stream);
  delete v;
}

int uvo_vocabulary_create(const uvo_vocabulary_desc* d, uvo_vocabulary** out) {
  if (!d || !out) return fail(UVO_E_BADARG, "null pointer");
  *out = nullptr;
  if (d->n_nodes < 1 || !d->child_start || !d->descriptor || !d->word_id || !d->weight || d->L < 0 || d->weighting < 0 || d->weighting > 3 ||
      d->normalize < 0 || d->normalize > 2)
    return fail(UVO_E_BADARG, "bad vocabulary description");
  if (d->child_start[0] != 0) return fail(UVO_E_BADARG, "child_start[0] must be 0");
  for (int i = 0; i < d->n_nodes; ++i)
    if (d->child_start[i + 1] < d->child_start[i] || d->child_start[i + 1] - d->child_start[i] > 65535)
      return fail(UVO_E_BADARG, "child_start must be non-decreasing, at most 65535 children per node");
  const int nch = d->child_start[d->n_nodes];
  if (nch > 0 && !d->children) return fail(UVO_E_BADARG, "null children");
  for (int c = 0; c < nch; ++c)
    if (d->children[c] <= 0 || d->children[c] >= d->n_nodes) return fail(UVO_E_BADARG, "child id outside 1..n_nodes-1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(UVO_E_NODEVICE, "no HIP device available (no CPU fallback exists)");
  if (d->device < 0 || d->device >= ndev) return fail(UVO_E_BADARG, "device ordinal out of range");
  uvo_vocabulary* v = new uvo_vocabulary();
  v->device = d->device, v->n_nodes = d->n_nodes, v->L = d->L, v->weighting = d->weighting, v->normalize = d->normalize;
  if (hipSetDevice(v->device) != hipSuccess || hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess) {
    delete v;
    return fail(UVO_E_HIP, "stream creation failed");
  }
  const size_t N = (size_t)d->n_nodes;
  int rc;
  if ((rc = v_alloc(&v->d_child_start, N + 1)) || (rc = v_alloc(&v->d_children, (size_t)nch)) || (rc = v_alloc(&v->d_word_id, N)) ||
      (rc = v_alloc(&v->d_desc, N * 32)) || (rc = v_alloc(&v->d_weight, N))) {
    uvo_vocabulary_destroy(v);
    return rc;
  }
  if (hipMemcpy(v->d_child_start, d->child_start, (N + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
      (nch && hipMemcpy(v->d_children, d->children, (size_t)nch * 4, hipMemcpyHostToDevice) != hipSuccess) ||
      hipMemcpy(v->d_word_id, d->word_id, N * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->d_desc, d->descriptor, N * 32, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->d_weight, d->weight, N * 8, hipMemcpyHostToDevice) != hipSuccess) {
    uvo_vocabulary_destroy(v);
    return fail(UVO_E_HIP, "vocabulary upload failed");
  }
  *out = v;
  return UVO_OK;
}

int uvo_bow_transform(uvo_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id, double* word_weight, int32_t* node_id,
                      uint32_t* bow_id, double* bow_value, int bow_cap, int* n_bow, uint32_t* fv_node, int32_t* fv_start, int32_t* fv_feat,
                      int fv_cap, int* n_fv) {
  if (!v || !n_bow || !n_fv) return fail(UVO_E_BADARG, "null pointer");
  *n_bow = 0, *n_fv = 0;
  if (n < 0 || bow_cap < 0 || fv_cap < 0) return fail(UVO_E_BADARG, "negative size");
  if (fv_start && fv_cap >= 0) fv_start[0] = 0;
  if (n == 0) return UVO_OK;
  if (!desc) return fail(UVO_E_BADARG, "null descriptors");
  UVO_HIP_CHECK(hipSetDevice(v->device));
  hipStream_t s = v->stream;
  if (n > v->cap) {
    UVO_HIP_CHECK(hipStreamSynchronize(s));
    for (void* p : {(void*)v->d_feat, (void*)v->d_word, (void*)v->d_node, (void*)v->d_w})
      if (p) hipFree(p);
    v->d_feat = nullptr, v->d_word = nullptr, v->d_node = nullptr, v->d_w = nullptr;
    const size_t c = (size_t)n * 2 + 256;
    int rc;
    if ((rc = v_alloc(&v->d_feat, c * 32)) || (rc = v_alloc(&v->d_word, c)) || (rc = v_alloc(&v->d_node, c)) || (rc = v_alloc(&v->d_w, c))) return rc;
    v->cap = (int)c;
  }
  UVO_HIP_CHECK(hipMemcpyAsync(v->d_feat, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
  launch_bow_descend(s, v->d_child_start, v->d_children, v->d_desc, v->d_word_id, v->d_weight, v->L, v->d_feat, n, levelsup, v->d_word, v->d_w,
                     v->d_node);
  UVO_HIP_CHECK(hipGetLastError());
  std::vector<int32_t> wid(n), nid(n);
  std::vector<double> ww(n);
  UVO_HIP_CHECK(hipMemcpyAsync(wid.data(), v->d_word, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(nid.data(), v->d_node, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(ww.data(), v->d_w, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  if (word_id) std::copy(wid.begin(), wid.end(), word_id);
  if (word_weight) std::copy(ww.begin(), ww.end(), word_weight);
  if (node_id) std::copy(nid.begin(), nid.end(), node_id);
  // the two containers, in the reference's insertion order (TemplatedVocabulary.h:1145-1187)
  std::map<uint32_t, double> bow;
  std::map<uint32_t, std::vector<uint32_t>> fv;
  const bool tf = v->weighting == 0 || v->weighting == 1;
  for (int i = 0; i < n; ++i) {
    if (!(ww[i] > 0)) continue;  // stopped word
    if (tf) {
      bow[(uint32_t)wid[i]] += ww[i];  // BowVector::addWeight
    } else {
      bow.insert(std::make_pair((uint32_t)wid[i], ww[i]));  // addIfNotExist
    }
    fv[(uint32_t)nid[i]].push_back((uint32_t)i);  // FeatureVector::addFeature
  }
  const bool must = v->normalize != 0;
  if (tf && !bow.empty() && !must) {
    const double nd = (double)bow.size();
    for (auto& kv : bow) kv.second /= nd;
  }
  if (must) {  // BowVector::normalize
    double norm = 0.0;
    if (v->normalize == 1) {
      for (auto& kv : bow) norm += fabs(kv.second);
    } else {
      for (auto& kv : bow) norm += kv.second * kv.second;
      norm = sqrt(norm);
    }
    if (norm > 0.0)
      for (auto& kv : bow) kv.second /= norm;
  }
  if ((int)bow.size() > bow_cap || (int)fv.size() > fv_cap) return fail(UVO_E_CAPACITY, "BowVector / FeatureVector larger than the output capacity");
  if (!bow.empty() && (!bow_id || !bow_value)) return fail(UVO_E_BADARG, "null BowVector outputs");
  if (!fv.empty() && (!fv_node || !fv_start || !fv_feat)) return fail(UVO_E_BADARG, "null FeatureVector outputs");
  int k = 0;
  for (auto& kv : bow) bow_id[k] = kv.first, bow_value[k] = kv.second, ++k;
  *n_bow = k;
  int j = 0, off = 0;
  for (auto& kv : fv) {
    fv_node[j] = kv.first;
    fv_start[j] = off;
    for (uint32_t f : kv.second) fv_feat[off++] = (int32_t)f;
    ++j;
  }
  if (fv_start) fv_start[j] = off;
  *n_fv = j;
  return UVO_OK;
}

}  // extern "C"
