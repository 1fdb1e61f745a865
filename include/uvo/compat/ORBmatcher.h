/*
 * USLAM::ORBmatcher-shaped adaptor over the uvo C ABI (include/uvo/uvo.h).
 *
 * Mirrors the part of the reference class that is on the hot path (include/ORBmatcher.h:41-88, src/ORBmatcher.cc):
 *   ORBmatcher(nnratio, checkOri), static DescriptorDistance, SearchByProjection(Frame&, vector<MapPoint*>&, th)
 * plus Utils::ratioMatching (include/utils.h:81-111) for the all-pairs matcher.
 * SearchByProjection is a template on the reference's own frame / map-point types: it reads exactly the members
 * src/ORBmatcher.cc:49-125 reads (mvKeysUn, mDescriptors, mvpMapPoints, mvScaleFactors, mnMinX/Y, mnMaxX/Y on the frame;
 * mbTrackInView, isBad(), mnTrackScaleLevel, mTrackViewCos, mTrackProjX/Y, GetDescriptor() on the map point),
 * marshals them into the ABI's SoA arrays, and writes the winners back into F.mvpMapPoints -- the greedy,
 * order-dependent assignment is reproduced exactly by the library.  All map mutation stays on the host.
 * The other Search* / Fuse members keep running the reference's own code; DESIGN.md lists them as next.
 */
#ifndef UVO_COMPAT_ORBMATCHER_H_
#define UVO_COMPAT_ORBMATCHER_H_

#include <cstring>
#include <string>
#include <vector>

#include "../uvo.h"

#ifdef UVO_COMPAT_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

/* Inside the reference tree the class name USLAM::ORBmatcher is already taken by include/ORBmatcher.h; define
 * UVO_COMPAT_MATCHER_NAME (e.g. ORBmatcherGPU) before including this header there and forward the GPU-backed members. */
#ifndef UVO_COMPAT_MATCHER_NAME
#define UVO_COMPAT_MATCHER_NAME ORBmatcher
#endif

namespace USLAM {

class UVO_COMPAT_MATCHER_NAME {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // src/ORBmatcher.cc:40-42

  UVO_COMPAT_MATCHER_NAME(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  ~UVO_COMPAT_MATCHER_NAME() { uvo_matcher_destroy(m_); }
  UVO_COMPAT_MATCHER_NAME(const UVO_COMPAT_MATCHER_NAME&) = delete;
  UVO_COMPAT_MATCHER_NAME& operator=(const UVO_COMPAT_MATCHER_NAME&) = delete;

  /* ORBmatcher::DescriptorDistance for one pair (src/ORBmatcher.cc:1794-1810): 8 x 32-bit popcount of the XOR.
   * A single 32-byte pair is host work in the reference too (src/MapPoint.cc:244); bulk distances go through
   * uvo_hamming_matrix / uvo_hamming_knn2 on the GPU. */
  static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
      uint32_t x, y;
      std::memcpy(&x, a + 4 * i, 4);
      std::memcpy(&y, b + 4 * i, 4);
      dist += __builtin_popcount(x ^ y);
    }
    return dist;
  }
#ifdef UVO_COMPAT_WITH_OPENCV
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return DescriptorDistance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }
#endif

  /* int ORBmatcher::SearchByProjection(FrameKTL &F, const vector<MapPoint*> &vpMapPoints, const float th) */
  template <class Frame, class MapPointT>
  int SearchByProjection(Frame& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3) {
    const int n = (int)F.mvKeysUn.size(), nmp = (int)vpMapPoints.size();
    if (n == 0 || nmp == 0) return 0;
    if (ensure(n, nmp) != UVO_OK) return 0;
    static_assert(sizeof(F.mvKeysUn[0]) == sizeof(uvo_keypoint), "keypoint layout must be cv::KeyPoint");
    std::vector<uint8_t> fdesc((size_t)n * 32), mdesc((size_t)nmp * 32), inview(nmp);
    std::vector<int32_t> assigned(n), level(nmp);
    std::vector<float> px(nmp), py(nmp), vc(nmp);
    for (int i = 0; i < n; ++i) {
      std::memcpy(&fdesc[(size_t)i * 32], F.mDescriptors.ptr(i), 32);
      assigned[i] = F.mvpMapPoints[i] ? 0x7fffffff : -1;  // already taken (src/ORBmatcher.cc:91)
    }
    for (int i = 0; i < nmp; ++i) {
      MapPointT* p = vpMapPoints[i];
      inview[i] = (p->mbTrackInView && !p->isBad()) ? 1 : 0;  // :58-62
      level[i] = p->mnTrackScaleLevel, vc[i] = p->mTrackViewCos, px[i] = p->mTrackProjX, py[i] = p->mTrackProjY;
      if (inview[i]) {
        auto d = p->GetDescriptor();
        std::memcpy(&mdesc[(size_t)i * 32], d.ptr(0), 32);
      }
    }
    int nmatches = 0;
    int rc = uvo_search_by_projection(m_, reinterpret_cast<const uvo_keypoint*>(F.mvKeysUn.data()), n, fdesc.data(), F.mnMinX, F.mnMinY,
                                      F.mnMaxX, F.mnMaxY, assigned.data(), nmp, px.data(), py.data(), level.data(), vc.data(), inview.data(),
                                      mdesc.data(), F.mvScaleFactors.data(), (int)F.mvScaleFactors.size(), th, mfNNratio, &nmatches);
    if (rc != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i = 0; i < n; ++i)
      if (assigned[i] >= 0 && assigned[i] != 0x7fffffff) F.mvpMapPoints[i] = vpMapPoints[assigned[i]];  // :119
    return nmatches;
  }

  /* Utils::ratioMatching (include/utils.h:81-111) on raw descriptor rows: accepted (query, train, distance) triples. */
  struct Match {
    int queryIdx, trainIdx;
    float distance;
  };
  void ratioMatching(const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, double ratio, const uint8_t* match_mask,
                     std::vector<Match>& matches) {
    matches.clear();
    if (n1 <= 0 || n2 <= 0) return;
    if (ensure(n1 > n2 ? n1 : n2, 1) != UVO_OK) return;
    std::vector<int32_t> i0(n1), i1(n1);
    std::vector<uint16_t> d0(n1), d1(n1);
    if (uvo_hamming_knn2(m_, desc1, n1, desc2, n2, match_mask, i0.data(), d0.data(), i1.data(), d1.data()) != UVO_OK) {
      err_ = uvo_last_error();
      return;
    }
    for (int m = 0; m < n1; ++m) {
      if (i1[m] < 0) continue;  // knn_matches[m].size() < 2
      if ((float)d0[m] <= (float)d1[m] * ratio) matches.push_back(Match{m, i0[m], (float)d0[m]});
    }
  }

  const std::string& last_error() const { return err_; }
  void set_device(int device) { device_ = device; }

 protected:
  float mfNNratio;
  bool mbCheckOrientation;

 private:
  int ensure(int n, int nmp) {
    if (m_ && n <= cap_n_ && nmp <= cap_mp_) return UVO_OK;
    uvo_matcher_destroy(m_);
    m_ = nullptr;
    cap_n_ = n > cap_n_ ? 2 * n : cap_n_;
    cap_mp_ = nmp > cap_mp_ ? 2 * nmp : cap_mp_;
    if (cap_n_ < 4096) cap_n_ = 4096;
    if (cap_n_ > 65535) cap_n_ = 65535;
    if (cap_mp_ < 8192) cap_mp_ = 8192;
    uvo_matcher_cfg c;
    c.max_query = cap_n_, c.max_train = cap_n_, c.max_batch = 1, c.max_map_points = cap_mp_, c.device = device_;
    int rc = uvo_matcher_create(&c, &m_);
    if (rc != UVO_OK) err_ = uvo_last_error();
    return rc;
  }
  uvo_matcher* m_ = nullptr;
  int cap_n_ = 0, cap_mp_ = 0, device_ = 0;
  std::string err_;
};

}  // namespace USLAM
#endif
