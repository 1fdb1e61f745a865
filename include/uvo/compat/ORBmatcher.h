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
 * The other search loops follow the same pattern (read the members the reference reads, marshal, call, write back):
 *   SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)   src/ORBmatcher.cc:1622-1746
 *   SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)                    :155-284
 *   SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)                       :715-850
 *   SearchForTriangulation(pKF1, pKF2, F12, keys1, keys2, pairs)         :852-1014
 *   Fuse(KeyFrame*, vpMapPoints, th)                                     :1016-1134 (search on the GPU, map mutation here)
 *   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)          :286-407   (loop closing)
 *   Fuse(KeyFrame*, Scw, vpPoints, th)                                   :1136-1265 (loop closing; map mutation here)
 *   SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)             :1267-1505
 * the two loops of the LocalMapping thread in batched form (one device round trip per loop instead of one to three per key frame):
 *   SearchForTriangulationBegin / SearchForTriangulationNext             the loop of src/LocalMapping.cc:1058-1080
 *   FuseTargets(vpTargetKFs, vpMapPoints, th)                            the loop of src/LocalMapping.cc:1228-1236
 * and the four members nothing in the reference calls (kept so that the class is complete):
 *   WindowSearch(F1, F2, windowSize, vpMapPointMatches2, minOctave, maxOctave)   :409-516
 *   SearchByProjection(F1, F2, windowSize, vpMapPointMatches2)                   :519-594
 *   SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)      :598-713
 *   SearchByProjection(CurrentFrame, LastFrame, th)                              :1507-1620
 */
#ifndef UVO_COMPAT_ORBMATCHER_H_
#define UVO_COMPAT_ORBMATCHER_H_

#include <cstring>
#include <map>
#include <set>
#include <string>
#include <utility>
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

  /* int ORBmatcher::SearchByProjection(FrameKTL &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, float th, int ORBdist) */
  template <class Frame, class KeyFrameT, class MapPointT>
  int SearchByProjection(Frame& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, const float th, const int ORBdist) {
    const std::vector<MapPointT*> vpMPs = pKF->GetMapPointMatches();
    const int n = (int)CurrentFrame.mvKeysUn.size(), nmp = (int)vpMPs.size();
    if (n == 0 || nmp == 0) return 0;
    if (ensure(n, nmp) != UVO_OK) return 0;
    uvo_camera_pose cam;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) cam.rcw[3 * r + c] = CurrentFrame.mTcw.template at<float>(r, c);  // :1626-1627
      cam.tcw[r] = CurrentFrame.mTcw.template at<float>(r, 3);
      cam.ow[r] = 0.f;  // derived by the library as -Rcw^T tcw (:1628)
    }
    cam.fx = CurrentFrame.fx, cam.fy = CurrentFrame.fy, cam.cx = CurrentFrame.cx, cam.cy = CurrentFrame.cy;
    cam.min_x = CurrentFrame.mnMinX, cam.max_x = CurrentFrame.mnMaxX, cam.min_y = CurrentFrame.mnMinY, cam.max_y = CurrentFrame.mnMaxY;
    std::vector<float> xyz((size_t)nmp * 3, 0.f), mind(nmp, 1.f), kfang(nmp, 0.f), u(nmp), v(nmp);
    std::vector<uint8_t> usable(nmp, 0), valid(nmp), mdesc((size_t)nmp * 32), fdesc((size_t)n * 32);
    std::vector<int32_t> level(nmp), assigned(n);
    for (int i = 0; i < nmp; ++i) {
      MapPointT* pMP = vpMPs[i];
      if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;  // :1642-1644
      usable[i] = 1;
      auto x3Dw = pMP->GetWorldPos();
      for (int k = 0; k < 3; ++k) xyz[(size_t)i * 3 + k] = x3Dw.template at<float>(k);
      mind[i] = pMP->GetMinDistanceInvariance();
      auto d = pMP->GetDescriptor();
      std::memcpy(&mdesc[(size_t)i * 32], d.ptr(0), 32);
      kfang[i] = pKF->GetKeyPointUn(i).angle;
    }
    const int nlev = (int)CurrentFrame.mvScaleFactors.size();
    if (uvo_project_points(m_, UVO_PROJECT_KF_RELOC, &cam, nmp, xyz.data(), nullptr, mind.data(), nullptr, nullptr, usable.data(),
                           CurrentFrame.mvScaleFactors.data(), nlev, 0.f, 0.f, valid.data(), u.data(), v.data(),
                           level.data(), nullptr) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int k = 0; k < n; ++k) {
      std::memcpy(&fdesc[(size_t)k * 32], CurrentFrame.mDescriptors.ptr(k), 32);
      assigned[k] = CurrentFrame.mvpMapPoints[k] ? 0x7fffffff : -1;  // :1690
    }
    int nmatches = 0;
    if (uvo_search_by_projection_kf(m_, reinterpret_cast<const uvo_keypoint*>(CurrentFrame.mvKeysUn.data()), n, fdesc.data(), (int)CurrentFrame.mnMinX,
                                    (int)CurrentFrame.mnMinY, (int)CurrentFrame.mnMaxX, (int)CurrentFrame.mnMaxY, assigned.data(), nmp, u.data(),
                                    v.data(), level.data(), valid.data(), mdesc.data(), kfang.data(), CurrentFrame.mvScaleFactors.data(), nlev, th,
                                    ORBdist, mbCheckOrientation ? 1 : 0, &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int k = 0; k < n; ++k)
      if (assigned[k] >= 0 && assigned[k] != 0x7fffffff) CurrentFrame.mvpMapPoints[k] = vpMPs[assigned[k]];  // :1703
    return nmatches;
  }

  /* int ORBmatcher::SearchByBoW(KeyFrame* pKF, FrameKTL &F, vector<MapPoint*> &vpMapPointMatches) */
  template <class KeyFrameT, class Frame, class MapPointT>
  int SearchByBoW(KeyFrameT* pKF, Frame& F, std::vector<MapPointT*>& vpMapPointMatches) {
    const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPointT*>(F.mvpMapPoints.size(), static_cast<MapPointT*>(NULL));
    const int n1 = (int)vpMapPointsKF.size(), n2 = (int)F.mvpMapPoints.size();
    if (n1 == 0 || n2 == 0 || ensure(n1 > n2 ? n1 : n2, 1) != UVO_OK) return 0;
    FlatFeatureVector f1(pKF->GetFeatureVector()), f2(F.mFeatVec);
    std::vector<uint8_t> d1((size_t)n1 * 32), d2((size_t)n2 * 32), usable1(n1);
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; ++i) {
      MapPointT* pMP = vpMapPointsKF[i];
      usable1[i] = pMP && !pMP->isBad();  // :188-194
      auto d = pKF->GetDescriptor(i);
      std::memcpy(&d1[(size_t)i * 32], d.ptr(0), 32);
      a1[i] = pKF->GetKeyPointUn(i).angle;
    }
    for (int k = 0; k < n2; ++k) {
      std::memcpy(&d2[(size_t)k * 32], F.mDescriptors.ptr(k), 32);
      a2[k] = F.mvKeys[k].angle;  // :225
    }
    std::vector<int32_t> match(n1, -1);
    int nmatches = 0;
    uvo_feature_vector c1 = f1.c(), c2 = f2.c();
    if (uvo_search_by_bow(m_, 0, &c1, n1, d1.data(), a1.data(), usable1.data(), &c2, n2, d2.data(), a2.data(), nullptr, mfNNratio,
                          mbCheckOrientation ? 1 : 0, match.data(), &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i = 0; i < n1; ++i)
      if (match[i] >= 0) vpMapPointMatches[match[i]] = vpMapPointsKF[i];  // :220
    return nmatches;
  }

  /* int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12) */
  template <class KeyFrameT, class MapPointT>
  int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12) {
    const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    vpMatches12 = std::vector<MapPointT*>(n1, static_cast<MapPointT*>(NULL));
    if (n1 == 0 || n2 == 0 || ensure(n1 > n2 ? n1 : n2, 1) != UVO_OK) return 0;
    FlatFeatureVector f1(pKF1->GetFeatureVector()), f2(pKF2->GetFeatureVector());
    std::vector<uint8_t> d1((size_t)n1 * 32), d2((size_t)n2 * 32), usable1(n1), usable2(n2);
    std::vector<float> a1(n1), a2(n2);
    fill_kf(pKF1, vpMapPoints1, d1, a1, usable1);
    fill_kf(pKF2, vpMapPoints2, d2, a2, usable2);
    std::vector<int32_t> match(n1, -1);
    int nmatches = 0;
    uvo_feature_vector c1 = f1.c(), c2 = f2.c();
    if (uvo_search_by_bow(m_, 1, &c1, n1, d1.data(), a1.data(), usable1.data(), &c2, n2, d2.data(), a2.data(), usable2.data(), mfNNratio,
                          mbCheckOrientation ? 1 : 0, match.data(), &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i = 0; i < n1; ++i)
      if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]];  // :790
    return nmatches;
  }

  /* int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vMatchedKeys1, vMatchedKeys2, vMatchedPairs) */
  template <class KeyFrameT, class Mat33, class KeyPointT>
  int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, const Mat33& F12, std::vector<KeyPointT>& vMatchedKeys1,
                             std::vector<KeyPointT>& vMatchedKeys2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
    static_assert(sizeof(KeyPointT) == sizeof(uvo_keypoint), "keypoint layout must be cv::KeyPoint");
    const auto vpMapPoints1 = pKF1->GetMapPointMatches();
    const auto vpMapPoints2 = pKF2->GetMapPointMatches();
    const std::vector<KeyPointT> vKeysUn1 = pKF1->GetKeyPointsUn(), vKeysUn2 = pKF2->GetKeyPointsUn();
    const int n1 = (int)vKeysUn1.size(), n2 = (int)vKeysUn2.size();
    vMatchedKeys1.clear(), vMatchedKeys2.clear(), vMatchedPairs.clear();
    if (n1 == 0 || n2 == 0 || ensure(n1 > n2 ? n1 : n2, 1) != UVO_OK) return 0;
    FlatFeatureVector f1(pKF1->GetFeatureVector()), f2(pKF2->GetFeatureVector());
    std::vector<uint8_t> d1((size_t)n1 * 32), d2((size_t)n2 * 32), has1(n1), has2(n2);
    for (int i = 0; i < n1; ++i) {
      auto d = pKF1->GetDescriptor(i);
      std::memcpy(&d1[(size_t)i * 32], d.ptr(0), 32);
      has1[i] = vpMapPoints1[i] != NULL;  // :885-889
    }
    for (int k = 0; k < n2; ++k) {
      auto d = pKF2->GetDescriptor(k);
      std::memcpy(&d2[(size_t)k * 32], d.ptr(0), 32);
      has2[k] = vpMapPoints2[k] != NULL;  // :903-905
    }
    float f12[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) f12[3 * r + c] = F12.template at<float>(r, c);
    const int nlev = pKF2->GetScaleLevels();
    std::vector<float> sigma2(nlev);
    for (int l = 0; l < nlev; ++l) sigma2[l] = pKF2->GetSigma2(l);
    std::vector<int32_t> match(n1, -1);
    int nmatches = 0;
    uvo_feature_vector c1 = f1.c(), c2 = f2.c();
    if (uvo_search_for_triangulation(m_, &c1, reinterpret_cast<const uvo_keypoint*>(vKeysUn1.data()), n1, d1.data(), has1.data(), &c2,
                                     reinterpret_cast<const uvo_keypoint*>(vKeysUn2.data()), n2, d2.data(), has2.data(), f12, sigma2.data(), nlev,
                                     mbCheckOrientation ? 1 : 0, match.data(), &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i = 0; i < n1; ++i) {  // :1000-1009
      if (match[i] < 0) continue;
      vMatchedKeys1.push_back(vKeysUn1[i]);
      vMatchedKeys2.push_back(vKeysUn2[match[i]]);
      vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)match[i]));
    }
    return nmatches;
  }

  /* int ORBmatcher::Fuse(KeyFrame *pKF, vector<MapPoint *> &vpMapPoints, float th): projection tests and the window search on the
   * GPU, the order-dependent map mutation (:1104-1118) here.  IsInKeyFrame is evaluated up front, as the reference's loop would see
   * it for every point that has not been touched by an earlier iteration (a point appears once in vpMapPoints). */
  template <class KeyFrameT, class MapPointT>
  int Fuse(KeyFrameT* pKF, std::vector<MapPointT*>& vpMapPoints, const float th = 3.0) {
    const int nmp = (int)vpMapPoints.size(), n = (int)pKF->N;
    if (nmp == 0 || n == 0 || ensure(n, nmp) != UVO_OK) return 0;
    uvo_camera_pose cam;
    auto Rcw = pKF->GetRotation();
    auto tcw = pKF->GetTranslation();
    auto Ow = pKF->GetCameraCenter();
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) cam.rcw[3 * r + c] = Rcw.template at<float>(r, c);
      cam.tcw[r] = tcw.template at<float>(r);
      cam.ow[r] = Ow.template at<float>(r);
    }
    cam.fx = pKF->fx, cam.fy = pKF->fy, cam.cx = pKF->cx, cam.cy = pKF->cy;
    cam.min_x = pKF->mnMinX, cam.max_x = pKF->mnMaxX, cam.min_y = pKF->mnMinY, cam.max_y = pKF->mnMaxY;
    const std::vector<float> vfScaleFactors = pKF->GetScaleFactors();
    std::vector<float> xyz((size_t)nmp * 3, 0.f), nrm((size_t)nmp * 3, 0.f), mind(nmp, 1.f), maxd(nmp, 1.f), u(nmp), v(nmp);
    std::vector<uint8_t> usable(nmp, 0), valid(nmp), mdesc((size_t)nmp * 32), kdesc((size_t)n * 32);
    std::vector<int32_t> level(nmp), best(nmp), bdist(nmp);
    for (int i = 0; i < nmp; ++i) {
      MapPointT* pMP = vpMapPoints[i];
      if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;  // :1031-1035
      usable[i] = 1;
      auto p = pMP->GetWorldPos();
      auto pn = pMP->GetNormal();
      for (int k = 0; k < 3; ++k) xyz[(size_t)i * 3 + k] = p.template at<float>(k), nrm[(size_t)i * 3 + k] = pn.template at<float>(k);
      mind[i] = pMP->GetMinDistanceInvariance(), maxd[i] = pMP->GetMaxDistanceInvariance();
      auto d = pMP->GetDescriptor();
      std::memcpy(&mdesc[(size_t)i * 32], d.ptr(0), 32);
    }
    const int nlev = (int)vfScaleFactors.size();
    if (uvo_project_points(m_, UVO_PROJECT_FUSE, &cam, nmp, xyz.data(), nrm.data(), mind.data(), maxd.data(), nullptr, usable.data(),
                           vfScaleFactors.data(), nlev, 0.f, 0.f, valid.data(), u.data(), v.data(), level.data(), nullptr) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    std::vector<uvo_keypoint> kps(n);
    for (int k = 0; k < n; ++k) {
      auto d = pKF->GetDescriptor(k);
      std::memcpy(&kdesc[(size_t)k * 32], d.ptr(0), 32);
      const auto kp = pKF->GetKeyPointUn(k);
      std::memcpy(&kps[k], &kp, sizeof(uvo_keypoint));
    }
    if (uvo_fuse(m_, kps.data(), n, kdesc.data(), (int)pKF->mnMinX, (int)pKF->mnMinY, (int)pKF->mnMaxX, (int)pKF->mnMaxY, nmp, u.data(), v.data(),
                 level.data(), valid.data(), mdesc.data(), vfScaleFactors.data(), nlev, th, best.data(), bdist.data()) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    int nFused = 0;
    for (int i = 0; i < nmp; ++i) {  // :1101-1119
      if (best[i] < 0) continue;
      MapPointT* pMP = vpMapPoints[i];
      MapPointT* pMPinKF = pKF->GetMapPoint(best[i]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) pMP->Replace(pMPinKF);
      } else {
        pMP->AddObservation(pKF, best[i]);
        pKF->AddMapPoint(pMP, best[i]);
      }
      nFused++;
    }
    return nFused;
  }

  /* The loop of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:1058-1080) as one device round trip: Begin computes the descriptor
   * distances and epipolar tests of SearchForTriangulation(pKF1, vpKF2[k], vF12[k], ...) for every neighbour at once; Next(k) then is
   * the k-th call of the loop -- same outputs as SearchForTriangulation -- evaluated on the host with pKF1's map points as they are at
   * that moment (the loop gives features map points between two calls, src/LocalMapping.cc:1177).  Call Next for k = 0, 1, ... in the
   * loop's order, each before the map-point creation that follows it in the reference. */
  template <class KeyFrameT, class Mat33>
  int SearchForTriangulationBegin(KeyFrameT* pKF1, const std::vector<KeyFrameT*>& vpKF2, const std::vector<Mat33>& vF12) {
    typedef decltype(pKF1->GetKeyPointUn(0)) KeyPointT;
    static_assert(sizeof(KeyPointT) == sizeof(uvo_keypoint), "keypoint layout must be cv::KeyPoint");
    const std::vector<KeyPointT> vKeysUn1 = pKF1->GetKeyPointsUn();
    const auto vpMapPoints1 = pKF1->GetMapPointMatches();
    const int n1 = (int)vKeysUn1.size(), np = (int)vpKF2.size();
    int nmax = n1;
    for (int k = 0; k < np; ++k) nmax = (int)vpKF2[k]->N > nmax ? (int)vpKF2[k]->N : nmax;
    if (vF12.size() != vpKF2.size() || ensure(nmax, 1) != UVO_OK) return UVO_E_BADARG;
    FlatFeatureVector f1(pKF1->GetFeatureVector());
    std::vector<uint8_t> d1((size_t)n1 * 32), has1(n1);
    for (int i = 0; i < n1; ++i) {
      auto d = pKF1->GetDescriptor(i);
      std::memcpy(&d1[(size_t)i * 32], d.ptr(0), 32);
      has1[i] = vpMapPoints1[i] != NULL;
    }
    // per pair: everything the single call marshals (see SearchForTriangulation above), kept alive until the batch call returns
    std::vector<FlatFeatureVector> f2;
    std::vector<uvo_feature_vector> c2(np);
    std::vector<std::vector<KeyPointT> > keys2(np);
    std::vector<std::vector<uint8_t> > d2(np), has2(np);
    std::vector<std::vector<float> > sigma2(np);
    std::vector<uvo_triangulation_pair> pairs(np);
    f2.reserve(np);
    for (int k = 0; k < np; ++k) {
      KeyFrameT* pKF2 = vpKF2[k];
      keys2[k] = pKF2->GetKeyPointsUn();
      const auto vpMapPoints2 = pKF2->GetMapPointMatches();
      const int n2 = (int)keys2[k].size();
      d2[k].resize((size_t)n2 * 32), has2[k].resize(n2);
      for (int j = 0; j < n2; ++j) {
        auto d = pKF2->GetDescriptor(j);
        std::memcpy(&d2[k][(size_t)j * 32], d.ptr(0), 32);
        has2[k][j] = vpMapPoints2[j] != NULL;
      }
      f2.push_back(FlatFeatureVector(pKF2->GetFeatureVector()));
      c2[k] = f2.back().c();
      const int nlev = pKF2->GetScaleLevels();
      sigma2[k].resize(nlev);
      for (int l = 0; l < nlev; ++l) sigma2[k][l] = pKF2->GetSigma2(l);
      uvo_triangulation_pair& P = pairs[k];
      P.fv2 = &c2[k], P.kp2 = reinterpret_cast<const uvo_keypoint*>(keys2[k].data()), P.n2 = n2, P.desc2 = d2[k].data(), P.has_mp2 = has2[k].data();
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) P.f12[3 * r + c] = vF12[k].template at<float>(r, c);
      P.sigma2 = sigma2[k].data(), P.nlevels = nlev;
    }
    uvo_feature_vector c1 = f1.c();
    const int rc = uvo_search_for_triangulation_batch(m_, &c1, reinterpret_cast<const uvo_keypoint*>(vKeysUn1.data()), n1, d1.data(), has1.data(), np,
                                                      pairs.data());
    if (rc != UVO_OK) err_ = uvo_last_error();
    return rc;
  }
  template <class KeyFrameT, class KeyPointT>
  int SearchForTriangulationNext(KeyFrameT* pKF1, KeyFrameT* pKF2, int k, std::vector<KeyPointT>& vMatchedKeys1, std::vector<KeyPointT>& vMatchedKeys2,
                                 std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
    const auto vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<KeyPointT> vKeysUn1 = pKF1->GetKeyPointsUn(), vKeysUn2 = pKF2->GetKeyPointsUn();
    const int n1 = (int)vKeysUn1.size();
    vMatchedKeys1.clear(), vMatchedKeys2.clear(), vMatchedPairs.clear();
    std::vector<uint8_t> has1(n1);
    for (int i = 0; i < n1; ++i) has1[i] = vpMapPoints1[i] != NULL;
    std::vector<int32_t> match(n1, -1);
    int nmatches = 0;
    if (uvo_search_for_triangulation_next(m_, k, has1.data(), mbCheckOrientation ? 1 : 0, match.data(), &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i = 0; i < n1; ++i) {  // :1000-1009
      if (match[i] < 0) continue;
      vMatchedKeys1.push_back(vKeysUn1[i]);
      vMatchedKeys2.push_back(vKeysUn2[match[i]]);
      vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)match[i]));
    }
    return nmatches;
  }

  /* The loop of LocalMapping::SearchInNeighbors (src/LocalMapping.cc:1228-1236): `matcher.Fuse(pKFi, vpMapPointMatches)` for every target
   * key frame, as one device round trip.  The projection tests and best key points of every (target, map point) come from
   * uvo_fuse_batch; the loop below then is the reference's, target after target: the tests that look at the map as it is by then
   * (isBad(), IsInKeyFrame(pKFi), :1031-1035) and the mutation (:1104-1118) run here.  A map point whose descriptor an earlier
   * target's Replace() recomputed (src/MapPoint.cc:166) is searched again with its new descriptor before it is used.  Returns the sum of
   * the per-target nFused. */
  template <class KeyFrameT, class MapPointT>
  int FuseTargets(const std::vector<KeyFrameT*>& vpTargetKFs, std::vector<MapPointT*>& vpMapPoints, const float th = 3.0) {
    const int nt = (int)vpTargetKFs.size(), nmp = (int)vpMapPoints.size();
    if (nt == 0 || nmp == 0) return 0;
    int nmax = 1;
    for (int t = 0; t < nt; ++t) nmax = (int)vpTargetKFs[t]->N > nmax ? (int)vpTargetKFs[t]->N : nmax;
    if (ensure(nmax, nmp) != UVO_OK) return 0;
    std::vector<float> xyz((size_t)nmp * 3, 0.f), nrm((size_t)nmp * 3, 0.f), mind(nmp, 1.f), maxd(nmp, 1.f);
    std::vector<uint8_t> usable(nmp, 0), mdesc((size_t)nmp * 32);
    for (int i = 0; i < nmp; ++i) {
      MapPointT* pMP = vpMapPoints[i];
      if (!pMP) continue;
      usable[i] = 1;
      auto p = pMP->GetWorldPos();
      auto pn = pMP->GetNormal();
      for (int k = 0; k < 3; ++k) xyz[(size_t)i * 3 + k] = p.template at<float>(k), nrm[(size_t)i * 3 + k] = pn.template at<float>(k);
      mind[i] = pMP->GetMinDistanceInvariance(), maxd[i] = pMP->GetMaxDistanceInvariance();
      auto d = pMP->GetDescriptor();
      std::memcpy(&mdesc[(size_t)i * 32], d.ptr(0), 32);
    }
    std::vector<uvo_fuse_target> targets(nt);
    std::vector<std::vector<uvo_keypoint> > kps(nt);
    std::vector<std::vector<uint8_t> > kdesc(nt);
    std::vector<std::vector<float> > sfs(nt);
    for (int t = 0; t < nt; ++t) {
      KeyFrameT* pKF = vpTargetKFs[t];
      const int n = (int)pKF->N;
      kps[t].resize(n), kdesc[t].resize((size_t)n * 32);
      for (int k = 0; k < n; ++k) {
        auto d = pKF->GetDescriptor(k);
        std::memcpy(&kdesc[t][(size_t)k * 32], d.ptr(0), 32);
        const auto kp = pKF->GetKeyPointUn(k);
        std::memcpy(&kps[t][k], &kp, sizeof(uvo_keypoint));
      }
      sfs[t] = pKF->GetScaleFactors();
      uvo_fuse_target& T = targets[t];
      T.kp = kps[t].data(), T.n = n, T.desc = kdesc[t].data();
      T.min_x = (int)pKF->mnMinX, T.min_y = (int)pKF->mnMinY, T.max_x = (int)pKF->mnMaxX, T.max_y = (int)pKF->mnMaxY;
      fill_cam(pKF, T.cam);
      T.scale_factors = sfs[t].data(), T.nlevels = (int)sfs[t].size();
    }
    std::vector<int32_t> best((size_t)nt * nmp, -1), bdist((size_t)nt * nmp, -1);
    if (uvo_fuse_batch(m_, nt, targets.data(), nmp, xyz.data(), nrm.data(), mind.data(), maxd.data(), usable.data(), mdesc.data(), th, best.data(),
                       bdist.data()) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    std::set<MapPointT*> redescribed;  // points whose descriptor changed after the batch was computed
    int nFusedTotal = 0;
    for (int t = 0; t < nt; ++t) {
      KeyFrameT* pKF = vpTargetKFs[t];
      int32_t* bt = &best[(size_t)t * nmp];
      // the reference evaluates isBad() / IsInKeyFrame(pKF) per point as the loop reaches it; like Fuse() above, up front for the target
      std::vector<uint8_t> take(nmp, 0);
      for (int i = 0; i < nmp; ++i) {
        MapPointT* pMP = vpMapPoints[i];
        take[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF);
        if (take[i] && redescribed.count(pMP)) {  // stale row: redo this one point against this target with its current descriptor
          auto d = pMP->GetDescriptor();
          uint8_t one_valid = 0;
          float u1 = 0.f, v1 = 0.f;
          int32_t l1 = 0, b1 = -1, bd1 = -1;
          const uint8_t one_usable = 1;
          bt[i] = -1;
          if (uvo_project_points(m_, UVO_PROJECT_FUSE, &targets[t].cam, 1, &xyz[(size_t)i * 3], &nrm[(size_t)i * 3], &mind[i], &maxd[i], nullptr, &one_usable,
                                 targets[t].scale_factors, targets[t].nlevels, 0.f, 0.f, &one_valid, &u1, &v1, &l1, nullptr) == UVO_OK &&
              uvo_fuse(m_, targets[t].kp, targets[t].n, targets[t].desc, targets[t].min_x, targets[t].min_y, targets[t].max_x, targets[t].max_y, 1, &u1, &v1, &l1,
                       &one_valid, d.ptr(0), targets[t].scale_factors, targets[t].nlevels, th, &b1, &bd1) == UVO_OK)
            bt[i] = b1;
        }
      }
      for (int i = 0; i < nmp; ++i) {  // :1101-1119
        if (!take[i] || bt[i] < 0) continue;
        MapPointT* pMP = vpMapPoints[i];
        MapPointT* pMPinKF = pKF->GetMapPoint(bt[i]);
        if (pMPinKF) {
          if (!pMPinKF->isBad()) {
            pMP->Replace(pMPinKF);
            redescribed.insert(pMPinKF);
          }
        } else {
          pMP->AddObservation(pKF, bt[i]);
          pKF->AddMapPoint(pMP, bt[i]);
        }
        nFusedTotal++;
      }
    }
    return nFusedTotal;
  }

  /* int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th) */
  template <class KeyFrameT, class MatT, class MapPointT>
  int SearchByProjection(KeyFrameT* pKF, const MatT& Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched, int th) {
    const int nmp = (int)vpPoints.size(), n = (int)pKF->N;
    if (nmp == 0 || n == 0 || ensure(n, nmp) != UVO_OK) return 0;
    std::set<MapPointT*> spAlreadyFound(vpMatched.begin(), vpMatched.end());  // :306-307
    spAlreadyFound.erase(static_cast<MapPointT*>(nullptr));
    Sim3Candidates<MapPointT> c;
    if (!project_with_scw(pKF, Scw, vpPoints, spAlreadyFound, c)) return 0;
    std::vector<uvo_keypoint> kps;
    std::vector<uint8_t> kdesc;
    marshal_keyframe(pKF, kps, kdesc);
    std::vector<int32_t> matched(n);
    for (int k = 0; k < n; ++k) matched[k] = vpMatched[k] ? 0x7fffffff : -1;  // :372
    int nmatches = 0;
    if (uvo_search_by_projection_sim3(m_, kps.data(), n, kdesc.data(), (int)pKF->mnMinX, (int)pKF->mnMinY, (int)pKF->mnMaxX, (int)pKF->mnMaxY,
                                      matched.data(), nmp, c.u.data(), c.v.data(), c.level.data(), c.valid.data(), c.mdesc.data(),
                                      c.scale.data(), (int)c.scale.size(), th, &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int k = 0; k < n; ++k)
      if (matched[k] >= 0 && matched[k] != 0x7fffffff) vpMatched[k] = vpPoints[matched[k]];  // :394
    return nmatches;
  }

  /* int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint *> &vpPoints, float th) */
  template <class KeyFrameT, class MatT, class MapPointT>
  int Fuse(KeyFrameT* pKF, const MatT& Scw, const std::vector<MapPointT*>& vpPoints, float th) {
    const int nmp = (int)vpPoints.size(), n = (int)pKF->N;
    if (nmp == 0 || n == 0 || ensure(n, nmp) != UVO_OK) return 0;
    const std::set<MapPointT*> spAlreadyFound = pKF->GetMapPoints();  // :1152
    Sim3Candidates<MapPointT> c;
    if (!project_with_scw(pKF, Scw, vpPoints, spAlreadyFound, c)) return 0;
    std::vector<uvo_keypoint> kps;
    std::vector<uint8_t> kdesc;
    marshal_keyframe(pKF, kps, kdesc);
    std::vector<int32_t> best(nmp), bdist(nmp);
    if (uvo_fuse(m_, kps.data(), n, kdesc.data(), (int)pKF->mnMinX, (int)pKF->mnMinY, (int)pKF->mnMaxX, (int)pKF->mnMaxY, nmp, c.u.data(), c.v.data(),
                 c.level.data(), c.valid.data(), c.mdesc.data(), c.scale.data(), (int)c.scale.size(), th, best.data(), bdist.data()) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    int nFused = 0;
    for (int i = 0; i < nmp; ++i) {  // :1241-1258
      if (best[i] < 0) continue;
      MapPointT* pMP = vpPoints[i];
      MapPointT* pMPinKF = pKF->GetMapPoint(best[i]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) pMPinKF->Replace(pMP);
      } else {
        pMP->AddObservation(pKF, best[i]);
        pKF->AddMapPoint(pMP, best[i]);
      }
      nFused++;
    }
    return nFused;
  }

  /* int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
   *                              const cv::Mat &t12, float th) */
  template <class KeyFrameT, class MapPointT, class MatT>
  int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12, const MatT& R12, const MatT& t12,
                   float th) {
    const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    if (N1 == 0 || N2 == 0 || ensure(N1 > N2 ? N1 : N2, N1 > N2 ? N1 : N2) != UVO_OK) return 0;
    float r12[9], t12v[3], sR12[9], sR21[9], t21[3], R1w[9], t1w[3], R2w[9], t2w[3];
    auto R1 = pKF1->GetRotation(), R2 = pKF2->GetRotation();
    auto T1 = pKF1->GetTranslation(), T2 = pKF2->GetTranslation();
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        r12[3 * r + c] = R12.template at<float>(r, c), R1w[3 * r + c] = R1.template at<float>(r, c), R2w[3 * r + c] = R2.template at<float>(r, c);
      t12v[r] = t12.template at<float>(r), t1w[r] = T1.template at<float>(r), t2w[r] = T2.template at<float>(r);
    }
    if (uvo_sim3_relative(s12, r12, t12v, sR12, sR21, t21) != UVO_OK) {  // :1284-1287
      err_ = uvo_last_error();
      return 0;
    }
    std::vector<uint8_t> already1(N1, 0), already2(N2, 0);  // :1302-1314
    for (int i = 0; i < N1; ++i) {
      MapPointT* pMP = vpMatches12[i];
      if (!pMP) continue;
      already1[i] = 1;
      const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
      if (idx2 >= 0 && idx2 < N2) already2[idx2] = 1;
    }
    struct Side {
      std::vector<float> xyz, mind, maxd, u, v;
      std::vector<uint8_t> usable, valid, mdesc;
      std::vector<int32_t> level;
    } a, b;
    auto gather = [](const std::vector<MapPointT*>& pts, const std::vector<uint8_t>& already, Side& s) {
      const int n = (int)pts.size();
      s.xyz.assign((size_t)n * 3, 0.f), s.mind.assign(n, 1.f), s.maxd.assign(n, 1.f), s.u.resize(n), s.v.resize(n);
      s.usable.assign(n, 0), s.valid.resize(n), s.mdesc.assign((size_t)n * 32, 0), s.level.resize(n);
      for (int i = 0; i < n; ++i) {
        MapPointT* pMP = pts[i];
        if (!pMP || already[i] || pMP->isBad()) continue;  // :1324-1328
        s.usable[i] = 1;
        auto p = pMP->GetWorldPos();
        for (int k = 0; k < 3; ++k) s.xyz[(size_t)i * 3 + k] = p.template at<float>(k);
        s.mind[i] = pMP->GetMinDistanceInvariance(), s.maxd[i] = pMP->GetMaxDistanceInvariance();
        auto d = pMP->GetDescriptor();
        std::memcpy(&s.mdesc[(size_t)i * 32], d.ptr(0), 32);
      }
    };
    gather(vpMapPoints1, already1, a);
    gather(vpMapPoints2, already2, b);
    const std::vector<float> sf1 = pKF1->GetScaleFactors(), sf2 = pKF2->GetScaleFactors();
    uvo_camera_pose cam1 = intrinsics_of(pKF1), cam2 = intrinsics_of(pKF2);
    // the reference projects with pKF1's fx, fy, cx, cy in both directions (:1270-1273) and tests the bounds of the target key frame
    cam2.fx = cam1.fx, cam2.fy = cam1.fy, cam2.cx = cam1.cx, cam2.cy = cam1.cy;
    if (uvo_project_sim3(m_, R1w, t1w, sR21, t21, &cam2, N1, a.xyz.data(), a.mind.data(), a.maxd.data(), a.usable.data(), sf2.data(), (int)sf2.size(),
                         a.valid.data(), a.u.data(), a.v.data(), a.level.data()) != UVO_OK ||
        uvo_project_sim3(m_, R2w, t2w, sR12, t12v, &cam1, N2, b.xyz.data(), b.mind.data(), b.maxd.data(), b.usable.data(), sf1.data(), (int)sf1.size(),
                         b.valid.data(), b.u.data(), b.v.data(), b.level.data()) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    std::vector<uvo_keypoint> kps1, kps2;
    std::vector<uint8_t> kdesc1, kdesc2;
    marshal_keyframe(pKF1, kps1, kdesc1);
    marshal_keyframe(pKF2, kps2, kdesc2);
    const int32_t b1[4] = {(int32_t)pKF1->mnMinX, (int32_t)pKF1->mnMinY, (int32_t)pKF1->mnMaxX, (int32_t)pKF1->mnMaxY};
    const int32_t b2[4] = {(int32_t)pKF2->mnMinX, (int32_t)pKF2->mnMinY, (int32_t)pKF2->mnMaxX, (int32_t)pKF2->mnMaxY};
    std::vector<int32_t> match12(N1, -1);
    int nFound = 0;
    if (uvo_search_by_sim3(m_, kps1.data(), N1, kdesc1.data(), b1, kps2.data(), N2, kdesc2.data(), b2, a.u.data(), a.v.data(), a.level.data(),
                           a.valid.data(), a.mdesc.data(), b.u.data(), b.v.data(), b.level.data(), b.valid.data(), b.mdesc.data(), sf1.data(),
                           (int)sf1.size(), sf2.data(), (int)sf2.size(), th, match12.data(), &nFound) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i1 = 0; i1 < N1; ++i1)
      if (match12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[match12[i1]];  // :1489
    return nFound;
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

  /* ---- members without a caller in the reference ---- */

  /* int ORBmatcher::WindowSearch(FrameKTL &F1, FrameKTL &F2, int windowSize, vector<MapPoint*> &vpMapPointMatches2, int minOctave, int maxOctave) */
  template <class Frame, class MapPointT>
  int WindowSearch(Frame& F1, Frame& F2, int windowSize, std::vector<MapPointT*>& vpMapPointMatches2, int minScaleLevel = -1,
                   int maxScaleLevel = 0x7fffffff) {
    const int n1 = (int)F1.mvpMapPoints.size(), n2 = (int)F2.mvKeysUn.size();
    vpMapPointMatches2 = std::vector<MapPointT*>(F2.mvpMapPoints.size(), static_cast<MapPointT*>(NULL));  // :412
    if (n1 == 0 || n2 == 0 || ensure(n1 > n2 ? n1 : n2, n1) != UVO_OK) return 0;
    static_assert(sizeof(F1.mvKeysUn[0]) == sizeof(uvo_keypoint), "keypoint layout must be cv::KeyPoint");
    const uvo_keypoint* k1 = reinterpret_cast<const uvo_keypoint*>(F1.mvKeysUn.data());
    std::vector<float> qx(n1), qy(n1), qr(n1, (float)windowSize), qa(n1);
    std::vector<int32_t> lv(n1), match(n1, -1);
    std::vector<uint8_t> valid(n1), d1((size_t)n1 * 32), d2((size_t)n2 * 32);
    for (int i = 0; i < n1; ++i) {
      MapPointT* p = F1.mvpMapPoints[i];
      const int level1 = k1[i].octave;
      valid[i] = p && !p->isBad() && !(minScaleLevel > 0 && level1 < minScaleLevel) && !(maxScaleLevel < 0x7fffffff && level1 > maxScaleLevel);  // :425-441
      qx[i] = k1[i].x, qy[i] = k1[i].y, qa[i] = k1[i].angle, lv[i] = level1;
      std::memcpy(&d1[(size_t)i * 32], F1.mDescriptors.ptr(i), 32);
    }
    for (int k = 0; k < n2; ++k) std::memcpy(&d2[(size_t)k * 32], F2.mDescriptors.ptr(k), 32);
    uvo_match_rule rule = {UVO_RULE_BEST_RATIO_LEQ, TH_HIGH, mfNNratio, 1, mbCheckOrientation ? 1 : 0};
    int nmatches = 0;
    if (uvo_match_windows(m_, reinterpret_cast<const uvo_keypoint*>(F2.mvKeysUn.data()), n2, d2.data(), nullptr, (int)F2.mnMinX, (int)F2.mnMinY,
                          (int)F2.mnMaxX, (int)F2.mnMaxY, n1, qx.data(), qy.data(), qr.data(), lv.data(), lv.data(), valid.data(), d1.data(), qa.data(),
                          &rule, match.data(), nullptr, &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i = 0; i < n1; ++i)
      if (match[i] >= 0) vpMapPointMatches2[match[i]] = F1.mvpMapPoints[i];  // :477
    return nmatches;
  }

  /* int ORBmatcher::SearchByProjection(FrameKTL &F1, FrameKTL &F2, int windowSize, vector<MapPoint*> &vpMapPointMatches2) */
  template <class Frame, class MapPointT>
  int SearchByProjection(Frame& F1, Frame& F2, int windowSize, std::vector<MapPointT*>& vpMapPointMatches2) {
    vpMapPointMatches2 = F2.mvpMapPoints;  // :521
    const std::set<MapPointT*> found(vpMapPointMatches2.begin(), vpMapPointMatches2.end());
    const int n1 = (int)F1.mvpMapPoints.size(), n2 = (int)F2.mvKeysUn.size();
    if (n1 == 0 || n2 == 0 || ensure(n1 > n2 ? n1 : n2, n1) != UVO_OK) return 0;
    uvo_camera_pose cam;
    pose_of(F2, cam);
    std::vector<float> xyz((size_t)n1 * 3, 0.f), u(n1), v(n1), qr(n1, (float)windowSize), one(1, 1.f);
    std::vector<uint8_t> usable(n1, 0), valid(n1), blocked(n2), d1((size_t)n1 * 32), d2((size_t)n2 * 32);
    std::vector<int32_t> lv(n1), dummy(n1), match(n1, -1);
    for (int i = 0; i < n1; ++i) {
      MapPointT* p = F1.mvpMapPoints[i];
      lv[i] = F1.mvKeysUn[i].octave;
      std::memcpy(&d1[(size_t)i * 32], F1.mDescriptors.ptr(i), 32);
      if (!p || p->isBad() || found.count(p)) continue;  // :533-537
      usable[i] = 1;
      auto x3Dw = p->GetWorldPos();
      for (int k = 0; k < 3; ++k) xyz[(size_t)i * 3 + k] = x3Dw.template at<float>(k);
    }
    for (int k = 0; k < n2; ++k) {
      std::memcpy(&d2[(size_t)k * 32], F2.mDescriptors.ptr(k), 32);
      blocked[k] = vpMapPointMatches2[k] ? 1 : 0;  // :566
    }
    if (uvo_project_points(m_, UVO_PROJECT_PIXEL, &cam, n1, xyz.data(), nullptr, nullptr, nullptr, nullptr, usable.data(), one.data(), 1, 0.f, 0.f,
                           valid.data(), u.data(), v.data(), dummy.data(), nullptr) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    uvo_match_rule rule = {UVO_RULE_BEST_RATIO_LEQ, TH_HIGH, mfNNratio, 1, 0};
    int nmatches = 0;
    if (uvo_match_windows(m_, reinterpret_cast<const uvo_keypoint*>(F2.mvKeysUn.data()), n2, d2.data(), blocked.data(), (int)F2.mnMinX, (int)F2.mnMinY,
                          (int)F2.mnMaxX, (int)F2.mnMaxY, n1, u.data(), v.data(), qr.data(), lv.data(), lv.data(), valid.data(), d1.data(), nullptr, &rule,
                          match.data(), nullptr, &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i = 0; i < n1; ++i)
      if (match[i] >= 0) vpMapPointMatches2[match[i]] = F1.mvpMapPoints[i];  // :585
    return nmatches;
  }

  /* int ORBmatcher::SearchForInitialization(FrameKTL &F1, FrameKTL &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize) */
  template <class Frame, class Point2fT>
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<Point2fT>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) {
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);  // :601
    if (n1 == 0 || n2 == 0 || ensure(n1 > n2 ? n1 : n2, n1) != UVO_OK) return 0;
    const uvo_keypoint* k1 = reinterpret_cast<const uvo_keypoint*>(F1.mvKeysUn.data());
    const uvo_keypoint* k2 = reinterpret_cast<const uvo_keypoint*>(F2.mvKeysUn.data());
    std::vector<float> qx(n1), qy(n1), qr(n1, (float)windowSize), qa(n1);
    std::vector<int32_t> lv(n1), match(n1, -1);
    std::vector<uint8_t> valid(n1), d1((size_t)n1 * 32), d2((size_t)n2 * 32);
    for (int i = 0; i < n1; ++i) {
      lv[i] = k1[i].octave;
      valid[i] = lv[i] <= 0;  // :620-622
      qx[i] = vbPrevMatched[i].x, qy[i] = vbPrevMatched[i].y, qa[i] = k1[i].angle;
      std::memcpy(&d1[(size_t)i * 32], F1.mDescriptors.ptr(i), 32);
    }
    for (int k = 0; k < n2; ++k) std::memcpy(&d2[(size_t)k * 32], F2.mDescriptors.ptr(k), 32);
    uvo_match_rule rule = {UVO_RULE_INIT_STEAL, TH_LOW, mfNNratio, 0, mbCheckOrientation ? 1 : 0};
    int nmatches = 0;
    if (uvo_match_windows(m_, reinterpret_cast<const uvo_keypoint*>(F2.mvKeysUn.data()), n2, d2.data(), nullptr, (int)F2.mnMinX, (int)F2.mnMinY,
                          (int)F2.mnMaxX, (int)F2.mnMaxY, n1, qx.data(), qy.data(), qr.data(), lv.data(), lv.data(), valid.data(), d1.data(), qa.data(),
                          &rule, match.data(), nullptr, &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int i = 0; i < n1; ++i) {
      vnMatches12[i] = match[i];
      if (match[i] >= 0) vbPrevMatched[i].x = k2[match[i]].x, vbPrevMatched[i].y = k2[match[i]].y;  // :705-708
    }
    return nmatches;
  }

  /* int ORBmatcher::SearchByProjection(FrameKTL &CurrentFrame, const FrameKTL &LastFrame, float th) */
  template <class Frame>
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th) {
    const int n = (int)CurrentFrame.mvKeysUn.size(), nl = (int)LastFrame.mvpMapPoints.size();
    if (n == 0 || nl == 0 || ensure(n, nl) != UVO_OK) return 0;
    uvo_camera_pose cam;
    pose_of(CurrentFrame, cam);
    std::vector<float> xyz((size_t)nl * 3, 0.f), u(nl), v(nl), ang(nl, 0.f);
    std::vector<uint8_t> usable(nl, 0), valid(nl), dl((size_t)nl * 32), fdesc((size_t)n * 32);
    std::vector<int32_t> oct(nl, 0), dummy(nl), assigned(n);
    for (int i = 0; i < nl; ++i) {
      auto* pMP = LastFrame.mvpMapPoints[i];
      std::memcpy(&dl[(size_t)i * 32], LastFrame.mDescriptors.ptr(i), 32);
      oct[i] = LastFrame.mvKeys[i].octave, ang[i] = LastFrame.mvKeysUn[i].angle;  // :1547, :1592
      if (!pMP || LastFrame.mvbOutlier[i]) continue;  // :1524-1526
      usable[i] = 1;
      auto x3Dw = pMP->GetWorldPos();
      for (int k = 0; k < 3; ++k) xyz[(size_t)i * 3 + k] = x3Dw.template at<float>(k);
    }
    const int nlev = (int)CurrentFrame.mvScaleFactors.size();
    if (uvo_project_points(m_, UVO_PROJECT_PIXEL_BOUNDED, &cam, nl, xyz.data(), nullptr, nullptr, nullptr, nullptr, usable.data(),
                           CurrentFrame.mvScaleFactors.data(), nlev, 0.f, 0.f, valid.data(), u.data(), v.data(), dummy.data(), nullptr) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int k = 0; k < n; ++k) {
      std::memcpy(&fdesc[(size_t)k * 32], CurrentFrame.mDescriptors.ptr(k), 32);
      assigned[k] = CurrentFrame.mvpMapPoints[k] ? 0x7fffffff : -1;  // :1566
    }
    int nmatches = 0;
    if (uvo_search_by_projection_kf(m_, reinterpret_cast<const uvo_keypoint*>(CurrentFrame.mvKeysUn.data()), n, fdesc.data(), (int)CurrentFrame.mnMinX,
                                    (int)CurrentFrame.mnMinY, (int)CurrentFrame.mnMaxX, (int)CurrentFrame.mnMaxY, assigned.data(), nl, u.data(), v.data(),
                                    oct.data(), valid.data(), dl.data(), ang.data(), CurrentFrame.mvScaleFactors.data(), nlev, th, TH_HIGH,
                                    mbCheckOrientation ? 1 : 0, &nmatches) != UVO_OK) {
      err_ = uvo_last_error();
      return 0;
    }
    for (int k = 0; k < n; ++k)
      if (assigned[k] >= 0 && assigned[k] != 0x7fffffff) CurrentFrame.mvpMapPoints[k] = LastFrame.mvpMapPoints[assigned[k]];  // :1585
    return nmatches;
  }

 protected:
  float mfNNratio;
  bool mbCheckOrientation;

 private:
  template <class KeyFrameT>
  static void marshal_keyframe(KeyFrameT* pKF, std::vector<uvo_keypoint>& kps, std::vector<uint8_t>& kdesc) {
    const int n = (int)pKF->N;
    kps.resize(n), kdesc.resize((size_t)n * 32);
    for (int k = 0; k < n; ++k) {
      auto d = pKF->GetDescriptor(k);
      std::memcpy(&kdesc[(size_t)k * 32], d.ptr(0), 32);
      const auto kp = pKF->GetKeyPointUn(k);
      static_assert(sizeof(kp) == sizeof(uvo_keypoint), "keypoint layout must be cv::KeyPoint");
      std::memcpy(&kps[k], &kp, sizeof(uvo_keypoint));
    }
  }
  template <class KeyFrameT>
  static uvo_camera_pose intrinsics_of(KeyFrameT* pKF) {
    uvo_camera_pose cam;
    std::memset(&cam, 0, sizeof(cam));
    cam.fx = pKF->fx, cam.fy = pKF->fy, cam.cx = pKF->cx, cam.cy = pKF->cy;
    cam.min_x = pKF->mnMinX, cam.max_x = pKF->mnMaxX, cam.min_y = pKF->mnMinY, cam.max_y = pKF->mnMaxY;
    return cam;
  }
  /* the candidates of the two Scw members after :299-361 / :1145-1207: decomposition of Scw, then the Fuse-form projection tests */
  template <class MapPointT>
  struct Sim3Candidates {
    std::vector<float> u, v, scale;
    std::vector<int32_t> level;
    std::vector<uint8_t> valid, mdesc;
  };
  template <class KeyFrameT, class MatT, class MapPointT>
  bool project_with_scw(KeyFrameT* pKF, const MatT& Scw, const std::vector<MapPointT*>& vpPoints, const std::set<MapPointT*>& spAlreadyFound,
                        Sim3Candidates<MapPointT>& c) {
    const int nmp = (int)vpPoints.size();
    float scw[12];
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 4; ++k) scw[4 * r + k] = Scw.template at<float>(r, k);
    uvo_camera_pose cam = intrinsics_of(pKF);
    if (uvo_sim3_decompose(scw, 4, &cam) != UVO_OK) {
      err_ = uvo_last_error();
      return false;
    }
    c.scale = pKF->GetScaleFactors();
    std::vector<float> xyz((size_t)nmp * 3, 0.f), nrm((size_t)nmp * 3, 0.f), mind(nmp, 1.f), maxd(nmp, 1.f);
    std::vector<uint8_t> usable(nmp, 0);
    c.u.resize(nmp), c.v.resize(nmp), c.level.resize(nmp), c.valid.resize(nmp), c.mdesc.assign((size_t)nmp * 32, 0);
    for (int i = 0; i < nmp; ++i) {
      MapPointT* pMP = vpPoints[i];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;  // :315-317 / :1167-1168
      usable[i] = 1;
      auto p = pMP->GetWorldPos();
      auto pn = pMP->GetNormal();
      for (int k = 0; k < 3; ++k) xyz[(size_t)i * 3 + k] = p.template at<float>(k), nrm[(size_t)i * 3 + k] = pn.template at<float>(k);
      mind[i] = pMP->GetMinDistanceInvariance(), maxd[i] = pMP->GetMaxDistanceInvariance();
      auto d = pMP->GetDescriptor();
      std::memcpy(&c.mdesc[(size_t)i * 32], d.ptr(0), 32);
    }
    if (uvo_project_points(m_, UVO_PROJECT_FUSE, &cam, nmp, xyz.data(), nrm.data(), mind.data(), maxd.data(), nullptr, usable.data(), c.scale.data(),
                           (int)c.scale.size(), 0.f, 0.f, c.valid.data(), c.u.data(), c.v.data(), c.level.data(), nullptr) != UVO_OK) {
      err_ = uvo_last_error();
      return false;
    }
    return true;
  }

  /* DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) flattened into the three arrays the ABI takes */
  struct FlatFeatureVector {
    std::vector<uint32_t> node;
    std::vector<int32_t> start, feat;
    template <class FeatVec>
    explicit FlatFeatureVector(const FeatVec& fv) {
      start.push_back(0);
      for (typename FeatVec::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        node.push_back((uint32_t)it->first);
        for (size_t k = 0; k < it->second.size(); ++k) feat.push_back((int32_t)it->second[k]);
        start.push_back((int32_t)feat.size());
      }
    }
    uvo_feature_vector c() const {
      uvo_feature_vector v;
      v.node = node.data(), v.start = start.data(), v.feat = feat.data(), v.n_nodes = (int32_t)node.size();
      return v;
    }
  };
  template <class KeyFrameT>
  static void fill_cam(KeyFrameT* pKF, uvo_camera_pose& cam) {
    auto Rcw = pKF->GetRotation();
    auto tcw = pKF->GetTranslation();
    auto Ow = pKF->GetCameraCenter();
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) cam.rcw[3 * r + c] = Rcw.template at<float>(r, c);
      cam.tcw[r] = tcw.template at<float>(r);
      cam.ow[r] = Ow.template at<float>(r);
    }
    cam.fx = pKF->fx, cam.fy = pKF->fy, cam.cx = pKF->cx, cam.cy = pKF->cy;
    cam.min_x = pKF->mnMinX, cam.max_x = pKF->mnMaxX, cam.min_y = pKF->mnMinY, cam.max_y = pKF->mnMaxY;
  }
  template <class KeyFrameT, class MapPointT>
  static void fill_kf(KeyFrameT* pKF, const std::vector<MapPointT*>& mps, std::vector<uint8_t>& desc, std::vector<float>& angle,
                      std::vector<uint8_t>& usable) {
    for (size_t i = 0; i < mps.size(); ++i) {
      usable[i] = mps[i] && !mps[i]->isBad();
      auto d = pKF->GetDescriptor((int)i);
      std::memcpy(&desc[i * 32], d.ptr(0), 32);
      angle[i] = pKF->GetKeyPointUn((int)i).angle;
    }
  }
  /* Rcw, tcw (mTcw.rowRange(0,3).colRange(0,3) / .col(3)), intrinsics and image bounds of a frame */
  template <class Frame>
  static void pose_of(const Frame& F, uvo_camera_pose& cam) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) cam.rcw[3 * r + c] = F.mTcw.template at<float>(r, c);
      cam.tcw[r] = F.mTcw.template at<float>(r, 3);
      cam.ow[r] = 0.f;
    }
    cam.fx = F.fx, cam.fy = F.fy, cam.cx = F.cx, cam.cy = F.cy;
    cam.min_x = F.mnMinX, cam.max_x = F.mnMaxX, cam.min_y = F.mnMinY, cam.max_y = F.mnMaxY;
  }
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
