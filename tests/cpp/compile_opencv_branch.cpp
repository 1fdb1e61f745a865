// Type-check of the UVO_COMPAT_WITH_OPENCV branches of include/uvo/compat/*.h (g++ -std=c++11 -fsyntax-only, never linked or run):
// the members INTEGRATION.md says the reference host calls unchanged must have the reference's signatures
// (include/ORBextractor.h:56-58, include/ORBmatcher.h:41-88, include/Grider_FAST.h:81-83) and must compile when instantiated with
// types that declare the members the reference's FrameKTL / KeyFrame / MapPoint declare (include/FrameKTL.h:71-136,
// include/KeyFrame.h:72-260, include/MapPoint.h:47-94), with the reference's cv:: types.  OpenCV / Eigen come from the
// declaration-only stand-ins in tests/cpp/opencv_decl_stub/ (no behaviour; they pin nothing).
#define UVO_COMPAT_WITH_OPENCV 1
#include <climits>
#include <map>
#include <set>
#include <type_traits>
#include <utility>
#include <vector>

#include "uvo/compat/Grider_FAST.h"
#include "uvo/compat/ORBextractor.h"
#include "uvo/compat/ORBmatcher.h"

namespace DBoW2 {  // Thirdparty/DBoW2/DBoW2/FeatureVector.h: class FeatureVector : public std::map<NodeId, std::vector<unsigned int> >
typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;
}

class KeyFrame;
class MapPoint {  // include/MapPoint.h
 public:
  cv::Mat GetWorldPos();
  cv::Mat GetNormal();
  void AddObservation(KeyFrame* pKF, size_t idx);
  int GetIndexInKeyFrame(KeyFrame* pKF);
  bool IsInKeyFrame(KeyFrame* pKF);
  bool isBad();
  void Replace(MapPoint* pMP);
  cv::Mat GetDescriptor();
  float GetMinDistanceInvariance();
  float GetMaxDistanceInvariance();
  int Observations();
  float mTrackProjX, mTrackProjY;
  bool mbTrackInView;
  int mnTrackScaleLevel;
  float mTrackViewCos;
};
class FrameKTL {  // include/FrameKTL.h
 public:
  static float fx, fy, cx, cy;
  int N;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  DBoW2::FeatureVector mFeatVec;
  cv::Mat mDescriptors;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  cv::Mat mTcw;
  std::vector<float> mvScaleFactors;
  static int mnMinX, mnMaxX, mnMinY, mnMaxY;
  size_t size() const;
};
class KeyFrame {  // include/KeyFrame.h
 public:
  cv::Mat GetCameraCenter();
  cv::Mat GetRotation();
  cv::Mat GetTranslation();
  DBoW2::FeatureVector GetFeatureVector();
  void AddMapPoint(MapPoint* pMP, const size_t& idx);
  std::set<MapPoint*> GetMapPoints();
  std::vector<MapPoint*> GetMapPointMatches();
  MapPoint* GetMapPoint(const size_t& idx);
  cv::KeyPoint GetKeyPointUn(const size_t& idx) const;
  cv::Mat GetDescriptor(const size_t& idx);
  std::vector<cv::KeyPoint> GetKeyPointsUn() const;
  cv::Mat GetDescriptors();
  bool isBad();
  std::vector<float> GetScaleFactors() const;
  float GetSigma2(int nLevel = 1) const;
  int GetScaleLevels() const;
  float fx, fy, cx, cy;
  int mnMinX, mnMinY, mnMaxX, mnMaxY;
  int N;
};

using USLAM::ORBextractor;
using USLAM::ORBmatcher;

// ---- the reference's operator() signature, verbatim (include/ORBextractor.h:56-58) ----
typedef void (ORBextractor::*ExtractOp)(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray, Eigen::MatrixXi&, int&, bool, int);
static_assert(std::is_same<decltype(&ORBextractor::operator()), ExtractOp>::value, "operator() must have the reference's signature");
static_assert(sizeof(cv::KeyPoint) == sizeof(uvo_keypoint), "cv::KeyPoint layout");
// static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b)   (include/ORBmatcher.h:47)
static_assert(std::is_same<decltype(static_cast<int (*)(const cv::Mat&, const cv::Mat&)>(&ORBmatcher::DescriptorDistance)), int (*)(const cv::Mat&, const cv::Mat&)>::value,
              "DescriptorDistance(cv::Mat, cv::Mat)");

// ---- the call sites, with the argument lists the reference host uses ----
void tracking_call_sites(ORBextractor* mpORBextractor, const std::vector<cv::Mat>& img0pyr, FrameKTL& mCurrentFrame, FrameKTL& mLastFrame, KeyFrame* pKF,
                         std::vector<MapPoint*>& vpMapPoints, uvo_extractor* scratch) {
  // src/Tracking.cc:896-946
  int min_px_dist = 20;
  Eigen::MatrixXi grid_2d = Eigen::MatrixXi::Zero((int)(img0pyr.at(0).rows / min_px_dist) + 2, (int)(img0pyr.at(0).cols / min_px_dist) + 2);
  std::vector<cv::KeyPoint> pts0_ext;
  cv::Mat New_Descriptors;
  bool FullDetect = true;
  int num_featsneeded = 100;
  (*mpORBextractor)(img0pyr.at(0), cv::Mat(), pts0_ext, New_Descriptors, grid_2d, min_px_dist, FullDetect, num_featsneeded);
  (void)mpORBextractor->GetLevels();
  (void)mpORBextractor->GetScaleFactor();
  // src/Tracking.cc:940 (commented out upstream): Grider_FAST::perform_griding(img0pyr.at(0), pts0_ext, num_featsneeded, grid_x, grid_y, fastTh, true)
  USLAM::Grider_FAST::perform_griding(scratch, img0pyr.at(0), pts0_ext, num_featsneeded, 8, 5, 20, true);

  ORBmatcher matcher(0.8);
  int th = 1;
  (void)matcher.SearchByProjection(mCurrentFrame, vpMapPoints, th);                              // src/Tracking.cc:2228
  (void)ORBmatcher::DescriptorDistance(mCurrentFrame.mDescriptors.row(0), mLastFrame.mDescriptors.row(0));   // src/MapPoint.cc:244
  std::set<MapPoint*> sFound;
  (void)matcher.SearchByProjection(mCurrentFrame, pKF, sFound, 10, 100);                         // src/Tracking.cc:2561
  std::vector<MapPoint*> vpMapPointMatches;
  (void)matcher.SearchByBoW(pKF, mCurrentFrame, vpMapPointMatches);                              // src/Tracking.cc:2500
  // the members nothing upstream calls
  (void)matcher.SearchByProjection(mCurrentFrame, mLastFrame, 7.f);                              // include/ORBmatcher.h:55
  (void)matcher.WindowSearch(mLastFrame, mCurrentFrame, 100, vpMapPointMatches, 0, INT_MAX);      // :73
  (void)matcher.SearchByProjection(mLastFrame, mCurrentFrame, 15, vpMapPointMatches);             // :75
  std::vector<cv::Point2f> vbPrevMatched;
  std::vector<int> vnMatches12;
  (void)matcher.SearchForInitialization(mLastFrame, mCurrentFrame, vbPrevMatched, vnMatches12, 100);   // :78
}

void mapping_and_loop_call_sites(KeyFrame* mpCurrentKeyFrame, KeyFrame* pKF2, std::vector<MapPoint*>& vpMapPointMatches, const cv::Mat& F12, const cv::Mat& Scw,
                                 const cv::Mat& R12, const cv::Mat& t12) {
  ORBmatcher matcher(0.6, false);
  std::vector<cv::KeyPoint> vMatchedKeysUn1, vMatchedKeysUn2;
  std::vector<std::pair<size_t, size_t> > vMatchedIndices;
  (void)matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, vMatchedKeysUn1, vMatchedKeysUn2, vMatchedIndices);   // src/LocalMapping.cc:1080
  (void)matcher.Fuse(pKF2, vpMapPointMatches);                                                                             // src/LocalMapping.cc:1236
  (void)matcher.Fuse(mpCurrentKeyFrame, vpMapPointMatches, 2.5f);                                                          // src/LocalMapping.cc:1261
  std::vector<MapPoint*> vpMatches12, vpMatched;
  (void)matcher.SearchByBoW(mpCurrentKeyFrame, pKF2, vpMatches12);                                                          // src/LoopClosing.cc (ComputeSim3)
  const float s12 = 1.f;
  (void)matcher.SearchBySim3(mpCurrentKeyFrame, pKF2, vpMatches12, s12, R12, t12, 7.5f);
  (void)matcher.SearchByProjection(mpCurrentKeyFrame, Scw, vpMapPointMatches, vpMatched, 10);
  // the batched forms of the two LocalMapping loops (src/LocalMapping.cc:1058-1080, :1228-1236)
  std::vector<KeyFrame*> vpNeighKFs(20, pKF2);
  std::vector<cv::Mat> vF12(20, F12);
  (void)matcher.SearchForTriangulationBegin(mpCurrentKeyFrame, vpNeighKFs, vF12);
  (void)matcher.SearchForTriangulationNext(mpCurrentKeyFrame, pKF2, 0, vMatchedKeysUn1, vMatchedKeysUn2, vMatchedIndices);
  (void)matcher.FuseTargets(vpNeighKFs, vpMapPointMatches);
  (void)matcher.Fuse(pKF2, Scw, vpMapPointMatches, 4);
}
