// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include,
// link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
//
// CPU restatement of the ORB front-end of chintha/U-VIP-SLAM
// (src/ORBextractor.cc, src/ORBmatcher.cc, src/FrameKTL.cc grid part,
// include/Grider_FAST.h, include/utils.h:81-111).
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
// this path, and its pixel arithmetic lives in OpenCV (3.4.x, un-vendored,
// absent from this image), so the OpenCV primitives below (copyMakeBorder,
// resize INTER_LINEAR, FAST, GaussianBlur, fastAtan2, cvRound) are restated from
// the published generic (non-IPP, non-OpenCL) C++ algorithms of OpenCV 3.4.x.
// The glue (pyramid sizing, cells, quad-tree, filter, steering, assembly,
// matching rules) follows the reference sources line by line; each function
// cites the file:line it restates.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace orc {

// cv::KeyPoint layout (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id
struct KeyPoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

struct View {  // 8-bit single channel image view (cv::Mat ROI stand-in)
  uint8_t* p;
  int w, h;
  ptrdiff_t step;
  uint8_t* row(int y) const { return p + (ptrdiff_t)y * step; }
  View roi(int x0, int y0, int x1, int y1) const { return View{p + (ptrdiff_t)y0 * step + x0, x1 - x0, y1 - y0, step}; }
};

constexpr int EDGE_THRESHOLD = 16;   // src/ORBextractor.cc:78
constexpr int HALF_PATCH_SIZE = 15;  // :77
constexpr int PATCH_SIZE = 31;       // :76

int cv_round_f(float v);   // cvRound(float): round half to even
int cv_round_d(double v);  // cvRound(double)

// ---- OpenCV primitives (restated, [OCV-RECALL]) ----
// cv::CLAHE::apply for CV_8UC1 (OpenCV 3.4 imgproc/src/clahe.cpp: CLAHE_Impl::apply, CLAHE_CalcLut_Body, CLAHE_Interpolation_Body)
// [OCV-RECALL]; call site src/Tracking.cc:425-431 (clip limit 4, 12 x 12 tiles).  dst may alias src.
void clahe_apply(const View& src, double clipLimit, int tilesX, int tilesY, uint8_t* dst, ptrdiff_t dstep);
void copy_make_border_reflect101(const View& src, uint8_t* dst, ptrdiff_t dstep, int top, int bottom, int left, int right);
void resize_linear_u8(const View& src, const View& dst);
void fast9_16(const View& img, int threshold, bool nms, std::vector<KeyPoint>& out);
void fast9_16_bruteforce(const View& img, int threshold, bool nms, std::vector<KeyPoint>& out);  // no early rejection
void gaussian_taps_7_sigma2(int taps[7]);
// blur the ROI `roi` (which must sit >= 3 px inside its parent buffer) in place,
// border taps read the parent's pixels (non-isolated sub-matrix semantics)
enum { kBlurRoundScalar = 0, kBlurRoundSse2 = 1 };
void gaussian_blur7_roi_inplace(const View& roi, int rounding = kBlurRoundSse2);
float fast_atan2(float y, float x);

// ---- reference glue ----
struct Extractor {
  // ctor: src/ORBextractor.cc:458-512
  Extractor(int nfeatures, float scaleFactor, int nlevels, int fastTh);
  int nfeatures;
  double scaleFactor;  // member is `double` in include/ORBextractor.h:79
  int nlevels, fastTh;
  int blur_rounding = kBlurRoundSse2;  // which contract the column pass of GaussianBlur rounds exact ties under (orb_oracle.cpp)
  std::vector<float> mvScaleFactor, mvInvScaleFactor;
  std::vector<int> mnFeaturesPerLevel;
  std::vector<int> umax;
  int pattern[1024];  // 512 (x,y) points

  // pyramid storage: padded planes, ROI at (16,16)
  std::vector<std::vector<uint8_t>> planes;
  std::vector<View> pyr;  // ROI views (mvImagePyramid)
  // debugging / test taps
  std::vector<std::vector<uint8_t>> planes_unblurred;  // snapshot taken before the blur step
  std::vector<std::vector<KeyPoint>> dbg_candidates;   // vToDistributeKeys per level (coords relative to minBorder)
  std::vector<std::vector<KeyPoint>> dbg_level_kps;    // after DistributeOctTree + orientation, level coords

  void ComputePyramid(const View& image);                                       // :963-1004
  void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& all);         // :748-836
  std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& v, int minX, int maxX, int minY, int maxY, int N);  // :1006-1230
  // operator(): :849-961.  grid2d is column-major (Eigen::MatrixXi::data()), rows x cols.
  void extract(const View& image, std::vector<KeyPoint>& keypoints, std::vector<uint8_t>& descriptors,
               int32_t* grid2d, int grid_rows, int grid_cols, int min_px_dist, bool fullDetect, int num_featsneeded);
};

float ic_angle(const View& image, float ptx, float pty, const std::vector<int>& umax);               // :125-152
void compute_orb_descriptor(const KeyPoint& kpt, const View& img, const int* pattern, uint8_t* desc);  // :156-195

// include/Grider_FAST.h:81-137 with the declared tie-break (response desc, then y, then x)
void grider_fast(const View& img, std::vector<KeyPoint>& pts, int num_features, int grid_x, int grid_y, int threshold, bool nms);

// ---- matcher (src/ORBmatcher.cc) ----
int descriptor_distance(const uint8_t* a, const uint8_t* b);  // :1794-1810
void knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* mask, int32_t* idx0, int32_t* d0, int32_t* idx1, int32_t* d1);

int distinctive_descriptor(const uint8_t* desc, int N, int* best_median);  // src/MapPoint.cc:197-270

struct FrameGrid {  // src/FrameKTL.cc:83-84,250-264,359-436
  int minX, minY, maxX, maxY;
  float invW, invH;
  std::vector<std::vector<int>> cells;  // [ix*48+iy]
  const KeyPoint* kps;
  int n;
  void build(const KeyPoint* kps, int n, int minX, int minY, int maxX, int maxY);
  std::vector<int> GetFeaturesInArea(float x, float y, float r, int minLevel, int maxLevel) const;
};

// SearchByProjection(FrameKTL&, vector<MapPoint*>&, th): src/ORBmatcher.cc:49-125.
// assigned[i] = index of the map point assigned to frame keypoint i, or -1 (in/out).
int search_by_projection(const FrameGrid& g, const uint8_t* fdesc, int32_t* assigned,
                         int nmp, const float* projx, const float* projy, const int32_t* level, const float* viewcos,
                         const uint8_t* inview, const uint8_t* mpdesc, const float* scaleFactors, float th, float nnratio);


// ---- the other search loops, restated on flat inputs (one array per member the reference reads) ----
// ComputeThreeMaxima: src/ORBmatcher.cc:1748-1789 (sizes of the 30 rotation bins in, three bin indices out)
void compute_three_maxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3);

// SearchByProjection(FrameKTL& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist): :1622-1746, from the point where
// u, v and nPredictedLevel are known (:1672); valid[i] = the map point reaches that line.  assigned in/out as above.
int search_by_projection_kf(const FrameGrid& g, const uint8_t* fdesc, int32_t* assigned, int nmp, const float* u, const float* v,
                            const int32_t* level, const uint8_t* valid, const uint8_t* mpdesc, const float* kf_angle,
                            const float* scaleFactors, float th, int ORBdist, bool checkOri);

struct FeatureVector {  // DBoW2::FeatureVector flattened: node ids ascending, features of node j = feat[start[j]..start[j+1])
  const uint32_t* node;
  const int32_t* start;
  const int32_t* feat;
  int n_nodes;
};
// SearchByBoW(KeyFrame*, FrameKTL&, ...) :155-284 (kf_kf false) / SearchByBoW(KeyFrame*, KeyFrame*, ...) :715-850 (true)
int search_by_bow(bool kf_kf, const FeatureVector& fv1, int n1, const uint8_t* desc1, const float* angle1, const uint8_t* usable1,
                  const FeatureVector& fv2, int n2, const uint8_t* desc2, const float* angle2, const uint8_t* usable2, float nnratio,
                  bool checkOri, int32_t* match12);
// SearchForTriangulation :852-1014 with CheckDistEpipolarLine :136-153
int search_for_triangulation(const FeatureVector& fv1, const KeyPoint* kp1, int n1, const uint8_t* desc1, const uint8_t* has_mp1,
                             const FeatureVector& fv2, const KeyPoint* kp2, int n2, const uint8_t* desc2, const uint8_t* has_mp2,
                             const float* F12, const float* sigma2, bool checkOri, int32_t* match12);
// search core of Fuse(pKF, vpMapPoints, th) :1077-1101 (KeyFrame::GetFeaturesInArea src/KeyFrame.cc:952-992)
void fuse_search(const FrameGrid& g, const uint8_t* kfdesc, int nmp, const float* u, const float* v, const int32_t* level, const uint8_t* valid,
                 const uint8_t* mpdesc, const float* scaleFactors, float th, int32_t* best_idx, int32_t* best_dist);


// ---- projection prologues.  OpenCV semantics assumed (unpinned, [OCV-RECALL]): `R*P+t` on 3x3 / 3x1 CV_32F takes cv::gemm's
// small-matrix path (row sum in float, `(float)(t0*alpha + c*beta)` with double alpha = beta = 1); `-R.t()*t` takes the general
// path (double accumulator, alpha = -1); cv::norm and cv::Mat::dot accumulate in double.
struct Camera {
  float Rcw[9], tcw[3], Ow[3];
  float fx, fy, cx, cy, minX, maxX, minY, maxY;
};
// FrameKTL::isInFrustum src/FrameKTL.cc:299-357 + MapPoint::PredictScale src/MapPoint.cc:373-388 (logScale = logf(scaleFactor))
bool is_in_frustum(const Camera& F, const float* P, const float* Pn, float mfMinDistance, float mfMaxDistance, float viewingCosLimit,
                   float scaleFactor, int nScaleLevels, float* u, float* v, int* level, float* viewCos);
// SearchByProjection(CurrentFrame, pKF, ...) src/ORBmatcher.cc:1626-1670
bool project_kf_reloc(const Camera& F, const float* x3Dw, float mfMinDistance, const float* scaleFactors, int nScaleLevels, float* u, float* v,
                      int* level);
// the four members without a caller in the reference (WindowSearch :409-516, SearchByProjection(F1, F2, windowSize) :519-594,
// SearchForInitialization :598-713, SearchByProjection(CurrentFrame, LastFrame, th) :1507-1620)
int window_search(const KeyPoint* kp1, int n1, const uint8_t* desc1, const uint8_t* has_mp1, const FrameGrid& g2, const uint8_t* desc2, int n2,
                  int windowSize, int minScaleLevel, int maxScaleLevel, float nnratio, bool checkOri, int32_t* match21);
int search_by_projection_frames(const KeyPoint* kp1, int n1, const uint8_t* desc1, const uint8_t* usable1, const float* xyz1, const Camera& F2,
                                const FrameGrid& g2, const uint8_t* desc2, int32_t* assigned2, int windowSize, float nnratio);
int search_for_initialization(const KeyPoint* kp1, int n1, const uint8_t* desc1, const FrameGrid& g2, const uint8_t* desc2, int n2,
                              float* prev_matched, int32_t* vnMatches12, int windowSize, float nnratio, bool checkOri);
int search_by_projection_last(const Camera& Cur, const FrameGrid& g, const uint8_t* fdesc, int32_t* assigned, int nlast, const uint8_t* usable_last,
                              const float* xyz_last, const int32_t* octave_last, const float* angle_last, const uint8_t* desc_last,
                              const float* scaleFactors, float th, bool checkOri);
// Fuse src/ORBmatcher.cc:1037-1075
bool project_fuse(const Camera& K, const float* p3Dw, const float* Pn, float mfMinDistance, float mfMaxDistance, const float* scaleFactors,
                  int nScaleLevels, float* u, float* v, int* level);
// Sim3 forms: src/ORBmatcher.cc:299-303 (= :1145-1149), :1284-1287, :1323-1359 (= :1403-1441), :357-398, :1361-1504
void sim3_decompose(const float* Scw, int row_stride, float* Rcw, float* tcw, float* Ow);
void sim3_relative(float s12, const float* R12, const float* t12, float* sR12, float* sR21, float* t21);
bool project_sim3(const float* Ra, const float* ta, const float* sR, const float* t, const Camera& K, const float* p3Dw, float minDistance,
                  float maxDistance, const float* scaleFactors, int nScaleLevels, float* u, float* v, int* level);
int search_by_projection_sim3(const FrameGrid& g, const uint8_t* kfdesc, int32_t* matched, int nmp, const float* u, const float* v,
                              const int32_t* level, const uint8_t* valid, const uint8_t* mpdesc, const float* scaleFactors, int th);
int search_by_sim3(const FrameGrid& g1, const uint8_t* desc1, int n1, const FrameGrid& g2, const uint8_t* desc2, int n2, const float* u12,
                   const float* v12, const int32_t* level12, const uint8_t* valid12, const uint8_t* mpdesc1, const float* u21, const float* v21,
                   const int32_t* level21, const uint8_t* valid21, const uint8_t* mpdesc2, const float* scaleFactors1,
                   const float* scaleFactors2, float th, int32_t* match12);


// ---- DBoW2 bag-of-words transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1125-1258, BowVector.cpp, FeatureVector.cpp) ----
struct Vocabulary {  // m_nodes flattened: children of node i = children[child_start[i] .. child_start[i+1])
  int n_nodes;
  const int32_t* child_start;
  const int32_t* children;
  const uint8_t* descriptor;
  const int32_t* word_id;
  const double* weight;
  int L, weighting /* 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY */, normalize /* 0 none, 1 L1, 2 L2 */;
};
// transform(feature, word_id, weight, nid, levelsup) :1207-1258.  nid is left unset by the reference when the descent ends above
// level L - levelsup; declared here: the leaf.
void bow_transform_one(const Vocabulary& voc, const uint8_t* feature, int levelsup, int* word_id, double* weight, int* nid);
// transform(features, v, fv, levelsup) :1125-1188; outputs flattened in map order
void bow_transform(const Vocabulary& voc, const uint8_t* features, int n, int levelsup, std::vector<std::pair<uint32_t, double>>& bow,
                   std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& fv);

// haloc::Hash::getHash: src/hash.cpp:57-85 (r = the projection vectors, each at least n long)
void haloc_hash(const float* r, int num_proj, int r_stride, const uint8_t* desc, int n, float* hash);

// ---- KLT front-end step (klt_oracle.cpp): cv::buildOpticalFlowPyramid + cv::calcOpticalFlowPyrLK ----
struct KltPyramid {
  struct Level {
    int w, h;
    ptrdiff_t istep, dstep;        // image bytes per row / derivative shorts per row, both incl. the window border
    std::vector<uint8_t> img;      // (w + 2bx) x (h + 2by), REFLECT_101 border
    std::vector<int16_t> deriv;    // interleaved (dx, dy), zero border
  };
  int bx = 0, by = 0;
  std::vector<Level> levels;
  void build(const uint8_t* img, int w, int h, ptrdiff_t stride, int win_w, int win_h, int maxLevel);
};
void undistort_points(const float* pts, int n, float fx, float fy, float cx, float cy, const float* dist, int n_dist, bool fisheye, float* out);
void klt_track(const KltPyramid& P0, const KltPyramid& P1, const float* prevPts, float* nextPts, int npts, int win_w, int win_h, int maxLevel,
               int maxCount, double epsilon, double minEigThreshold, uint8_t* status, float* err, int sum_mode = 0, float* margin = nullptr);

}  // namespace orc
