/*
 * USLAM::ORBextractor-shaped adaptor over the uvo C ABI (include/uvo/uvo.h).
 *
 * Mirrors the public surface of the reference class (include/ORBextractor.h:47-95): same constructor arguments and
 * defaults, same operator() argument list and ownership rules (src/ORBextractor.cc:849-961), GetLevels(),
 * GetScaleFactor().  The reference host (src/Tracking.cc:145,946; src/FrameKTL.cc:233-234) compiles against this
 * header unchanged when UVO_COMPAT_WITH_OPENCV is defined (needs OpenCV core + Eigen, exactly what that host already
 * uses); without the macro only the plain-pointer form `extract()` is available, which is what this repo's own tests
 * use on machines without OpenCV.
 *
 * Error behaviour: the reference's operator() is void and has no failure path other than "empty outputs".  A failing
 * ABI call therefore leaves `keypoints` empty and releases `descriptors`, and the text is kept in last_error().
 * The GPU handle is created on the first call (the reference constructor does not know the image size) and re-created
 * only if a larger image arrives.  Not re-entrant, like the reference (scratch pyramid member, include/ORBextractor.h:90).
 */
#ifndef UVO_COMPAT_ORBEXTRACTOR_H_
#define UVO_COMPAT_ORBEXTRACTOR_H_

#include <cstring>
#include <string>
#include <vector>

#include "../uvo.h"

#ifdef UVO_COMPAT_WITH_OPENCV
#include <Eigen/Core>
#include <opencv2/core/core.hpp>
#endif

namespace USLAM {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures = 1000, float scaleFactor = 1.2f, int nlevels = 8, int scoreType = HARRIS_SCORE, int fastTh = 7)
      : nfeatures_(nfeatures), scaleFactor_(scaleFactor), nlevels_(nlevels), scoreType_(scoreType), fastTh_(fastTh) {}
  ~ORBextractor() { uvo_extractor_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  int GetLevels() { return nlevels_; }
  float GetScaleFactor() { return (float)(double)scaleFactor_; }
  const std::string& last_error() const { return err_; }
  void set_device(int device) { device_ = device; }
  void set_max_input_keypoints(int n) { max_in_ = n; }

  /* Plain-pointer form of operator(): `keypoints` is in/out exactly like the reference's vector (caller keypoints on
   * entry, result on return); grid is Eigen::MatrixXi::data() (column-major, rows x cols), mutated when !FullDetect. */
  int extract(const uint8_t* img, int width, int height, ptrdiff_t stride, std::vector<uvo_keypoint>& keypoints,
              std::vector<uint8_t>& descriptors, int32_t* grid, int grid_rows, int grid_cols, int min_px_dist, bool FullDetect,
              int num_featsneeded) {
    if (!img || width <= 0 || height <= 0) return UVO_OK;  // `if(_image.empty()) return;` (src/ORBextractor.cc:852-853)
    int rc = ensure(width, height, (int)keypoints.size());
    if (rc == UVO_OK) {
      const int cap = cap_;
      out_kp_.resize(cap);
      out_desc_.resize((size_t)cap * 32);
      int n = 0;
      rc = uvo_extract(h_, img, width, height, stride, keypoints.empty() ? nullptr : keypoints.data(), (int)keypoints.size(), grid, grid_rows,
                       grid_cols, min_px_dist, FullDetect ? 1 : 0, num_featsneeded, out_kp_.data(), out_desc_.data(), cap, &n);
      if (rc == UVO_OK) {
        keypoints.assign(out_kp_.begin(), out_kp_.begin() + n);
        descriptors.assign(out_desc_.begin(), out_desc_.begin() + (size_t)n * 32);
        return UVO_OK;
      }
    }
    err_ = uvo_last_error();
    keypoints.clear();
    descriptors.clear();
    return rc;
  }

#ifdef UVO_COMPAT_WITH_OPENCV
  /* The reference signature, verbatim (include/ORBextractor.h:56-58). */
  void operator()(cv::InputArray _image, cv::InputArray /*mask: ignored by the live path*/, std::vector<cv::KeyPoint>& _keypoints,
                  cv::OutputArray _descriptors, Eigen::MatrixXi& grid_2d, int& min_px_dist, bool FullDetect, int num_featsneeded) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    static_assert(sizeof(cv::KeyPoint) == sizeof(uvo_keypoint), "cv::KeyPoint layout");
    std::vector<uvo_keypoint> kps(_keypoints.size());
    if (!kps.empty()) std::memcpy(kps.data(), _keypoints.data(), kps.size() * sizeof(uvo_keypoint));
    std::vector<uint8_t> desc;
    extract(image.data, image.cols, image.rows, (ptrdiff_t)image.step, kps, desc, grid_2d.data(), (int)grid_2d.rows(), (int)grid_2d.cols(),
            min_px_dist, FullDetect, num_featsneeded);
    _keypoints.resize(kps.size());
    if (kps.empty()) {
      _descriptors.release();
      return;
    }
    std::memcpy(_keypoints.data(), kps.data(), kps.size() * sizeof(uvo_keypoint));
    _descriptors.create((int)kps.size(), 32, CV_8U);
    cv::Mat d = _descriptors.getMat();
    for (int i = 0; i < d.rows; ++i) std::memcpy(d.ptr(i), &desc[(size_t)i * 32], 32);
  }
#endif

 private:
  int ensure(int w, int h, int n_in) {
    if (h_ && w <= max_w_ && h <= max_h_ && n_in <= max_in_) return UVO_OK;
    uvo_extractor_destroy(h_);
    h_ = nullptr;
    max_w_ = w > max_w_ ? w : max_w_;
    max_h_ = h > max_h_ ? h : max_h_;
    if (n_in > max_in_) max_in_ = n_in * 2;
    if (max_in_ < nfeatures_ * 2) max_in_ = nfeatures_ * 2;
    uvo_extractor_cfg c;
    c.nfeatures = nfeatures_, c.scale_factor = (float)scaleFactor_, c.nlevels = nlevels_, c.score_type = scoreType_, c.fast_th = fastTh_;
    c.max_width = max_w_, c.max_height = max_h_, c.max_batch = 1, c.max_input_keypoints = max_in_, c.device = device_;
    int rc = uvo_extractor_create(&c, &h_);
    if (rc != UVO_OK) return rc;
    cap_ = uvo_extractor_max_keypoints(h_);  // can never overflow: sum over levels of max(quota, 4 * nIni) + 4, plus the pass-through points
    return cap_ < 0 ? cap_ : UVO_OK;
  }

  int nfeatures_;
  double scaleFactor_;  // `double scaleFactor;` in the reference (include/ORBextractor.h:79)
  int nlevels_, scoreType_, fastTh_;
  int device_ = 0, max_in_ = 0, max_w_ = 0, max_h_ = 0, cap_ = 0;
  uvo_extractor* h_ = nullptr;
  std::vector<uvo_keypoint> out_kp_;
  std::vector<uint8_t> out_desc_;
  std::string err_;
};

}  // namespace USLAM
#endif
