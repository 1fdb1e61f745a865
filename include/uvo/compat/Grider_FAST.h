/*
 * Grider_FAST-shaped adaptor over the uvo C ABI: same static entry point as include/Grider_FAST.h:81-83
 *   static void perform_griding(const cv::Mat& img, std::vector<cv::KeyPoint>& pts, int num_features,
 *                               int grid_x, int grid_y, int threshold, bool nonmaxSuppression)
 * (needs -DUVO_COMPAT_WITH_OPENCV), plus the plain-pointer form used by this repo's tests.  Points are APPENDED to
 * `pts`, like the reference (:131-134).  Ties in response are ordered (y, x); the reference's std::sort leaves them
 * unspecified.  The extractor handle that owns the device scratch is passed in, so no hidden global state exists.
 */
#ifndef UVO_COMPAT_GRIDER_FAST_H_
#define UVO_COMPAT_GRIDER_FAST_H_

#include <vector>

#include "../uvo.h"

#ifdef UVO_COMPAT_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <cstring>
#endif

namespace USLAM {

class Grider_FAST {
 public:
  static int perform_griding(uvo_extractor* scratch, const uint8_t* img, int width, int height, ptrdiff_t stride, std::vector<uvo_keypoint>& pts,
                             int num_features, int grid_x, int grid_y, int threshold, bool nonmaxSuppression) {
    std::vector<uvo_keypoint> out((size_t)num_features + (size_t)grid_x * grid_y + 64);
    int n = 0;
    const int rc = uvo_grider_fast(scratch, img, width, height, stride, num_features, grid_x, grid_y, threshold, nonmaxSuppression ? 1 : 0,
                                   out.data(), (int)out.size(), &n);
    if (rc != UVO_OK) return rc;
    pts.insert(pts.end(), out.begin(), out.begin() + n);
    return UVO_OK;
  }
#ifdef UVO_COMPAT_WITH_OPENCV
  static void perform_griding(uvo_extractor* scratch, const cv::Mat& img, std::vector<cv::KeyPoint>& pts, int num_features, int grid_x, int grid_y,
                              int threshold, bool nonmaxSuppression) {
    std::vector<uvo_keypoint> p;
    if (perform_griding(scratch, img.data, img.cols, img.rows, (ptrdiff_t)img.step, p, num_features, grid_x, grid_y, threshold,
                        nonmaxSuppression) != UVO_OK)
      return;
    const size_t n0 = pts.size();
    pts.resize(n0 + p.size());
    if (!p.empty()) std::memcpy(&pts[n0], p.data(), p.size() * sizeof(uvo_keypoint));
  }
#endif
};

}  // namespace USLAM
#endif
