/*
 * DECLARATION-ONLY stand-in for <opencv2/core/core.hpp>, test infrastructure of tests/test_cpp_compat.py.
 *
 * It exists for exactly one purpose: to let g++ -fsyntax-only TYPE-CHECK the UVO_COMPAT_WITH_OPENCV branches of
 * the headers under include/uvo/compat/ (the cv::-signature adaptors INTEGRATION.md promises) in an image that has no OpenCV.  It declares
 * the handful of names those branches touch, with the shapes OpenCV 3.x gives them, and has NO behaviour: nothing here is ever
 * linked or executed, nothing is compared with it, and it pins nothing about OpenCV's semantics (the oracle's parity stays
 * unpinned).  It is not used to build any reference source.
 */
#ifndef UVO_TEST_OPENCV_DECL_STUB_CORE_HPP_
#define UVO_TEST_OPENCV_DECL_STUB_CORE_HPP_
#include <cstddef>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_Assert(expr) ((void)(expr))

namespace cv {
typedef unsigned char uchar;

template <class T>
struct Point_ {
  T x, y;
};
typedef Point_<float> Point2f;

struct KeyPoint {  /* 28 bytes, the layout uvo_keypoint mirrors */
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};

class Mat {
 public:
  Mat();
  Mat(int rows, int cols, int type);
  int rows, cols;
  uchar* data;
  struct Step {
    operator size_t() const;
  } step;
  int type() const;
  bool empty() const;
  uchar* ptr(int row = 0);
  const uchar* ptr(int row = 0) const;
  template <class T>
  T* ptr(int row = 0);
  template <class T>
  const T* ptr(int row = 0) const;
  template <class T>
  T& at(int row);
  template <class T>
  const T& at(int row) const;
  template <class T>
  T& at(int row, int col);
  template <class T>
  const T& at(int row, int col) const;
  Mat row(int y) const;
  Mat clone() const;
};

class _InputArray {
 public:
  _InputArray(const Mat& m);
  bool empty() const;
  Mat getMat(int idx = -1) const;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m);
  void release() const;
  void create(int rows, int cols, int type) const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
#endif
