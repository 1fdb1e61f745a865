// ORACLE -- TEST INFRASTRUCTURE ONLY (see orb_oracle.hpp header).  PARITY UNPINNED.
//
// Extractor half of the oracle.  Built with plain `g++ -O3 -ffp-contract=off`
// (no -march=native), mirroring the reference's CMakeLists.txt:19-22.
#include "orb_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <list>
#include <utility>

namespace orc {

static const int8_t kPattern[1024] = {
#include "rbrief_pattern.inc"
};

// OpenCV cvRound on x86-64: cvtss2si / cvtsd2si under the default MXCSR = round half to even.
int cv_round_f(float v) { return (int)lrintf(v); }
int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}
static inline int cv_ceil_f(float v) {
  int i = (int)v;
  return i + (i < v);
}

// ---------------------------------------------------------------------------------------------
// cv::copyMakeBorder(..., BORDER_REFLECT_101)   [OCV-RECALL]  gfedcb|abcdefgh|gfedcba
// call sites: src/ORBextractor.cc:988,996
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0)
      p = -p;
    else
      p = 2 * (len - 1) - p;
  }
  return p;
}

void copy_make_border_reflect101(const View& src, uint8_t* dst, ptrdiff_t dstep, int top, int bottom, int left, int right) {
  const int dw = src.w + left + right, dh = src.h + top + bottom;
  for (int y = 0; y < dh; ++y) {
    const uint8_t* srow = src.row(reflect101(y - top, src.h));
    uint8_t* drow = dst + (ptrdiff_t)y * dstep;
    for (int x = 0; x < dw; ++x) drow[x] = srow[reflect101(x - left, src.w)];
  }
}

// ---------------------------------------------------------------------------------------------
// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1, generic C++ path of OpenCV 3.4.x
// (resizeGeneric_ + HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<..,22>>)
// [OCV-RECALL].  call site: src/ORBextractor.cc:982
void resize_linear_u8(const View& src, const View& dst) {
  const int sw = src.w, sh = src.h, dw = dst.w, dh = dst.h;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  const int COEF = 2048;  // INTER_RESIZE_COEF_SCALE
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * (size_t)dw), ibeta(2 * (size_t)dh);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx >= sw - 1) {
      fx = 0;
      sx = sw - 1;
    }
    xofs[dx] = sx;
    float c0 = 1.f - fx, c1 = fx;
    ialpha[2 * dx] = (short)cv_round_f(c0 * COEF);
    ialpha[2 * dx + 1] = (short)cv_round_f(c1 * COEF);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    yofs[dy] = sy;
    float c0 = 1.f - fy, c1 = fy;
    ibeta[2 * dy] = (short)cv_round_f(c0 * COEF);
    ibeta[2 * dy + 1] = (short)cv_round_f(c1 * COEF);
  }
  std::vector<int> r0(dw), r1(dw);
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  auto hresize = [&](const uint8_t* S, std::vector<int>& D) {
    for (int dx = 0; dx < dw; ++dx) {
      int sx = xofs[dx];
      int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
      if (sx >= sw - 1)
        D[dx] = S[sx] * COEF;  // dx >= xmax branch of HResizeLinear
      else
        D[dx] = S[sx] * a0 + S[sx + 1] * a1;
    }
  };
  for (int dy = 0; dy < dh; ++dy) {
    int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hresize(src.row(sy0), r0);
    hresize(src.row(sy1), r1);
    int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t* D = dst.row(dy);
    for (int x = 0; x < dw; ++x) D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---------------------------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, nonmaxSuppression) type 9_16, generic path FAST_t<16> [OCV-RECALL].
// call sites: src/ORBextractor.cc:792,797 ; include/Grider_FAST.h:114
static const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// definitional: is (x,y) a FAST-9 corner at threshold t; *score = largest t' for which it still is
static bool fast_corner(const View& img, int x, int y, int t, int* score) {
  const int v = img.row(y)[x];
  int d[16];
  for (int k = 0; k < 16; ++k) d[k] = (int)img.row(y + kCircle[k][1])[x + kCircle[k][0]] - v;
  bool corner = false;
  int best = 0;
  for (int s = 0; s < 16; ++s) {
    int mb = 1 << 30, md = 1 << 30;
    for (int k = 0; k < 9; ++k) {
      int dd = d[(s + k) & 15];
      mb = std::min(mb, dd);   // brighter arc: all p - v > t
      md = std::min(md, -dd);  // darker arc : all v - p > t
    }
    if (mb > t || md > t) corner = true;
    best = std::max(best, std::max(mb, md));
  }
  // cornerScore<16>: max over arcs of min|diff| minus 1 (== largest threshold that keeps it a corner)
  *score = best - 1;
  return corner;
}

// `quick` enables FAST_t's cheap necessary conditions (any 9-arc contains one pixel of every opposite pair
// (k, k+8)) before the definitional test; results are identical with and without it (tests check that).
static void fast9_16_impl(const View& img, int threshold, bool nms, std::vector<KeyPoint>& out, bool quick) {
  out.clear();
  const int w = img.w, h = img.h;
  if (w < 7 || h < 7) return;
  threshold = std::min(std::max(threshold, 0), 255);
  std::vector<int> score((size_t)w * h, 0);  // 0 = not a corner
  std::vector<uint8_t> is((size_t)w * h, 0);
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      if (quick) {
        const int v = img.row(y)[x];
        auto cls = [&](int k) {
          const int p = img.row(y + kCircle[k][1])[x + kCircle[k][0]];
          return p < v - threshold ? 1 : (p > v + threshold ? 2 : 0);
        };
        int d = cls(0) | cls(8);
        if (!d) continue;
        d &= cls(2) | cls(10);
        d &= cls(4) | cls(12);
        d &= cls(6) | cls(14);
        if (!d) continue;
        d &= cls(1) | cls(9);
        d &= cls(3) | cls(11);
        d &= cls(5) | cls(13);
        d &= cls(7) | cls(15);
        if (!d) continue;
      }
      int s;
      if (fast_corner(img, x, y, threshold, &s)) {
        is[(size_t)y * w + x] = 1;
        score[(size_t)y * w + x] = s;
      }
    }
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      size_t i = (size_t)y * w + x;
      if (!is[i]) continue;
      int s = score[i];
      if (nms) {
        bool keep = true;
        for (int dy = -1; dy <= 1 && keep; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            if (!dx && !dy) continue;
            if (!(s > score[(size_t)(y + dy) * w + (x + dx)])) {
              keep = false;
              break;
            }
          }
        if (!keep) continue;
      }
      out.push_back(KeyPoint{(float)x, (float)y, 7.f, -1.f, (float)s, 0, -1});
    }
}

void fast9_16(const View& img, int threshold, bool nms, std::vector<KeyPoint>& out) { fast9_16_impl(img, threshold, nms, out, true); }
void fast9_16_bruteforce(const View& img, int threshold, bool nms, std::vector<KeyPoint>& out) {
  fast9_16_impl(img, threshold, nms, out, false);
}

// ---------------------------------------------------------------------------------------------
// cv::GaussianBlur(m, m, Size(7,7), 2, 2, BORDER_REFLECT_101) on a CV_8U *sub-matrix* without
// BORDER_ISOLATED: the bit-exact ufixedpoint16 branch is skipped and sepFilter2D runs with the
// symmetric-smooth integer engine (taps * 256 per pass, (sum + 2^15) >> 16)  [OCV-RECALL].
// call site: src/ORBextractor.cc:942
void gaussian_taps_7_sigma2(int taps[7]) {
  // cv::getGaussianKernel(7, 2.0, CV_32F)
  const int n = 7;
  const double sigma = 2.0;
  const double scale2X = -0.5 / (sigma * sigma);
  float cf[7];
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    double t = std::exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) cf[i] = (float)(cf[i] * sum);
  // createSeparableLinearFilter: kernel.convertTo(CV_32S, 1 << 8)
  for (int i = 0; i < n; ++i) taps[i] = cv_round_f(cf[i] * 256.f);
}

// `rounding`: how the column pass rounds an exact .5 (every other sum rounds the same either way).
//   kBlurRoundScalar: (s + 2^15) >> 16 on every column -- FixedPtCastEx<int, uchar>(16), the generic C++ column filter;
//   kBlurRoundSse2 (default): what an x86-64 OpenCV 3.4.x build executes [OCV-RECALL]: SymmColumnVec_32s8u's vector body takes the columns
//     0 .. (w & ~3) - 1 (16 and then 4 at a time), computes the sum in fp32 -- exact here: every partial sum is a multiple of 2^-16 below
//     2^8 -- and converts with cvtps2dq, round-half-to-EVEN; only the last w % 4 columns fall to the scalar loop above.
void gaussian_blur7_roi_inplace(const View& roi, int rounding) {
  int k[7];
  gaussian_taps_7_sigma2(k);
  const int w = roi.w, h = roi.h;
  std::vector<int> rows((size_t)(h + 6) * w);
  for (int y = -3; y < h + 3; ++y) {
    const uint8_t* S = roi.p + (ptrdiff_t)y * roi.step;  // parent pixels above/below the ROI
    int* R = &rows[(size_t)(y + 3) * w];
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = 0; i < 7; ++i) s += k[i] * S[x + i - 3];  // parent pixels left/right of the ROI
      R[x] = s;
    }
  }
  for (int y = 0; y < h; ++y) {
    uint8_t* D = roi.row(y);
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int j = 0; j < 7; ++j) s += k[j] * rows[(size_t)(y + j) * w + x];
      int v = (s + (1 << 15)) >> 16;  // FixedPtCastEx<int,uchar>(16)
      if (rounding == kBlurRoundSse2 && x < (w & ~3) && (s & 0xffff) == 0x8000) v &= ~1;  // an exact tie: to even
      D[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// cv::fastAtan2(y, x) -> degrees, OpenCV 3.4.x atan_f32 [OCV-RECALL].  call site: src/ORBextractor.cc:151
float fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---------------------------------------------------------------------------------------------
// IC_Angle: src/ORBextractor.cc:125-152
float ic_angle(const View& image, float ptx, float pty, const std::vector<int>& u_max) {
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = image.row(cv_round_f(pty)) + cv_round_f(ptx);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  const int step = (int)image.step;
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2((float)m_01, (float)m_10);
}

// computeOrbDescriptor: src/ORBextractor.cc:156-195.  `cos`/`sin` resolve to the float
// overloads (libm cosf/sinf) because `angle` is float and the file is `using namespace std`.
void compute_orb_descriptor(const KeyPoint& kpt, const View& img, const int* pattern, uint8_t* desc) {
  const float factorPI = (float)(M_PI / 180.f);
  float angle = (float)kpt.angle * factorPI;
  float a = (float)cosf(angle), b = (float)sinf(angle);
  const uint8_t* center = img.row(cv_round_f(kpt.y)) + cv_round_f(kpt.x);
  const int step = (int)img.step;
  auto get = [&](int idx) -> int {
    const float px = (float)pattern[2 * idx], py = (float)pattern[2 * idx + 1];
    return center[cv_round_f(px * b + py * a) * step + cv_round_f(px * a - py * b)];
  };
  for (int i = 0; i < 32; ++i, pattern += 32) {
    int val = 0;
    for (int j = 0; j < 8; ++j) {
      int t0 = get(2 * j), t1 = get(2 * j + 1);
      val |= (t0 < t1) << j;
    }
    desc[i] = (uint8_t)val;
  }
}

// ---------------------------------------------------------------------------------------------
// ORBextractor::ORBextractor: src/ORBextractor.cc:458-512
Extractor::Extractor(int _nfeatures, float _scaleFactor, int _nlevels, int _fastTh)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), fastTh(_fastTh) {
  mvScaleFactor.resize(nlevels);
  mvScaleFactor[0] = 1;
  for (int i = 1; i < nlevels; i++) mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
  float invScaleFactor = (float)(1.0f / scaleFactor);
  mvInvScaleFactor.resize(nlevels);
  mvInvScaleFactor[0] = 1;
  for (int i = 1; i < nlevels; i++) mvInvScaleFactor[i] = mvInvScaleFactor[i - 1] * invScaleFactor;

  mnFeaturesPerLevel.resize(nlevels);
  float factor = (float)(1.0 / scaleFactor);
  float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sumFeatures = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    mnFeaturesPerLevel[level] = cv_round_f(nDesiredFeaturesPerScale);
    sumFeatures += mnFeaturesPerLevel[level];
    nDesiredFeaturesPerScale *= factor;
  }
  mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

  for (int i = 0; i < 1024; ++i) pattern[i] = kPattern[i];

  umax.resize(HALF_PATCH_SIZE + 1);
  int v, v0, vmax = cv_floor_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
  int vmin = cv_ceil_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

// ORBextractor::ComputePyramid: src/ORBextractor.cc:963-1004 (mask branch never taken)
void Extractor::ComputePyramid(const View& image) {
  planes.assign(nlevels, {});
  pyr.assign(nlevels, View{});
  for (int level = 0; level < nlevels; ++level) {
    float scale = mvInvScaleFactor[level];
    const int sw = cv_round_f((float)image.w * scale), sh = cv_round_f((float)image.h * scale);
    const int ww = sw + EDGE_THRESHOLD * 2, wh = sh + EDGE_THRESHOLD * 2;
    planes[level].assign((size_t)ww * wh, 0);
    View temp{planes[level].data(), ww, wh, ww};
    pyr[level] = temp.roi(EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD + sw, EDGE_THRESHOLD + sh);
    if (level != 0) {
      resize_linear_u8(pyr[level - 1], pyr[level]);
      // copyMakeBorder(ROI -> its own parent, REFLECT_101 + ISOLATED): pad from the ROI's own pixels
      std::vector<uint8_t> tmp((size_t)sw * sh);
      for (int y = 0; y < sh; ++y) memcpy(&tmp[(size_t)y * sw], pyr[level].row(y), sw);
      copy_make_border_reflect101(View{tmp.data(), sw, sh, sw}, temp.p, temp.step, EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD,
                                  EDGE_THRESHOLD);
    } else {
      copy_make_border_reflect101(image, temp.p, temp.step, EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ExtractorNode + DivideNode: include/ORBextractor.h:32-45, src/ORBextractor.cc:1231-1287
namespace {
struct Node {
  std::vector<KeyPoint> vKeys;
  int ULx = 0, ULy = 0, URx = 0, URy = 0, BLx = 0, BLy = 0, BRx = 0, BRy = 0;
  std::list<Node>::iterator lit;
  bool bNoMore = false;
  long seq = 0;  // creation sequence: stands in for the heap address in the pair<int,ExtractorNode*> sort
  void DivideNode(Node& n1, Node& n2, Node& n3, Node& n4) const {
    const int halfX = (int)ceilf((float)(URx - ULx) / 2);
    const int halfY = (int)ceilf((float)(BRy - ULy) / 2);
    n1.ULx = ULx, n1.ULy = ULy;
    n1.URx = ULx + halfX, n1.URy = ULy;
    n1.BLx = ULx, n1.BLy = ULy + halfY;
    n1.BRx = ULx + halfX, n1.BRy = ULy + halfY;
    n2.ULx = n1.URx, n2.ULy = n1.URy;
    n2.URx = URx, n2.URy = URy;
    n2.BLx = n1.BRx, n2.BLy = n1.BRy;
    n2.BRx = URx, n2.BRy = ULy + halfY;
    n3.ULx = n1.BLx, n3.ULy = n1.BLy;
    n3.URx = n1.BRx, n3.URy = n1.BRy;
    n3.BLx = BLx, n3.BLy = BLy;
    n3.BRx = n1.BRx, n3.BRy = BLy;
    n4.ULx = n3.URx, n4.ULy = n3.URy;
    n4.URx = n2.BRx, n4.URy = n2.BRy;
    n4.BLx = n3.BRx, n4.BLy = n3.BRy;
    n4.BRx = BRx, n4.BRy = BRy;
    for (size_t i = 0; i < vKeys.size(); i++) {
      const KeyPoint& kp = vKeys[i];
      if (kp.x < n1.URx) {
        if (kp.y < n1.BRy)
          n1.vKeys.push_back(kp);
        else
          n3.vKeys.push_back(kp);
      } else if (kp.y < n1.BRy)
        n2.vKeys.push_back(kp);
      else
        n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
  }
};
}  // namespace

// ORBextractor::DistributeOctTree: src/ORBextractor.cc:1006-1230.
// Declared tie-break for the address-dependent std::sort of pair<int,ExtractorNode*> (:1151):
// among equal sizes the most recently created node is split first (= what strictly increasing heap
// addresses would give).  SURVEY.md Appendix C.
std::vector<KeyPoint> Extractor::DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX, int maxX, int minY, int maxY,
                                                   int N) {
  const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
  const float hX = (float)(maxX - minX) / nIni;
  std::list<Node> lNodes;
  std::vector<Node*> vpIniNodes(nIni);
  long seq = 0;
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.ULx = (int)(hX * (float)i), ni.ULy = 0;
    ni.URx = (int)(hX * (float)(i + 1)), ni.URy = 0;
    ni.BLx = ni.ULx, ni.BLy = maxY - minY;
    ni.BRx = ni.URx, ni.BRy = maxY - minY;
    ni.seq = seq++;
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
    const KeyPoint& kp = vToDistributeKeys[i];
    vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
  }
  auto lit = lNodes.begin();
  while (lit != lNodes.end()) {
    if (lit->vKeys.size() == 1) {
      lit->bNoMore = true;
      lit++;
    } else if (lit->vKeys.empty())
      lit = lNodes.erase(lit);
    else
      lit++;
  }

  bool bFinish = false;
  typedef std::pair<int, std::pair<long, Node*>> SizeSeqNode;  // (size, (seq, node)): seq replaces the pointer value
  std::vector<SizeSeqNode> vSizeAndPointerToNode;

  auto push_children = [&](Node& n, int* nToExpand) {
    if (n.vKeys.size() > 0) {
      n.seq = seq++;
      lNodes.push_front(n);
      if (n.vKeys.size() > 1) {
        if (nToExpand) (*nToExpand)++;
        vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), std::make_pair(lNodes.front().seq, &lNodes.front())));
        lNodes.front().lit = lNodes.begin();
      }
    }
  };

  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) {
        lit++;
        continue;
      } else {
        Node n1, n2, n3, n4;
        lit->DivideNode(n1, n2, n3, n4);
        push_children(n1, &nToExpand);
        push_children(n2, &nToExpand);
        push_children(n3, &nToExpand);
        push_children(n4, &nToExpand);
        lit = lNodes.erase(lit);
        continue;
      }
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<SizeSeqNode> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end());
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          Node n1, n2, n3, n4;
          Node* parent = vPrev[j].second.second;
          parent->DivideNode(n1, n2, n3, n4);
          push_children(n1, nullptr);
          push_children(n2, nullptr);
          push_children(n3, nullptr);
          push_children(n4, nullptr);
          lNodes.erase(parent->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }

  std::vector<KeyPoint> vResultKeys;
  vResultKeys.reserve(nfeatures);
  for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
    std::vector<KeyPoint>& vNodeKeys = it->vKeys;
    KeyPoint* pKP = &vNodeKeys[0];
    float maxResponse = pKP->response;
    for (size_t k = 1; k < vNodeKeys.size(); k++) {
      if (vNodeKeys[k].response > maxResponse) {
        pKP = &vNodeKeys[k];
        maxResponse = vNodeKeys[k].response;
      }
    }
    vResultKeys.push_back(*pKP);
  }
  return vResultKeys;
}

// ORBextractor::ComputeKeyPointsOctTree: src/ORBextractor.cc:748-836
void Extractor::ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints) {
  allKeypoints.assign(nlevels, {});
  dbg_candidates.assign(nlevels, {});
  const float W = 30;
  for (int level = 0; level < nlevels; ++level) {
    const int minBorderX = EDGE_THRESHOLD - 3;
    const int minBorderY = minBorderX;
    const int maxBorderX = pyr[level].w - EDGE_THRESHOLD + 3;
    const int maxBorderY = pyr[level].h - EDGE_THRESHOLD + 3;
    std::vector<KeyPoint> vToDistributeKeys;
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W);
    const int nRows = (int)(height / W);
    const int wCell = (int)ceilf(width / nCols);
    const int hCell = (int)ceilf(height / nRows);
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBorderY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBorderY - 3) continue;
      if (maxY > maxBorderY) maxY = (float)maxBorderY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBorderX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBorderX - 6) continue;
        if (maxX > maxBorderX) maxX = (float)maxBorderX;
        std::vector<KeyPoint> vKeysCell;
        View cell = pyr[level].roi((int)iniX, (int)iniY, (int)maxX, (int)maxY);
        fast9_16(cell, fastTh, true, vKeysCell);
        if (vKeysCell.empty()) fast9_16(cell, 7, true, vKeysCell);
        for (auto& kp : vKeysCell) {
          kp.x += j * wCell;
          kp.y += i * hCell;
          vToDistributeKeys.push_back(kp);
        }
      }
    }
    dbg_candidates[level] = vToDistributeKeys;
    std::vector<KeyPoint>& keypoints = allKeypoints[level];
    keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY, mnFeaturesPerLevel[level]);
    const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
    for (auto& kp : keypoints) {
      kp.x += minBorderX;
      kp.y += minBorderY;
      kp.octave = level;
      kp.size = (float)scaledPatchSize;
    }
  }
  for (int level = 0; level < nlevels; ++level)
    for (auto& kp : allKeypoints[level]) kp.angle = ic_angle(pyr[level], kp.x, kp.y, umax);
}

// ORBextractor::operator(): src/ORBextractor.cc:849-961
void Extractor::extract(const View& image, std::vector<KeyPoint>& _keypoints, std::vector<uint8_t>& _descriptors, int32_t* grid2d,
                        int grid_rows, int grid_cols, int min_px_dist, bool FullDetect, int num_featsneeded) {
  (void)grid_cols;
  if (image.w == 0 || image.h == 0) return;
  ComputePyramid(image);
  planes_unblurred = planes;

  std::vector<std::vector<KeyPoint>> allKeypoints, allKeypointsTemp;
  // ComputeKeyPointsCopy: :523-534
  allKeypoints.assign(nlevels, {});
  allKeypoints[0] = _keypoints;
  for (int level = 0; level < nlevels; ++level)
    for (auto& kp : allKeypoints[level]) kp.angle = ic_angle(pyr[level], kp.x, kp.y, umax);
  ComputeKeyPointsOctTree(allKeypointsTemp);
  dbg_level_kps = allKeypointsTemp;
  allKeypoints.resize(allKeypointsTemp.size());
  int Total_counter = 0, KP_counter = 0;
  bool break_key = false;
  auto G = [&](int r, int c) -> int32_t& { return grid2d[(size_t)c * grid_rows + r]; };  // Eigen column-major
  if (!FullDetect) {
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KeyPoint>& keypoints = allKeypointsTemp[level];
      if (keypoints.empty()) continue;
      int numofpoint = num_featsneeded * (8 - level) / 30;
      float scale = mvScaleFactor[level];
      for (auto& kp : keypoints) {
        float tx = kp.x * scale, ty = kp.y * scale;
        int r = (int)(ty / min_px_dist), c = (int)(tx / min_px_dist);
        if (G(r, c) > 0) continue;
        allKeypoints[level].push_back(kp);
        G(r, c)++;
        KP_counter++;
        Total_counter++;
        if (KP_counter == numofpoint) {
          KP_counter = 0;
          break;
        }
        if (Total_counter == num_featsneeded) {
          break_key = true;
          break;
        }
      }
      if (break_key) break;
    }
  } else {
    allKeypoints = allKeypointsTemp;
  }

  int nkeypoints = 0;
  for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
  _descriptors.assign((size_t)nkeypoints * 32, 0);
  _keypoints.clear();
  _keypoints.reserve(nkeypoints);
  int offset = 0;
  for (int level = 0; level < nlevels; ++level) {
    std::vector<KeyPoint>& keypoints = allKeypoints[level];
    int nkeypointsLevel = (int)keypoints.size();
    if (nkeypointsLevel == 0) continue;
    gaussian_blur7_roi_inplace(pyr[level], blur_rounding);
    for (int i = 0; i < nkeypointsLevel; ++i) compute_orb_descriptor(keypoints[i], pyr[level], pattern, &_descriptors[(size_t)(offset + i) * 32]);
    offset += nkeypointsLevel;
    if (level != 0) {
      float scale = mvScaleFactor[level];
      for (auto& kp : keypoints) {
        kp.x *= scale;
        kp.y *= scale;
      }
    }
    _keypoints.insert(_keypoints.end(), keypoints.begin(), keypoints.end());
  }
}

// ---------------------------------------------------------------------------------------------
// Grider_FAST::perform_griding: include/Grider_FAST.h:81-137.  std::sort by response is unstable in the
// reference; declared tie-break here: response desc, then y asc, then x asc (ROI coords).
void grider_fast(const View& img, std::vector<KeyPoint>& pts, int num_features, int grid_x, int grid_y, int threshold, bool nms) {
  int size_x = img.w / grid_x;
  int size_y = img.h / grid_y;
  assert(size_x > 0 && size_y > 0);
  int num_features_grid = (int)(num_features / (grid_x * grid_y)) + 1;
  int ct_cols = (int)std::floor(img.w / size_x);
  int ct_rows = (int)std::floor(img.h / size_y);
  for (int r = 0; r < ct_cols * ct_rows; r++) {
    int x = r % ct_cols * size_x;
    int y = r / ct_cols * size_y;
    if (x + size_x > img.w || y + size_y > img.h) continue;
    std::vector<KeyPoint> pts_new;
    fast9_16(img.roi(x, y, x + size_x, y + size_y), threshold, nms, pts_new);
    std::stable_sort(pts_new.begin(), pts_new.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
    for (size_t i = 0; i < (size_t)num_features_grid && i < pts_new.size(); i++) {
      KeyPoint pt_cor = pts_new[i];
      pt_cor.x += x;
      pt_cor.y += y;
      pts.push_back(pt_cor);
    }
  }
}

// ---- cv::CLAHE (8-bit) ----
void clahe_apply(const View& src, double clipLimit_, int tilesX_, int tilesY_, uint8_t* dst, ptrdiff_t dstep) {
  const int histSize = 256;
  // CLAHE_Impl::apply: extend to a multiple of the tile grid when needed
  std::vector<uint8_t> ext;
  View srcForLut = src;
  int tileW, tileH;
  if (src.w % tilesX_ == 0 && src.h % tilesY_ == 0) {
    tileW = src.w / tilesX_, tileH = src.h / tilesY_;
  } else {
    const int bottom = tilesY_ - (src.h % tilesY_), right = tilesX_ - (src.w % tilesX_);
    const int ew = src.w + right, eh = src.h + bottom;
    ext.resize((size_t)ew * eh);
    copy_make_border_reflect101(src, ext.data(), ew, 0, bottom, 0, right);
    srcForLut = View{ext.data(), ew, eh, ew};
    tileW = ew / tilesX_, tileH = eh / tilesY_;
  }
  const int tileSizeTotal = tileW * tileH;
  const float lutScale = static_cast<float>(histSize - 1) / tileSizeTotal;
  int clipLimit = 0;
  if (clipLimit_ > 0.0) {
    clipLimit = static_cast<int>(clipLimit_ * tileSizeTotal / histSize);
    clipLimit = std::max(clipLimit, 1);
  }
  std::vector<uint8_t> lut((size_t)tilesX_ * tilesY_ * histSize);
  // CLAHE_CalcLut_Body
  for (int k = 0; k < tilesX_ * tilesY_; ++k) {
    const int ty = k / tilesX_, tx = k % tilesX_;
    uint8_t* tileLut = &lut[(size_t)k * histSize];
    int tileHist[256] = {0};
    for (int y = 0; y < tileH; ++y) {
      const uint8_t* ptr = srcForLut.row(ty * tileH + y) + tx * tileW;
      for (int x = 0; x < tileW; ++x) tileHist[ptr[x]]++;
    }
    if (clipLimit > 0) {
      int clipped = 0;
      for (int i = 0; i < histSize; ++i) {
        if (tileHist[i] > clipLimit) {
          clipped += tileHist[i] - clipLimit;
          tileHist[i] = clipLimit;
        }
      }
      int redistBatch = clipped / histSize;
      int residual = clipped - redistBatch * histSize;
      for (int i = 0; i < histSize; ++i) tileHist[i] += redistBatch;
      if (residual != 0) {
        int residualStep = std::max(histSize / residual, 1);
        for (int i = 0; i < histSize && residual > 0; i += residualStep, residual--) tileHist[i]++;
      }
    }
    int sum = 0;
    for (int i = 0; i < histSize; ++i) {
      sum += tileHist[i];
      int v = cv_round_f(sum * lutScale);  // saturate_cast<uchar>(float)
      tileLut[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  // CLAHE_Interpolation_Body
  std::vector<int> ind1_p(src.w), ind2_p(src.w);
  std::vector<float> xa_p(src.w), xa1_p(src.w);
  const int lut_step = histSize;
  float inv_tw = 1.0f / tileW;
  for (int x = 0; x < src.w; ++x) {
    float txf = x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf);
    int tx2 = tx1 + 1;
    xa_p[x] = txf - tx1;
    xa1_p[x] = 1.0f - xa_p[x];
    tx1 = std::max(tx1, 0);
    tx2 = std::min(tx2, tilesX_ - 1);
    ind1_p[x] = tx1 * lut_step;
    ind2_p[x] = tx2 * lut_step;
  }
  float inv_th = 1.0f / tileH;
  for (int y = 0; y < src.h; ++y) {
    const uint8_t* srcRow = src.row(y);
    uint8_t* dstRow = dst + (ptrdiff_t)y * dstep;
    float tyf = y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf);
    int ty2 = ty1 + 1;
    float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = std::max(ty1, 0);
    ty2 = std::min(ty2, tilesY_ - 1);
    const uint8_t* lutPlane1 = &lut[(size_t)ty1 * tilesX_ * histSize];
    const uint8_t* lutPlane2 = &lut[(size_t)ty2 * tilesX_ * histSize];
    for (int x = 0; x < src.w; ++x) {
      int srcVal = srcRow[x];
      int ind1 = ind1_p[x] + srcVal;
      int ind2 = ind2_p[x] + srcVal;
      float res = (lutPlane1[ind1] * xa1_p[x] + lutPlane1[ind2] * xa_p[x]) * ya1 + (lutPlane2[ind1] * xa1_p[x] + lutPlane2[ind2] * xa_p[x]) * ya;
      int v = cv_round_f(res);
      dstRow[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

}  // namespace orc
