// Pin kit, step 2: reference vectors from the UNMODIFIED src/ORBextractor.cc of chintha/U-VIP-SLAM linked against a REAL OpenCV 3.4.x.
// Not built in this repository's image (it has no OpenCV / Eigen / ROS): see tools/pin/README.md for the build line.
//
//   pin_dump <input dir written by make_inputs.py> <output dir>
//
// Every array goes to <output dir>/<name>.bin, described by one line of <output dir>/manifest.txt: name dtype ndim dims...
// tools/pin/pack_npz.py turns that into tests/golden/reference_pins.npz.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include <opencv2/calib3d/calib3d.hpp>
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#include <opencv2/video/tracking.hpp>

#include "ORBextractor.h"  // the reference's own header (-I$REF/include)

// ---- never-reusing bump allocator: list-node addresses then grow in creation order, which turns the pointer-valued tie-break of
// DistributeOctTree (src/ORBextractor.cc:1151) into "newest node first" (SURVEY.md Appendix C).  -DPIN_SYSTEM_ALLOCATOR turns it off.
#ifndef PIN_SYSTEM_ALLOCATOR
namespace {
const size_t kArena = (size_t)6 << 30;
char* g_arena = nullptr;
size_t g_used = 0;
void* bump(size_t n) {
  if (!g_arena) g_arena = static_cast<char*>(std::malloc(kArena));
  n = (n + 15) & ~(size_t)15;
  if (!g_arena || g_used + n > kArena) {
    std::fprintf(stderr, "pin_dump: bump arena exhausted\n");
    std::abort();
  }
  void* p = g_arena + g_used;
  g_used += n;
  return p;
}
}  // namespace
void* operator new(size_t n) { return bump(n); }
void* operator new[](size_t n) { return bump(n); }
void operator delete(void*) noexcept {}
void operator delete[](void*) noexcept {}
void operator delete(void*, size_t) noexcept {}
void operator delete[](void*, size_t) noexcept {}
#endif

namespace {

std::string g_out;
std::ofstream g_manifest;

void put(const std::string& name, const char* dtype, const void* data, size_t elem, const std::vector<size_t>& dims) {
  size_t n = 1;
  for (size_t d : dims) n *= d;
  std::ofstream f(g_out + "/" + name + ".bin", std::ios::binary);
  if (n) f.write(static_cast<const char*>(data), (std::streamsize)(n * elem));
  g_manifest << name << " " << dtype << " " << dims.size();
  for (size_t d : dims) g_manifest << " " << d;
  g_manifest << "\n";
}
void put_mat_u8(const std::string& name, const cv::Mat& m) {
  cv::Mat c = m.clone();  // tight rows
  put(name, "u1", c.data, 1, {(size_t)c.rows, (size_t)c.cols});
}
std::vector<char> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "pin_dump: cannot read %s\n", path.c_str());
    std::exit(2);
  }
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
void put_keypoints(const std::string& name, const std::vector<cv::KeyPoint>& k) {
  static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint is expected to be 28 bytes");
  put(name, "kp", k.empty() ? nullptr : &k[0], 28, {k.size()});
}

// protected members of the reference class, reached without touching its source
struct Probe : USLAM::ORBextractor {
  Probe(int n, float s, int l, int score, int th) : USLAM::ORBextractor(n, s, l, score, th) {}
  using USLAM::ORBextractor::ComputePyramid;
  using USLAM::ORBextractor::DistributeOctTree;
  using USLAM::ORBextractor::mvImagePyramid;
};

// the padded parent buffer of a pyramid level (the ROI sits at (16, 16) of it: src/ORBextractor.cc:969-971)
cv::Mat padded(const cv::Mat& level) {
  cv::Mat p = level;
  p.adjustROI(16, 16, 16, 16);
  return p;
}

struct Frame {
  cv::Mat img;
  int nfeatures, fast_th;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: pin_dump <input dir> <output dir>\n");
    return 2;
  }
  const std::string in = argv[1];
  g_out = argv[2];
  g_manifest.open(g_out + "/manifest.txt");
  {
    std::ofstream bi(g_out + "/build_info.txt");
    bi << cv::getBuildInformation();
  }
  std::map<std::string, Frame> frames;
  std::ifstream cases(in + "/cases.txt");
  std::string line;
  int nfast = 0;
  while (std::getline(cases, line)) {
    std::istringstream ss(line);
    std::string kind;
    ss >> kind;
    if (kind == "frame") {
      std::string name, file;
      int w, h, nf, th;
      ss >> name >> w >> h >> nf >> th >> file;
      std::vector<char> raw = slurp(in + "/" + file);
      Frame f;
      f.img = cv::Mat(h, w, CV_8UC1, raw.data()).clone();
      f.nfeatures = nf, f.fast_th = th;
      frames[name] = f;
      // ---- un-blurred pyramid: ORBextractor::ComputePyramid (:963-1004) ----
      Probe ex(nf, 1.2f, 8, 0, th);
      ex.ComputePyramid(f.img);
      for (int l = 0; l < 8; ++l) put_mat_u8(name + "/pyr_L" + std::to_string(l), padded(ex.mvImagePyramid[l]));
      // ---- the whole extractor, FullDetect (:849-961; call site src/Tracking.cc:946) ----
      std::vector<cv::KeyPoint> kps;
      cv::Mat desc;
      Eigen::MatrixXi grid = Eigen::MatrixXi::Zero(h / 20 + 2, w / 20 + 2);
      int min_px = 20;
      ex(f.img, cv::Mat(), kps, desc, grid, min_px, true, 0);
      put_keypoints(name + "/kp", kps);
      cv::Mat dc = desc.empty() ? cv::Mat(0, 32, CV_8U) : desc.clone();
      put(name + "/desc", "u1", dc.data, 1, {(size_t)dc.rows, (size_t)32});
      // after operator() the levels that kept keypoints are blurred in place, their pad ring is not (:941-942)
      for (int l = 0; l < 8; ++l) put_mat_u8(name + "/blur_L" + std::to_string(l), padded(ex.mvImagePyramid[l]));
    } else if (kind == "topup") {
      std::string name, fkp, fgrid;
      int n_in, rows, cols, d, need;
      ss >> name >> n_in >> rows >> cols >> d >> need >> fkp >> fgrid;
      const Frame& f = frames.at(name);
      std::vector<char> rk = slurp(in + "/" + fkp), rg = slurp(in + "/" + fgrid);
      std::vector<cv::KeyPoint> kps(n_in);
      std::memcpy(kps.data(), rk.data(), (size_t)n_in * 28);
      Eigen::MatrixXi grid(rows, cols);  // column-major, like the file
      std::memcpy(grid.data(), rg.data(), (size_t)rows * cols * 4);
      Probe ex(f.nfeatures, 1.2f, 8, 0, f.fast_th);
      cv::Mat desc;
      int min_px = d;
      ex(f.img, cv::Mat(), kps, desc, grid, min_px, false, need);
      put_keypoints(name + "/kp_topup", kps);
      cv::Mat dc = desc.empty() ? cv::Mat(0, 32, CV_8U) : desc.clone();
      put(name + "/desc_topup", "u1", dc.data, 1, {(size_t)dc.rows, (size_t)32});
      put(name + "/grid_topup", "i4", grid.data(), 4, {(size_t)cols, (size_t)rows});  // column-major: [col][row]
    } else if (kind == "fast") {
      std::string name;
      int x, y, w, h, th;
      ss >> name >> x >> y >> w >> h >> th;
      const Frame& f = frames.at(name);
      std::vector<cv::KeyPoint> k;
      cv::FAST(f.img(cv::Rect(x, y, w, h)), k, th, true);  // :792 / :797
      const int32_t roi[5] = {x, y, w, h, th};
      put(name + "/fast_" + std::to_string(nfast) + "_roi", "i4", roi, 4, {5});
      put_keypoints(name + "/fast_" + std::to_string(nfast), k);
      ++nfast;
    } else if (kind == "knn") {
      // Utils::ratioMatching (include/utils.h:92-101): BFMatcher(NORM_HAMMING).knnMatch(descriptors_1, descriptors_2, matches, 2)
      std::string name, fq, ft;
      int nq, nt;
      ss >> name >> nq >> nt >> fq >> ft;
      std::vector<char> rq = slurp(in + "/" + fq), rt = slurp(in + "/" + ft);
      cv::Mat Q(nq, 32, CV_8U, rq.data()), T(nt, 32, CV_8U, rt.data());
      cv::BFMatcher matcher(cv::NORM_HAMMING);
      std::vector<std::vector<cv::DMatch> > m;
      matcher.knnMatch(Q, T, m, 2);
      std::vector<int32_t> idx(2 * (size_t)nq, -1), dist(2 * (size_t)nq, -1);
      for (int i = 0; i < nq; ++i)
        for (size_t k = 0; k < m[i].size() && k < 2; ++k) idx[2 * i + k] = m[i][k].trainIdx, dist[2 * i + k] = (int32_t)m[i][k].distance;
      put("knn_" + name + "_q", "u1", rq.data(), 1, {(size_t)nq, (size_t)32});
      put("knn_" + name + "_t", "u1", rt.data(), 1, {(size_t)nt, (size_t)32});
      put("knn_" + name + "_idx", "i4", idx.data(), 4, {(size_t)nq, (size_t)2});
      put("knn_" + name + "_dist", "i4", dist.data(), 4, {(size_t)nq, (size_t)2});
    } else if (kind == "gauss") {
      // the blur exactly as operator() applies it (:942): in place on the ROI of a padded parent, border flag without BORDER_ISOLATED
      int k, w, h;
      std::string file;
      ss >> k >> w >> h >> file;
      std::vector<char> raw = slurp(in + "/" + file);
      cv::Mat parent = cv::Mat(h + 32, w + 32, CV_8UC1, raw.data()).clone();
      put_mat_u8("gauss_" + std::to_string(k) + "_in", parent);
      cv::Mat roi = parent(cv::Rect(16, 16, w, h));
      cv::GaussianBlur(roi, roi, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
      put_mat_u8("gauss_" + std::to_string(k) + "_out", parent);
    } else if (kind == "atan2") {
      int n;
      std::string fy, fx;
      ss >> n >> fy >> fx;
      std::vector<char> ry = slurp(in + "/" + fy), rx = slurp(in + "/" + fx);
      const float* y = reinterpret_cast<const float*>(ry.data());
      const float* x = reinterpret_cast<const float*>(rx.data());
      std::vector<float> deg(n);
      for (int i = 0; i < n; ++i) deg[i] = cv::fastAtan2(y[i], x[i]);  // :151
      put("atan2_y", "f4", y, 4, {(size_t)n});
      put("atan2_x", "f4", x, 4, {(size_t)n});
      put("atan2_deg", "f4", deg.data(), 4, {(size_t)n});
    } else if (kind == "octree") {
      int k, n, minX, maxX, minY, maxY, N, level;
      std::string file;
      ss >> k >> n >> minX >> maxX >> minY >> maxY >> N >> level >> file;
      std::vector<char> raw = slurp(in + "/" + file);
      std::vector<cv::KeyPoint> cand(n);
      std::memcpy(cand.data(), raw.data(), (size_t)n * 28);
      Probe ex(1000, 1.2f, 8, 0, 20);
      std::vector<cv::KeyPoint> out = ex.DistributeOctTree(cand, minX, maxX, minY, maxY, N, level);  // :1006-1230
      const int32_t par[6] = {minX, maxX, minY, maxY, N, level};
      put("oct_" + std::to_string(k) + "_par", "i4", par, 4, {6});
      put_keypoints("oct_" + std::to_string(k) + "_in", cand);
      put_keypoints("oct_" + std::to_string(k) + "_out", out);
    } else if (kind == "gemm") {
      int n;
      std::string fR, fP, ft;
      ss >> n >> fR >> fP >> ft;
      std::vector<char> rR = slurp(in + "/" + fR), rP = slurp(in + "/" + fP), rt = slurp(in + "/" + ft);
      std::vector<float> out(3 * (size_t)n), neg(3 * (size_t)n), nrm(n);
      std::vector<double> dot(n);
      for (int i = 0; i < n; ++i) {
        cv::Mat R(3, 3, CV_32F, rR.data() + (size_t)i * 36), P(3, 1, CV_32F, rP.data() + (size_t)i * 12), t(3, 1, CV_32F, rt.data() + (size_t)i * 12);
        cv::Mat x3Dc = R * P + t;       // src/ORBmatcher.cc:1650, src/FrameKTL.cc:309
        cv::Mat Ow = -R.t() * t;        // src/ORBmatcher.cc:1628
        for (int c = 0; c < 3; ++c) out[3 * i + c] = x3Dc.at<float>(c), neg[3 * i + c] = Ow.at<float>(c);
        nrm[i] = (float)cv::norm(P);    // src/ORBmatcher.cc:1663
        dot[i] = P.dot(t);              // src/FrameKTL.cc:335
      }
      put("gemm_R", "f4", rR.data(), 4, {(size_t)n, 3, 3});
      put("gemm_P", "f4", rP.data(), 4, {(size_t)n, 3});
      put("gemm_t", "f4", rt.data(), 4, {(size_t)n, 3});
      put("gemm_out", "f4", out.data(), 4, {(size_t)n, 3});
      put("gemm_negRt_out", "f4", neg.data(), 4, {(size_t)n, 3});
      put("gemm_norm", "f4", nrm.data(), 4, {(size_t)n});
      put("gemm_dot", "f8", dot.data(), 8, {(size_t)n});
    } else if (kind == "clahe") {
      std::string name;
      double clip;
      int tx, ty;
      ss >> name >> clip >> tx >> ty;
      cv::Ptr<cv::CLAHE> clahe = cv::createCLAHE(clip, cv::Size(tx, ty));  // src/Tracking.cc:426-430
      cv::Mat dst;
      clahe->apply(frames.at(name).img, dst);
      put_mat_u8(name + "/clahe", dst);
    } else if (kind == "undistort") {
      std::string name, file;
      int fisheye, n, nd;
      double fx, fy, cx, cy;
      ss >> name >> fisheye >> n >> fx >> fy >> cx >> cy >> nd;
      cv::Mat K = cv::Mat::eye(3, 3, CV_32F), D(nd, 1, CV_32F);  // mK / mDistCoef are CV_32F (src/Tracking.cc:100-130)
      K.at<float>(0, 0) = (float)fx, K.at<float>(1, 1) = (float)fy, K.at<float>(0, 2) = (float)cx, K.at<float>(1, 2) = (float)cy;
      for (int i = 0; i < nd; ++i) {
        double v;
        ss >> v;
        D.at<float>(i) = (float)v;
      }
      ss >> file;
      std::vector<char> raw = slurp(in + "/" + file);
      std::vector<float> out(2 * (size_t)n);
      for (int i = 0; i < n; ++i) {  // exactly Tracking::undistort_point (:1265-1283), point by point
        cv::Mat mat(1, 2, CV_32F);
        mat.at<float>(0, 0) = reinterpret_cast<const float*>(raw.data())[2 * i];
        mat.at<float>(0, 1) = reinterpret_cast<const float*>(raw.data())[2 * i + 1];
        mat = mat.reshape(2);
        if (fisheye)
          cv::fisheye::undistortPoints(mat, mat, K, D, cv::Mat(), K);
        else
          cv::undistortPoints(mat, mat, K, D, cv::Mat(), K);
        mat = mat.reshape(1);
        out[2 * i] = mat.at<float>(0, 0), out[2 * i + 1] = mat.at<float>(0, 1);
      }
      const float par[4] = {(float)fx, (float)fy, (float)cx, (float)cy};
      put("undistort_" + name + "_K", "f4", par, 4, {4});
      put("undistort_" + name + "_D", "f4", D.data, 4, {(size_t)nd});
      put("undistort_" + name + "_in", "f4", raw.data(), 4, {(size_t)n, 2});
      put("undistort_" + name + "_out", "f4", out.data(), 4, {(size_t)n, 2});
    } else if (kind == "klt") {
      std::string a, b, fpts;
      int ww, wh, maxLevel, iters, n;
      double eps, minEig;
      ss >> a >> b >> ww >> wh >> maxLevel >> iters >> eps >> minEig >> n >> fpts;
      std::vector<cv::Mat> pa, pb;
      cv::buildOpticalFlowPyramid(frames.at(a).img, pa, cv::Size(ww, wh), maxLevel);  // src/FrameKTL.cc:76 (withDerivatives = true)
      cv::buildOpticalFlowPyramid(frames.at(b).img, pb, cv::Size(ww, wh), maxLevel);
      for (size_t l = 0; l + 1 < pa.size(); l += 2) {  // (image, derivative) pairs
        put_mat_u8("klt_pyr_L" + std::to_string(l / 2), pa[l]);
        cv::Mat d = pa[l + 1].clone();
        put("klt_deriv_L" + std::to_string(l / 2), "i2", d.data, 2, {(size_t)d.rows, (size_t)d.cols, 2});
      }
      std::vector<char> rp = slurp(in + "/" + fpts);
      std::vector<cv::Point2f> p0(n), p1;
      std::memcpy(p0.data(), rp.data(), (size_t)n * 8);
      p1 = p0;  // OPTFLOW_USE_INITIAL_FLOW with the previous positions as the guess (src/Tracking.cc:1040-1047)
      std::vector<uchar> status;
      std::vector<float> err;
      cv::calcOpticalFlowPyrLK(pa, pb, p0, p1, status, err, cv::Size(ww, wh), maxLevel,
                               cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, iters, eps),
                               cv::OPTFLOW_USE_INITIAL_FLOW + cv::OPTFLOW_LK_GET_MIN_EIGENVALS, minEig);
      put("klt_pts0", "f4", p0.data(), 4, {(size_t)n, 2});
      put("klt_pts1", "f4", p1.data(), 4, {(size_t)n, 2});
      put("klt_status", "u1", status.data(), 1, {(size_t)n});
      put("klt_err", "f4", err.data(), 4, {(size_t)n});
    }
  }
  std::printf("pin_dump: done (%s)\n", g_out.c_str());
  return 0;
}
