// Drives the USLAM::ORBextractor / USLAM::ORBmatcher adaptors (include/uvo/compat/) the way src/Tracking.cc does:
// construct once, call per frame.  Frame / MapPoint below carry the member names src/ORBmatcher.cc:49-125 reads.
// Reads a raw u8 image + map-point table from files written by tests/test_cpp_compat.py and writes the results back.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#include "uvo/compat/ORBextractor.h"
#include "uvo/compat/ORBmatcher.h"

struct DescRows {  // the two cv::Mat calls the adaptor makes
  std::vector<uint8_t> d;
  const uint8_t* ptr(int i) const { return &d[(size_t)i * 32]; }
};
struct MapPoint {
  bool mbTrackInView = true, bad = false;
  int mnTrackScaleLevel = 0;
  float mTrackViewCos = 0.9f, mTrackProjX = 0, mTrackProjY = 0;
  DescRows desc;
  bool isBad() const { return bad; }
  const DescRows& GetDescriptor() const { return desc; }
};
struct Frame {
  std::vector<uvo_keypoint> mvKeysUn;
  DescRows mDescriptors;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<float> mvScaleFactors;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
};

// ---- stand-ins with the member names the other search loops read (src/ORBmatcher.cc:155-284, :715-1134, :1622-1746) ----
struct Vec3 {  // cv::Mat 3x1 CV_32F
  float v[3];
  template <class T>
  T at(int i) const { return v[i]; }
};
struct Mat44 {  // cv::Mat 4x4 / 3x3 CV_32F
  float m[4][4];
  template <class T>
  T at(int r, int c) const { return m[r][c]; }
  template <class T>
  T at(int r) const { return m[r][0]; }
};
struct Desc1 {
  const uint8_t* p;
  const uint8_t* ptr(int) const { return p; }
};
struct KeyFrame;
struct MP2 {
  Vec3 pos, normal;
  float mind = 1.f, maxd = 100.f;
  std::vector<uint8_t> d;
  bool bad = false;
  int replaced = 0, observations = 0;
  bool isBad() const { return bad; }
  Vec3 GetWorldPos() const { return pos; }
  Vec3 GetNormal() const { return normal; }
  float GetMinDistanceInvariance() const { return 0.8f * mind; }
  float GetMaxDistanceInvariance() const { return 1.2f * maxd; }
  Desc1 GetDescriptor() const { return Desc1{d.data()}; }
  bool IsInKeyFrame(KeyFrame*) const { return false; }
  int index_in_kf = -1;
  int GetIndexInKeyFrame(KeyFrame*) const { return index_in_kf; }
  void Replace(MP2*) { ++replaced; }
  void AddObservation(KeyFrame*, int) { ++observations; }
};
struct KeyFrame {
  int N = 0;
  std::vector<uvo_keypoint> keys;
  std::vector<uint8_t> desc;
  std::vector<MP2*> mps;
  std::map<unsigned, std::vector<unsigned> > featvec;
  std::vector<float> scale, sigma2;
  Mat44 R, t3, ow;
  float fx = 458.f, fy = 457.f, cx = 367.f, cy = 248.f;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 752, mnMaxY = 480;
  std::vector<MP2*> GetMapPointMatches() const { return mps; }
  std::vector<uvo_keypoint> GetKeyPointsUn() const { return keys; }
  uvo_keypoint GetKeyPointUn(int i) const { return keys[i]; }
  Desc1 GetDescriptor(int i) const { return Desc1{&desc[(size_t)i * 32]}; }
  const std::map<unsigned, std::vector<unsigned> >& GetFeatureVector() const { return featvec; }
  int GetScaleLevels() const { return (int)scale.size(); }
  float GetSigma2(int l) const { return sigma2[l]; }
  std::vector<float> GetScaleFactors() const { return scale; }
  Mat44 GetRotation() const { return R; }
  Mat44 GetTranslation() const { return t3; }
  Mat44 GetCameraCenter() const { return ow; }
  MP2* GetMapPoint(int i) const { return mps[i]; }
  std::set<MP2*> GetMapPoints() const {
    std::set<MP2*> s(mps.begin(), mps.end());
    s.erase(nullptr);
    return s;
  }
  void AddMapPoint(MP2* p, int i) { mps[i] = p; }
};
struct Frame2 {
  std::vector<uvo_keypoint> mvKeysUn, mvKeys;
  DescRows mDescriptors;
  std::vector<MP2*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<float> mvScaleFactors;
  std::map<unsigned, std::vector<unsigned> > mFeatVec;
  Mat44 mTcw;
  float fx = 458.f, fy = 457.f, cx = 367.f, cy = 248.f;
  float mnMinX = 0, mnMinY = 0, mnMaxX = 752, mnMaxY = 480;
};

// exercises the remaining adaptor members on data derived from one extraction (identity pose, every keypoint a map point
// back-projected at depth 5); returns non-zero on an implausible outcome
static int exercise_other_members(const std::vector<uvo_keypoint>& kps, const std::vector<uint8_t>& desc, float sf) {
  const int n = (int)kps.size();
  KeyFrame kf1, kf2;
  std::vector<MP2> pts(n);
  for (KeyFrame* k : {&kf1, &kf2}) {
    k->N = n, k->keys = kps, k->desc = desc;
    k->scale.assign(8, 1.f), k->sigma2.assign(8, 1.f);
    for (int l = 1; l < 8; ++l) k->scale[l] = k->scale[l - 1] * sf, k->sigma2[l] = k->scale[l] * k->scale[l];
    memset(&k->R, 0, sizeof(Mat44)), memset(&k->t3, 0, sizeof(Mat44)), memset(&k->ow, 0, sizeof(Mat44));
    for (int i = 0; i < 3; ++i) k->R.m[i][i] = 1.f;
    for (int i = 0; i < n; ++i) k->featvec[desc[(size_t)i * 32] & 15u].push_back((unsigned)i);
  }
  for (int i = 0; i < n; ++i) {
    const float z = 5.f;
    pts[i].pos = Vec3{{(kps[i].x - 367.f) / 458.f * z, (kps[i].y - 248.f) / 457.f * z, z}};
    pts[i].normal = Vec3{{0.f, 0.f, 1.f}};
    pts[i].mind = 5.f / kf1.scale[kps[i].octave] / 0.8f * 0.999f;  // dist / (0.8 * mind) just above scale[octave - 1]
    pts[i].maxd = 50.f;
    pts[i].d.assign(desc.begin() + (size_t)i * 32, desc.begin() + (size_t)i * 32 + 32);
  }
  kf1.mps.assign(n, nullptr), kf2.mps.assign(n, nullptr);
  for (int i = 0; i < n; ++i) kf1.mps[i] = &pts[i], kf2.mps[i] = &pts[i];
  USLAM::ORBmatcher m(0.9f, true);
  std::vector<MP2*> v12;
  const int nb = m.SearchByBoW(&kf1, &kf2, v12);  // identical key frames: every descriptor finds itself
  if (nb < n * 9 / 10) return 10;
  Frame2 F;
  F.mvKeysUn = kps, F.mvKeys = kps, F.mDescriptors.d = desc, F.mvpMapPoints.assign(n, nullptr), F.mvScaleFactors = kf1.scale;
  F.mFeatVec = kf1.featvec;
  memset(&F.mTcw, 0, sizeof(Mat44));
  for (int i = 0; i < 4; ++i) F.mTcw.m[i][i] = 1.f;
  std::vector<MP2*> vF;
  const int nbf = m.SearchByBoW(&kf1, F, vF);
  if (nbf < n * 9 / 10) return 11;
  std::set<MP2*> found;
  const int np = m.SearchByProjection(F, &kf1, found, 10.f, 100);
  if (np < n * 8 / 10) return 12;
  // the four members nothing in the reference calls, on two identical frames at the identity pose
  {
    struct P2 {
      float x, y;
    };
    Frame2 A = F, B = F;
    A.mvpMapPoints.assign(n, nullptr), B.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; ++i) A.mvpMapPoints[i] = &pts[i];
    A.mvbOutlier.assign(n, false), B.mvbOutlier.assign(n, false);
    std::vector<MP2*> v2;
    const int nw = m.WindowSearch(A, B, 10, v2);  // every keypoint finds itself at distance 0
    if (nw < n * 8 / 10 || (int)v2.size() != n) return 20;
    const int nq = m.SearchByProjection(A, B, 10, v2);
    if (nq < n * 8 / 10) return 21;
    std::vector<P2> prev(n);
    for (int i = 0; i < n; ++i) prev[i].x = kps[i].x + 1.f, prev[i].y = kps[i].y;
    std::vector<int> m12;
    USLAM::ORBmatcher mi(0.9f, true);
    const int ni = mi.SearchForInitialization(A, B, prev, m12, 10);
    int lvl0 = 0;
    for (int i = 0; i < n; ++i) lvl0 += kps[i].octave == 0;
    if (ni < lvl0 * 8 / 10 || ni > lvl0) return 22;
    for (int i = 0; i < n; ++i)
      if (m12[i] >= 0 && (prev[i].x != kps[m12[i]].x || prev[i].y != kps[m12[i]].y)) return 23;
    const int nl = m.SearchByProjection(B, static_cast<const Frame2&>(A), 7.f);
    if (nl < n * 8 / 10) return 24;
  }
  // triangulation between two key frames without map points; F12 of a pure x translation: epipolar lines y = const
  KeyFrame ka = kf1, kb = kf2;
  ka.mps.assign(n, nullptr), kb.mps.assign(n, nullptr);
  Mat44 F12;
  memset(&F12, 0, sizeof(F12));
  F12.m[1][2] = -1.f, F12.m[2][1] = 1.f;
  std::vector<uvo_keypoint> k1, k2;
  std::vector<std::pair<size_t, size_t> > pairs;
  const int nt = m.SearchForTriangulation(&ka, &kb, F12, k1, k2, pairs);
  if (nt < n * 8 / 10 || (int)pairs.size() != nt) return 13;
  // Fuse the same map points into a key frame that holds none: every usable point adds an observation
  KeyFrame kc = kf1;
  kc.mps.assign(n, nullptr);
  std::vector<MP2*> vp2(n);
  for (int i = 0; i < n; ++i) vp2[i] = &pts[i];
  const int nf = m.Fuse(&kc, vp2, 3.f);
  if (nf < n * 8 / 10) return 14;
  // loop-closing forms with Scw = [2 I | 0]: scale 2 drops out of Rcw / tcw, so every point projects onto its own key point
  Mat44 Scw;
  memset(&Scw, 0, sizeof(Scw));
  for (int i = 0; i < 3; ++i) Scw.m[i][i] = 2.f;
  Scw.m[3][3] = 1.f;
  KeyFrame kd = kf1;
  kd.mps.assign(n, nullptr);
  std::vector<MP2*> vpMatched(n, nullptr);
  const int ns = m.SearchByProjection(&kd, Scw, vp2, vpMatched, 10);
  int placed = 0;
  for (int i = 0; i < n; ++i) placed += vpMatched[i] != nullptr;
  if (ns < n * 8 / 10 || placed != ns) return 15;
  const int nfs = m.Fuse(&kd, Scw, vp2, 4.f);
  if (nfs < n * 8 / 10) return 16;
  // SearchBySim3 between two identical key frames under the identity Sim3: mutual matches for (nearly) every point
  Mat44 I3;
  memset(&I3, 0, sizeof(I3));
  for (int i = 0; i < 3; ++i) I3.m[i][i] = 1.f;
  Mat44 zero;
  memset(&zero, 0, sizeof(zero));
  std::vector<MP2*> vm12(n, nullptr);
  const float s12 = 1.f;
  const int n3 = m.SearchBySim3(&kf1, &kf2, vm12, s12, I3, zero, 7.5f);
  if (n3 < n * 8 / 10) return 17;
  printf("other members: bow_kk=%d bow_kf=%d proj_kf=%d triang=%d fuse=%d proj_scw=%d fuse_scw=%d sim3=%d of %d\n", nb, nbf, np, nt, nf, ns, nfs, n3, n);
  return 0;
}

static std::vector<uint8_t> slurp(const char* p) {
  FILE* f = fopen(p, "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", p);
    exit(2);
  }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> v(n);
  if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const int w = atoi(argv[2]), h = atoi(argv[3]);
  std::vector<uint8_t> img = slurp(argv[1]);
  USLAM::ORBextractor ex(1000, 1.2f, 8, USLAM::ORBextractor::HARRIS_SCORE, 7);
  std::vector<uvo_keypoint> kps;
  std::vector<uint8_t> desc;
  int min_px = 20;
  if (ex.extract(img.data(), w, h, w, kps, desc, nullptr, 0, 0, min_px, true, 0) != UVO_OK) {
    fprintf(stderr, "extract failed: %s\n", ex.last_error().c_str());
    return 1;
  }
  // second call through the same object (scratch reuse), must be identical
  std::vector<uvo_keypoint> kps2;
  std::vector<uint8_t> desc2;
  ex.extract(img.data(), w, h, w, kps2, desc2, nullptr, 0, 0, min_px, true, 0);
  if (kps2.size() != kps.size() || desc2 != desc) return 3;

  // map points: file of records {float x, y; int level; float viewcos; uint8 inview; uint8 desc[32]} packed
  std::vector<uint8_t> mp = slurp(argv[4]);
  const size_t rec = 4 + 4 + 4 + 4 + 1 + 32;
  const int nmp = (int)(mp.size() / rec);
  std::vector<MapPoint> pts(nmp);
  std::vector<MapPoint*> vp(nmp);
  for (int i = 0; i < nmp; ++i) {
    const uint8_t* r = &mp[i * rec];
    memcpy(&pts[i].mTrackProjX, r, 4), memcpy(&pts[i].mTrackProjY, r + 4, 4), memcpy(&pts[i].mnTrackScaleLevel, r + 8, 4);
    memcpy(&pts[i].mTrackViewCos, r + 12, 4);
    pts[i].mbTrackInView = r[16] != 0;
    pts[i].desc.d.assign(r + 17, r + 17 + 32);
    vp[i] = &pts[i];
  }
  Frame F;
  F.mvKeysUn = kps;
  F.mDescriptors.d = desc;
  F.mvpMapPoints.assign(kps.size(), nullptr);
  F.mvScaleFactors.assign(8, 1.f);
  for (int i = 1; i < 8; ++i) F.mvScaleFactors[i] = F.mvScaleFactors[i - 1] * ex.GetScaleFactor();  // src/FrameKTL.cc:240
  F.mnMaxX = w, F.mnMaxY = h;
  USLAM::ORBmatcher matcher(0.8f);
  const int nmatches = matcher.SearchByProjection(F, vp, 1.0f);

  FILE* o = fopen(argv[5], "wb");
  int n = (int)kps.size();
  fwrite(&n, 4, 1, o);
  fwrite(kps.data(), sizeof(uvo_keypoint), n, o);
  fwrite(desc.data(), 32, n, o);
  fwrite(&nmatches, 4, 1, o);
  for (int i = 0; i < n; ++i) {
    int a = F.mvpMapPoints[i] ? (int)(F.mvpMapPoints[i] - pts.data()) : -1;
    fwrite(&a, 4, 1, o);
  }
  fclose(o);
  printf("ok %d keypoints %d matches dd=%d\n", n, nmatches, USLAM::ORBmatcher::DescriptorDistance(desc.data(), desc.data() + 32));
  return exercise_other_members(kps, desc, ex.GetScaleFactor());
}
