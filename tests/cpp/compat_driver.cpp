// Drives the USLAM::ORBextractor / USLAM::ORBmatcher adaptors (include/uvo/compat/) the way src/Tracking.cc does:
// construct once, call per frame.  Frame / MapPoint below carry the member names src/ORBmatcher.cc:49-125 reads.
// Reads a raw u8 image + map-point table from files written by tests/test_cpp_compat.py and writes the results back.
#include <cstdio>
#include <cstdlib>
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
  return 0;
}
