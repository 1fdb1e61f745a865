// ORACLE -- TEST INFRASTRUCTURE ONLY (see orb_oracle.hpp header).  PARITY UNPINNED.
// Plain C entry points so tests/ and bench.py's cpu_baseline leg can drive the oracle through ctypes.
#include <cmath>
#include <cstring>

#include "orb_oracle.hpp"

using namespace orc;

extern "C" {

void* orc_extractor_create(int nfeatures, float scaleFactor, int nlevels, int fastTh) { return new Extractor(nfeatures, scaleFactor, nlevels, fastTh); }
void orc_extractor_destroy(void* h) { delete (Extractor*)h; }

void orc_extractor_tables(void* h, float* scale, float* invscale, int* quota, int* umax16) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i < e->nlevels; ++i) {
    scale[i] = e->mvScaleFactor[i];
    invscale[i] = e->mvInvScaleFactor[i];
    quota[i] = e->mnFeaturesPerLevel[i];
  }
  for (int i = 0; i < 16; ++i) umax16[i] = e->umax[i];
}

// returns number of keypoints written (<= cap) or -needed if cap is too small
int orc_extract(void* h, const uint8_t* img, int w, int hgt, long stride, const KeyPoint* in_kp, int n_in, int32_t* grid2d, int grid_rows,
                int grid_cols, int min_px_dist, int full_detect, int num_feats_needed, KeyPoint* out_kp, uint8_t* out_desc, int cap) {
  Extractor* e = (Extractor*)h;
  std::vector<KeyPoint> kps(in_kp, in_kp + n_in);
  std::vector<uint8_t> desc;
  e->extract(View{(uint8_t*)img, w, hgt, stride}, kps, desc, grid2d, grid_rows, grid_cols, min_px_dist, full_detect != 0, num_feats_needed);
  if ((int)kps.size() > cap) return -(int)kps.size();
  if (!kps.empty()) {
    memcpy(out_kp, kps.data(), kps.size() * sizeof(KeyPoint));
    memcpy(out_desc, desc.data(), desc.size());
  }
  return (int)kps.size();
}

// --- taps into the last extract() call ---
int orc_level_dims(void* h, int level, int* w, int* hgt) {
  Extractor* e = (Extractor*)h;
  if (level < 0 || level >= (int)e->pyr.size()) return -1;
  *w = e->pyr[level].w, *hgt = e->pyr[level].h;
  return 0;
}
// copy the padded plane ((w+32) x (h+32), tight rows); which: 0 = unblurred snapshot, 1 = current (blurred where blurred)
int orc_level_plane(void* h, int level, int which, uint8_t* out) {
  Extractor* e = (Extractor*)h;
  const std::vector<uint8_t>& p = which ? e->planes[level] : e->planes_unblurred[level];
  memcpy(out, p.data(), p.size());
  return (int)p.size();
}
int orc_level_candidates(void* h, int level, KeyPoint* out, int cap) {
  Extractor* e = (Extractor*)h;
  const auto& v = e->dbg_candidates[level];
  int n = (int)v.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
  return n;
}
int orc_level_keypoints(void* h, int level, KeyPoint* out, int cap) {
  Extractor* e = (Extractor*)h;
  const auto& v = e->dbg_level_kps[level];
  int n = (int)v.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
  return n;
}

// --- primitives ---
void orc_border101(const uint8_t* src, int w, int hgt, long stride, uint8_t* dst, int pad) {
  copy_make_border_reflect101(View{(uint8_t*)src, w, hgt, stride}, dst, w + 2 * pad, pad, pad, pad, pad);
}
void orc_resize_linear(const uint8_t* src, int sw, int sh, long sstride, uint8_t* dst, int dw, int dh) {
  resize_linear_u8(View{(uint8_t*)src, sw, sh, sstride}, View{dst, dw, dh, dw});
}
int orc_fast(const uint8_t* img, int w, int hgt, long stride, int threshold, int nms, KeyPoint* out, int cap) {
  std::vector<KeyPoint> v;
  fast9_16(View{(uint8_t*)img, w, hgt, stride}, threshold, nms != 0, v);
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
}
int orc_fast_bruteforce(const uint8_t* img, int w, int hgt, long stride, int threshold, int nms, KeyPoint* out, int cap) {
  std::vector<KeyPoint> v;
  fast9_16_bruteforce(View{(uint8_t*)img, w, hgt, stride}, threshold, nms != 0, v);
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
}
void orc_gauss_taps(int* taps7) { gaussian_taps_7_sigma2(taps7); }
// plane = padded buffer (w+2*pad) x (h+2*pad) tight; blurs the interior in place
void orc_gauss7_padded(uint8_t* plane, int w, int hgt, int pad) {
  View full{plane, w + 2 * pad, hgt + 2 * pad, w + 2 * pad};
  gaussian_blur7_roi_inplace(full.roi(pad, pad, pad + w, pad + hgt));
}
void orc_gauss7_padded_ex(uint8_t* plane, int w, int hgt, int pad, int rounding) {
  View full{plane, w + 2 * pad, hgt + 2 * pad, w + 2 * pad};
  gaussian_blur7_roi_inplace(full.roi(pad, pad, pad + w, pad + hgt), rounding);
}
void orc_extractor_set_blur_rounding(void* h, int rounding) { ((Extractor*)h)->blur_rounding = rounding; }
void orc_extractor_pattern(void* h, int* out1024) {  // the 512 (x, y) points of bit_pattern_31_ as the extractor holds them
  for (int i = 0; i < 1024; ++i) out1024[i] = ((Extractor*)h)->pattern[i];
}
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float orc_ic_angle(void* h, const uint8_t* plane, int w, int hgt, int pad, float x, float y) {
  Extractor* e = (Extractor*)h;
  View full{(uint8_t*)plane, w + 2 * pad, hgt + 2 * pad, w + 2 * pad};
  return ic_angle(full.roi(pad, pad, pad + w, pad + hgt), x, y, e->umax);
}
void orc_descriptor(void* h, const uint8_t* plane, int w, int hgt, int pad, float x, float y, float angle_deg, uint8_t* desc32) {
  Extractor* e = (Extractor*)h;
  View full{(uint8_t*)plane, w + 2 * pad, hgt + 2 * pad, w + 2 * pad};
  KeyPoint kp{x, y, 31.f, angle_deg, 0.f, 0, -1};
  compute_orb_descriptor(kp, full.roi(pad, pad, pad + w, pad + hgt), e->pattern, desc32);
}
void orc_sincosf(float a, float* s, float* c) {
  *s = sinf(a);
  *c = cosf(a);
}
int orc_octree(void* h, const KeyPoint* cand, int n, int minX, int maxX, int minY, int maxY, int N, KeyPoint* out, int cap) {
  Extractor* e = (Extractor*)h;
  std::vector<KeyPoint> v(cand, cand + n);
  std::vector<KeyPoint> r = e->DistributeOctTree(v, minX, maxX, minY, maxY, N);
  for (int i = 0; i < (int)r.size() && i < cap; ++i) out[i] = r[i];
  return (int)r.size();
}
int orc_grider_fast(const uint8_t* img, int w, int hgt, long stride, int num_features, int grid_x, int grid_y, int threshold, int nms,
                    KeyPoint* out, int cap) {
  std::vector<KeyPoint> v;
  grider_fast(View{(uint8_t*)img, w, hgt, stride}, v, num_features, grid_x, grid_y, threshold, nms != 0);
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
}

// --- matcher ---
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* mask, int32_t* idx0, int32_t* d0, int32_t* idx1, int32_t* d1) {
  knn2(q, nq, t, nt, mask, idx0, d0, idx1, d1);
}
int orc_distinctive_descriptor(const uint8_t* desc, int n, int* best_median) { return distinctive_descriptor(desc, n, best_median); }
int orc_features_in_area(const KeyPoint* kps, int n, int minX, int minY, int maxX, int maxY, float x, float y, float r, int minLevel,
                         int maxLevel, int32_t* out, int cap) {
  FrameGrid g;
  g.build(kps, n, minX, minY, maxX, maxY);
  std::vector<int> v = g.GetFeaturesInArea(x, y, r, minLevel, maxLevel);
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
}
int orc_search_by_projection(const KeyPoint* kps, int n, const uint8_t* fdesc, int minX, int minY, int maxX, int maxY, int32_t* assigned,
                             int nmp, const float* projx, const float* projy, const int32_t* level, const float* viewcos,
                             const uint8_t* inview, const uint8_t* mpdesc, const float* scaleFactors, float th, float nnratio) {
  FrameGrid g;
  g.build(kps, n, minX, minY, maxX, maxY);
  return search_by_projection(g, fdesc, assigned, nmp, projx, projy, level, viewcos, inview, mpdesc, scaleFactors, th, nnratio);
}

int orc_search_by_projection_kf(const KeyPoint* kps, int n, const uint8_t* fdesc, int minX, int minY, int maxX, int maxY, int32_t* assigned,
                                int nmp, const float* u, const float* v, const int32_t* level, const uint8_t* valid, const uint8_t* mpdesc,
                                const float* kf_angle, const float* scaleFactors, float th, int orbDist, int checkOri) {
  FrameGrid g;
  g.build(kps, n, minX, minY, maxX, maxY);
  return search_by_projection_kf(g, fdesc, assigned, nmp, u, v, level, valid, mpdesc, kf_angle, scaleFactors, th, orbDist, checkOri != 0);
}

static Camera make_camera(const float* cam23) {
  Camera F;
  memcpy(F.Rcw, cam23, 36), memcpy(F.tcw, cam23 + 9, 12), memcpy(F.Ow, cam23 + 12, 12);
  F.fx = cam23[15], F.fy = cam23[16], F.cx = cam23[17], F.cy = cam23[18];
  F.minX = cam23[19], F.maxX = cam23[20], F.minY = cam23[21], F.maxY = cam23[22];
  return F;
}
int orc_window_search(const KeyPoint* kp1, int n1, const uint8_t* desc1, const uint8_t* has_mp1, const KeyPoint* kp2, int n2, const uint8_t* desc2,
                      int minX, int minY, int maxX, int maxY, int windowSize, int minLevel, int maxLevel, float nnratio, int checkOri,
                      int32_t* match21) {
  FrameGrid g;
  g.build(kp2, n2, minX, minY, maxX, maxY);
  return window_search(kp1, n1, desc1, has_mp1, g, desc2, n2, windowSize, minLevel, maxLevel, nnratio, checkOri != 0, match21);
}
int orc_search_by_projection_frames(const KeyPoint* kp1, int n1, const uint8_t* desc1, const uint8_t* usable1, const float* xyz1, const float* cam23,
                                    const KeyPoint* kp2, int n2, const uint8_t* desc2, int32_t* assigned2, int windowSize, float nnratio) {
  const Camera F2 = make_camera(cam23);
  FrameGrid g;
  g.build(kp2, n2, (int)F2.minX, (int)F2.minY, (int)F2.maxX, (int)F2.maxY);
  return search_by_projection_frames(kp1, n1, desc1, usable1, xyz1, F2, g, desc2, assigned2, windowSize, nnratio);
}
int orc_search_for_initialization(const KeyPoint* kp1, int n1, const uint8_t* desc1, const KeyPoint* kp2, int n2, const uint8_t* desc2, int minX,
                                  int minY, int maxX, int maxY, float* prev_matched, int32_t* vnMatches12, int windowSize, float nnratio,
                                  int checkOri) {
  FrameGrid g;
  g.build(kp2, n2, minX, minY, maxX, maxY);
  return search_for_initialization(kp1, n1, desc1, g, desc2, n2, prev_matched, vnMatches12, windowSize, nnratio, checkOri != 0);
}
int orc_search_by_projection_last(const float* cam23, const KeyPoint* kps, int n, const uint8_t* fdesc, int32_t* assigned, int nlast,
                                  const uint8_t* usable_last, const float* xyz_last, const int32_t* octave_last, const float* angle_last,
                                  const uint8_t* desc_last, const float* scaleFactors, float th, int checkOri) {
  const Camera C = make_camera(cam23);
  FrameGrid g;
  g.build(kps, n, (int)C.minX, (int)C.minY, (int)C.maxX, (int)C.maxY);
  return search_by_projection_last(C, g, fdesc, assigned, nlast, usable_last, xyz_last, octave_last, angle_last, desc_last, scaleFactors, th,
                                   checkOri != 0);
}

int orc_search_by_bow(int kf_kf, const uint32_t* node1, const int32_t* start1, const int32_t* feat1, int nn1, int n1, const uint8_t* desc1,
                      const float* angle1, const uint8_t* usable1, const uint32_t* node2, const int32_t* start2, const int32_t* feat2, int nn2,
                      int n2, const uint8_t* desc2, const float* angle2, const uint8_t* usable2, float nnratio, int checkOri, int32_t* match12) {
  FeatureVector a{node1, start1, feat1, nn1}, b{node2, start2, feat2, nn2};
  return search_by_bow(kf_kf != 0, a, n1, desc1, angle1, usable1, b, n2, desc2, angle2, usable2, nnratio, checkOri != 0, match12);
}

int orc_search_for_triangulation(const uint32_t* node1, const int32_t* start1, const int32_t* feat1, int nn1, const KeyPoint* kp1, int n1,
                                 const uint8_t* desc1, const uint8_t* has_mp1, const uint32_t* node2, const int32_t* start2,
                                 const int32_t* feat2, int nn2, const KeyPoint* kp2, int n2, const uint8_t* desc2, const uint8_t* has_mp2,
                                 const float* F12, const float* sigma2, int checkOri, int32_t* match12) {
  FeatureVector a{node1, start1, feat1, nn1}, b{node2, start2, feat2, nn2};
  return search_for_triangulation(a, kp1, n1, desc1, has_mp1, b, kp2, n2, desc2, has_mp2, F12, sigma2, checkOri != 0, match12);
}

void orc_fuse_search(const KeyPoint* kps, int n, const uint8_t* kfdesc, int minX, int minY, int maxX, int maxY, int nmp, const float* u,
                     const float* v, const int32_t* level, const uint8_t* valid, const uint8_t* mpdesc, const float* scaleFactors, float th,
                     int32_t* best_idx, int32_t* best_dist) {
  FrameGrid g;
  g.build(kps, n, minX, minY, maxX, maxY);
  fuse_search(g, kfdesc, nmp, u, v, level, valid, mpdesc, scaleFactors, th, best_idx, best_dist);
}

// mode 0 = isInFrustum, 1 = SearchByProjection(F, KF) prologue, 2 = Fuse prologue; cam = 23 floats in Camera order
void orc_project_points(int mode, const float* cam, int n, const float* xyz, const float* normal, const float* minD, const float* maxD,
                        const uint8_t* usable, const float* scaleFactors, int nlevels, float scaleFactor, float cosLimit, uint8_t* valid,
                        float* u, float* v, int32_t* level, float* viewcos) {
  Camera C;
  memcpy(&C, cam, sizeof(Camera));
  for (int i = 0; i < n; ++i) {
    valid[i] = 0, u[i] = v[i] = 0.f, level[i] = 0;
    if (viewcos) viewcos[i] = 0.f;
    if (usable && !usable[i]) continue;
    float uu = 0, vv = 0, vc = 0;
    int lv = 0;
    bool ok;
    if (mode == 0)
      ok = is_in_frustum(C, xyz + 3 * i, normal + 3 * i, minD[i], maxD[i], cosLimit, scaleFactor, nlevels, &uu, &vv, &lv, &vc);
    else if (mode == 1)
      ok = project_kf_reloc(C, xyz + 3 * i, minD[i], scaleFactors, nlevels, &uu, &vv, &lv);
    else
      ok = project_fuse(C, xyz + 3 * i, normal + 3 * i, minD[i], maxD[i], scaleFactors, nlevels, &uu, &vv, &lv);
    if (!ok) continue;
    valid[i] = 1, u[i] = uu, v[i] = vv, level[i] = lv;
    if (viewcos) viewcos[i] = vc;
  }
}

void orc_sim3_decompose(const float* Scw, int row_stride, float* Rcw, float* tcw, float* Ow) { sim3_decompose(Scw, row_stride, Rcw, tcw, Ow); }
void orc_sim3_relative(float s12, const float* R12, const float* t12, float* sR12, float* sR21, float* t21) {
  sim3_relative(s12, R12, t12, sR12, sR21, t21);
}
// cam = 23 floats in Camera order (only the intrinsics and bounds are read)
void orc_project_sim3(const float* Ra, const float* ta, const float* sR, const float* t, const float* cam, int n, const float* xyz,
                      const float* minD, const float* maxD, const uint8_t* usable, const float* scaleFactors, int nlevels, uint8_t* valid,
                      float* u, float* v, int32_t* level) {
  Camera C;
  memcpy(&C, cam, sizeof(Camera));
  for (int i = 0; i < n; ++i) {
    valid[i] = 0, u[i] = v[i] = 0.f, level[i] = 0;
    if (usable && !usable[i]) continue;
    float uu, vv;
    int lv;
    if (!project_sim3(Ra, ta, sR, t, C, xyz + 3 * i, minD[i], maxD[i], scaleFactors, nlevels, &uu, &vv, &lv)) continue;
    valid[i] = 1, u[i] = uu, v[i] = vv, level[i] = lv;
  }
}
int orc_search_by_projection_sim3(const KeyPoint* kps, int n, const uint8_t* kfdesc, int minX, int minY, int maxX, int maxY, int32_t* matched,
                                  int nmp, const float* u, const float* v, const int32_t* level, const uint8_t* valid, const uint8_t* mpdesc,
                                  const float* scaleFactors, int th) {
  FrameGrid g;
  g.build(kps, n, minX, minY, maxX, maxY);
  return search_by_projection_sim3(g, kfdesc, matched, nmp, u, v, level, valid, mpdesc, scaleFactors, th);
}
int orc_search_by_sim3(const KeyPoint* kp1, int n1, const uint8_t* desc1, const int32_t* bounds1, const KeyPoint* kp2, int n2,
                       const uint8_t* desc2, const int32_t* bounds2, const float* u12, const float* v12, const int32_t* level12,
                       const uint8_t* valid12, const uint8_t* mpdesc1, const float* u21, const float* v21, const int32_t* level21,
                       const uint8_t* valid21, const uint8_t* mpdesc2, const float* sf1, const float* sf2, float th, int32_t* match12) {
  FrameGrid g1, g2;
  g1.build(kp1, n1, bounds1[0], bounds1[1], bounds1[2], bounds1[3]);
  g2.build(kp2, n2, bounds2[0], bounds2[1], bounds2[2], bounds2[3]);
  return search_by_sim3(g1, desc1, n1, g2, desc2, n2, u12, v12, level12, valid12, mpdesc1, u21, v21, level21, valid21, mpdesc2, sf1, sf2, th,
                        match12);
}

int orc_bow_transform(int n_nodes, const int32_t* child_start, const int32_t* children, const uint8_t* descriptor, const int32_t* word_id,
                      const double* weight, int L, int weighting, int normalize, const uint8_t* features, int n, int levelsup,
                      int32_t* out_word, double* out_weight, int32_t* out_node, uint32_t* bow_id, double* bow_value, int* n_bow,
                      uint32_t* fv_node, int32_t* fv_start, int32_t* fv_feat, int* n_fv) {
  Vocabulary V{n_nodes, child_start, children, descriptor, word_id, weight, L, weighting, normalize};
  for (int i = 0; i < n; ++i) {
    int id, nid;
    double w;
    bow_transform_one(V, features + (size_t)i * 32, levelsup, &id, &w, &nid);
    out_word[i] = id, out_weight[i] = w, out_node[i] = nid;
  }
  std::vector<std::pair<uint32_t, double>> bow;
  std::vector<std::pair<uint32_t, std::vector<uint32_t>>> fv;
  bow_transform(V, features, n, levelsup, bow, fv);
  for (size_t k = 0; k < bow.size(); ++k) bow_id[k] = bow[k].first, bow_value[k] = bow[k].second;
  *n_bow = (int)bow.size();
  int off = 0;
  for (size_t j = 0; j < fv.size(); ++j) {
    fv_node[j] = fv[j].first;
    fv_start[j] = off;
    for (uint32_t f : fv[j].second) fv_feat[off++] = (int32_t)f;
  }
  fv_start[fv.size()] = off;
  *n_fv = (int)fv.size();
  return 0;
}

void orc_clahe(const uint8_t* img, int w, int h, long stride, double clip, int tx, int ty, uint8_t* dst, long dstep) {
  clahe_apply(View{const_cast<uint8_t*>(img), w, h, (ptrdiff_t)stride}, clip, tx, ty, dst, (ptrdiff_t)dstep);
}

void orc_haloc_hash(const float* r, int num_proj, int r_stride, const uint8_t* desc, int n, float* hash) { haloc_hash(r, num_proj, r_stride, desc, n, hash); }

// KLT: pyramids are opaque handles
void* orc_klt_pyramid(const uint8_t* img, int w, int h, long stride, int win_w, int win_h, int maxLevel) {
  KltPyramid* p = new KltPyramid();
  p->build(img, w, h, (ptrdiff_t)stride, win_w, win_h, maxLevel);
  return p;
}
void orc_klt_pyramid_free(void* p) { delete static_cast<KltPyramid*>(p); }
int orc_klt_levels(void* p) { return (int)static_cast<KltPyramid*>(p)->levels.size(); }
void orc_klt_level_dims(void* p, int l, int* w, int* h) {
  KltPyramid* P = static_cast<KltPyramid*>(p);
  *w = P->levels[l].w, *h = P->levels[l].h;
}
// copies level l without its border: image (w x h u8) and derivatives (w x h x 2 int16)
void orc_klt_level(void* p, int l, uint8_t* img, int16_t* deriv) {
  KltPyramid* P = static_cast<KltPyramid*>(p);
  const KltPyramid::Level& L = P->levels[l];
  for (int y = 0; y < L.h; ++y) {
    memcpy(img + (size_t)y * L.w, L.img.data() + (size_t)(y + P->by) * L.istep + P->bx, L.w);
    memcpy(deriv + (size_t)y * L.w * 2, L.deriv.data() + (size_t)(y + P->by) * L.dstep + 2 * P->bx, (size_t)L.w * 4);
  }
}
void orc_klt_track(void* p0, void* p1, const float* prevPts, float* nextPts, int n, int win_w, int win_h, int maxLevel, int maxCount, double eps,
                   double minEig, uint8_t* status, float* err) {
  klt_track(*static_cast<KltPyramid*>(p0), *static_cast<KltPyramid*>(p1), prevPts, nextPts, n, win_w, win_h, maxLevel, maxCount, eps, minEig, status,
            err);
}
void orc_undistort_points(const float* pts, int n, float fx, float fy, float cx, float cy, const float* dist, int n_dist, int fisheye, float* out) {
  undistort_points(pts, n, fx, fy, cx, cy, dist, n_dist, fisheye != 0, out);
}
// sum_mode / margin: see klt_oracle.cpp
void orc_klt_track_ex(void* p0, void* p1, const float* prevPts, float* nextPts, int n, int win_w, int win_h, int maxLevel, int maxCount, double eps,
                      double minEig, uint8_t* status, float* err, int sum_mode, float* margin) {
  klt_track(*static_cast<KltPyramid*>(p0), *static_cast<KltPyramid*>(p1), prevPts, nextPts, n, win_w, win_h, maxLevel, maxCount, eps, minEig, status,
            err, sum_mode, margin);
}

void orc_compute_three_maxima(const int* sizes, int L, int* ind) {
  int a = -1, b = -1, c = -1;
  compute_three_maxima(sizes, L, a, b, c);
  ind[0] = a, ind[1] = b, ind[2] = c;
}

}  // extern "C"
