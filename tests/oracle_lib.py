"""ctypes front-end of the CPU oracle (oracle/liborb_oracle.so).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liborb_oracle.so")

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp", ".inc"))]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liborb_oracle.so"])
    return LIB


class Oracle:
    def __init__(self, lib_path=None):
        """lib_path: another build of the same sources (bench.py's -march=native row); default = liborb_oracle.so, built on demand."""
        self.L = L = ctypes.CDLL(lib_path or build_oracle())
        vp, ci, cf, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long
        L.orc_extractor_create.restype = vp
        L.orc_extractor_create.argtypes = [ci, cf, ci, ci]
        L.orc_extractor_destroy.argtypes = [vp]
        L.orc_extractor_tables.argtypes = [vp, vp, vp, vp, vp]
        L.orc_extract.argtypes = [vp, vp, ci, ci, cl, vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, ci]
        L.orc_level_dims.argtypes = [vp, ci, vp, vp]
        L.orc_level_plane.argtypes = [vp, ci, ci, vp]
        L.orc_level_candidates.argtypes = [vp, ci, vp, ci]
        L.orc_level_keypoints.argtypes = [vp, ci, vp, ci]
        L.orc_border101.argtypes = [vp, ci, ci, cl, vp, ci]
        L.orc_resize_linear.argtypes = [vp, ci, ci, cl, vp, ci, ci]
        L.orc_fast.argtypes = [vp, ci, ci, cl, ci, ci, vp, ci]
        L.orc_fast_bruteforce.argtypes = [vp, ci, ci, cl, ci, ci, vp, ci]
        L.orc_gauss_taps.argtypes = [vp]
        L.orc_gauss7_padded.argtypes = [vp, ci, ci, ci]
        L.orc_gauss7_padded_ex.argtypes = [vp, ci, ci, ci, ci]
        L.orc_extractor_set_blur_rounding.argtypes = [vp, ci]
        L.orc_extractor_pattern.argtypes = [vp, vp]
        L.orc_fast_atan2.restype = cf
        L.orc_fast_atan2.argtypes = [cf, cf]
        L.orc_ic_angle.restype = cf
        L.orc_ic_angle.argtypes = [vp, vp, ci, ci, ci, cf, cf]
        L.orc_descriptor.argtypes = [vp, vp, ci, ci, ci, cf, cf, cf, vp]
        L.orc_sincosf.argtypes = [cf, vp, vp]
        L.orc_octree.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp, ci]
        L.orc_grider_fast.argtypes = [vp, ci, ci, cl, ci, ci, ci, ci, ci, vp, ci]
        L.orc_descriptor_distance.argtypes = [vp, vp]
        L.orc_knn2.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp]
        L.orc_distinctive_descriptor.argtypes = [vp, ci, vp]
        L.orc_features_in_area.argtypes = [vp, ci, ci, ci, ci, ci, cf, cf, cf, ci, ci, vp, ci]
        L.orc_search_by_projection.argtypes = [vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, cf, cf]
        L.orc_window_search.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, cf, ci, vp]
        L.orc_search_by_projection_frames.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, ci, cf]
        L.orc_search_for_initialization.argtypes = [vp, ci, vp, vp, ci, vp, ci, ci, ci, ci, vp, vp, ci, cf, ci]
        L.orc_search_by_projection_last.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, cf, ci]
        L.orc_search_by_projection_kf.argtypes = [vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci]
        L.orc_search_by_bow.argtypes = [ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, cf, ci, vp]
        L.orc_search_for_triangulation.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, vp]
        L.orc_fuse_search.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, cf, vp, vp]
        L.orc_compute_three_maxima.argtypes = [vp, ci, vp]
        L.orc_klt_pyramid.restype = vp
        L.orc_klt_pyramid.argtypes = [vp, ci, ci, cl, ci, ci, ci]
        L.orc_klt_pyramid_free.argtypes = [vp]
        L.orc_klt_levels.argtypes = [vp]
        L.orc_klt_level_dims.argtypes = [vp, ci, vp, vp]
        L.orc_klt_level.argtypes = [vp, ci, vp, vp]
        L.orc_klt_track.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ctypes.c_double, ctypes.c_double, vp, vp]
        L.orc_undistort_points.argtypes = [vp, ci, cf, cf, cf, cf, vp, ci, ci, vp]
        L.orc_klt_track_ex.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ctypes.c_double, ctypes.c_double, vp, vp, ci, vp]
        L.orc_haloc_hash.argtypes = [vp, ci, ci, vp, ci, vp]
        L.orc_clahe.argtypes = [vp, ci, ci, cl, ctypes.c_double, ci, ci, vp, cl]
        L.orc_bow_transform.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_project_points.argtypes = [ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp, vp, vp, vp, vp]
        L.orc_sim3_decompose.argtypes = [vp, ci, vp, vp, vp]
        L.orc_sim3_relative.argtypes = [cf, vp, vp, vp, vp, vp]
        L.orc_project_sim3.argtypes = [vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp]
        L.orc_search_by_projection_sim3.argtypes = [vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci]
        L.orc_search_by_sim3.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, vp]

    # ---- extractor ----
    def extractor(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, fastTh=20):
        return OracleExtractor(self, nfeatures, scaleFactor, nlevels, fastTh)

    # ---- primitives ----
    def border101(self, img, pad=16):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((h + 2 * pad, w + 2 * pad), np.uint8)
        self.L.orc_border101(img.ctypes.data, w, h, img.strides[0], out.ctypes.data, pad)
        return out

    def resize_linear(self, img, dw, dh):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((dh, dw), np.uint8)
        self.L.orc_resize_linear(img.ctypes.data, w, h, img.strides[0], out.ctypes.data, dw, dh)
        return out

    def clahe(self, img, clip_limit=4.0, tiles=(12, 12)):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.empty_like(img)
        self.L.orc_clahe(img.ctypes.data, w, h, img.strides[0], float(clip_limit), int(tiles[0]), int(tiles[1]), out.ctypes.data, out.strides[0])
        return out

    def fast(self, img, threshold, nms=True, bruteforce=False):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros(w * h, KP)
        fn = self.L.orc_fast_bruteforce if bruteforce else self.L.orc_fast
        n = fn(img.ctypes.data, w, h, img.strides[0], threshold, 1 if nms else 0, out.ctypes.data, len(out))
        return out[:n].copy()

    def gauss_taps(self):
        t = np.zeros(7, np.int32)
        self.L.orc_gauss_taps(t.ctypes.data)
        return t

    def gauss7_padded_ex(self, plane, pad=16, rounding=0):
        """rounding: 0 = the generic column filter (half up), 1 = the x86-64 SSE2 contract (vector-body columns: exact ties to even)"""
        plane = np.ascontiguousarray(plane, np.uint8).copy()
        ph, pw = plane.shape
        self.L.orc_gauss7_padded_ex(plane.ctypes.data, pw - 2 * pad, ph - 2 * pad, pad, int(rounding))
        return plane

    def gauss7_padded(self, plane, pad=16):
        plane = np.array(plane, np.uint8, copy=True, order="C")
        ph, pw = plane.shape
        self.L.orc_gauss7_padded(plane.ctypes.data, pw - 2 * pad, ph - 2 * pad, pad)
        return plane

    def fast_atan2(self, y, x):
        return float(self.L.orc_fast_atan2(float(y), float(x)))

    def sincosf(self, a):
        s, c = ctypes.c_float(), ctypes.c_float()
        self.L.orc_sincosf(float(a), ctypes.byref(s), ctypes.byref(c))
        return s.value, c.value

    def descriptor_distance(self, a, b):
        a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
        return self.L.orc_descriptor_distance(a.ctypes.data, b.ctypes.data)

    def knn2(self, q, t, mask=None):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq = len(q)
        o = [np.zeros(nq, np.int32) for _ in range(4)]
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.L.orc_knn2(q.ctypes.data, nq, t.ctypes.data, len(t), None if m is None else m.ctypes.data, *[a.ctypes.data for a in o])
        return o  # idx0, d0, idx1, d1

    def distinctive_descriptor(self, desc):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        med = ctypes.c_int()
        idx = self.L.orc_distinctive_descriptor(desc.ctypes.data, len(desc), ctypes.byref(med))
        return idx, med.value

    def features_in_area(self, kps, bounds, x, y, r, min_level, max_level):
        kps = np.ascontiguousarray(kps, KP)
        out = np.zeros(max(len(kps), 1), np.int32)
        n = self.L.orc_features_in_area(kps.ctypes.data, len(kps), *[int(b) for b in bounds], float(x), float(y), float(r), min_level, max_level,
                                        out.ctypes.data, len(out))
        return out[:n].copy()

    def search_by_projection(self, kps, desc, bounds, assigned, proj_x, proj_y, level, view_cos, in_view, mp_desc, scale_factors, th, nnratio):
        kps = np.ascontiguousarray(kps, KP)
        desc = np.ascontiguousarray(desc, np.uint8)
        a = [np.ascontiguousarray(proj_x, np.float32), np.ascontiguousarray(proj_y, np.float32), np.ascontiguousarray(level, np.int32),
             np.ascontiguousarray(view_cos, np.float32), np.ascontiguousarray(in_view, np.uint8), np.ascontiguousarray(mp_desc, np.uint8),
             np.ascontiguousarray(scale_factors, np.float32)]
        assert assigned.dtype == np.int32
        return self.L.orc_search_by_projection(kps.ctypes.data, len(kps), desc.ctypes.data, *[int(b) for b in bounds], assigned.ctypes.data,
                                               len(a[0]), *[v.ctypes.data for v in a], float(th), float(nnratio))

    def search_by_projection_kf(self, kps, desc, bounds, assigned, u, v, level, valid, mp_desc, kf_angle, scale_factors, th, orb_dist, check_ori):
        kps = np.ascontiguousarray(kps, KP)
        desc = np.ascontiguousarray(desc, np.uint8)
        a = [np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32), np.ascontiguousarray(level, np.int32),
             np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(mp_desc, np.uint8), np.ascontiguousarray(kf_angle, np.float32),
             np.ascontiguousarray(scale_factors, np.float32)]
        assert assigned.dtype == np.int32
        return self.L.orc_search_by_projection_kf(kps.ctypes.data, len(kps), desc.ctypes.data, *[int(b) for b in bounds], assigned.ctypes.data,
                                                  len(a[0]), *[x.ctypes.data for x in a], float(th), int(orb_dist), 1 if check_ori else 0)

    @staticmethod
    def _fv(groups):
        nodes = sorted(groups)
        node = np.asarray(nodes, np.uint32)
        start = np.zeros(len(nodes) + 1, np.int32)
        feats = []
        for j, k in enumerate(nodes):
            feats.extend(int(x) for x in groups[k])
            start[j + 1] = len(feats)
        return node, start, np.asarray(feats if feats else [0], np.int32), len(nodes)

    def search_by_bow(self, kf_kf, groups1, desc1, angle1, usable1, groups2, desc2, angle2, usable2, nnratio, check_ori):
        """groups*: {node id: [feature indices]} (a DBoW2::FeatureVector).  Returns (match12, nmatches)."""
        f1, f2 = self._fv(groups1), self._fv(groups2)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        a1, a2 = np.ascontiguousarray(angle1, np.float32), np.ascontiguousarray(angle2, np.float32)
        u1 = np.ascontiguousarray(usable1, np.uint8)
        u2 = None if usable2 is None else np.ascontiguousarray(usable2, np.uint8)
        match = np.full(len(d1), -1, np.int32)
        n = self.L.orc_search_by_bow(1 if kf_kf else 0, f1[0].ctypes.data, f1[1].ctypes.data, f1[2].ctypes.data, f1[3], len(d1), d1.ctypes.data,
                                     a1.ctypes.data, u1.ctypes.data, f2[0].ctypes.data, f2[1].ctypes.data, f2[2].ctypes.data, f2[3], len(d2),
                                     d2.ctypes.data, a2.ctypes.data, None if u2 is None else u2.ctypes.data, float(nnratio), 1 if check_ori else 0,
                                     match.ctypes.data)
        return match, n

    def search_for_triangulation(self, groups1, kp1, desc1, has_mp1, groups2, kp2, desc2, has_mp2, F12, sigma2, check_ori):
        f1, f2 = self._fv(groups1), self._fv(groups2)
        kp1, kp2 = np.ascontiguousarray(kp1, KP), np.ascontiguousarray(kp2, KP)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
        f, s2 = np.ascontiguousarray(F12, np.float32).reshape(9), np.ascontiguousarray(sigma2, np.float32)
        match = np.full(len(kp1), -1, np.int32)
        n = self.L.orc_search_for_triangulation(f1[0].ctypes.data, f1[1].ctypes.data, f1[2].ctypes.data, f1[3], kp1.ctypes.data, len(kp1), d1.ctypes.data,
                                                h1.ctypes.data, f2[0].ctypes.data, f2[1].ctypes.data, f2[2].ctypes.data, f2[3], kp2.ctypes.data, len(kp2),
                                                d2.ctypes.data, h2.ctypes.data, f.ctypes.data, s2.ctypes.data, 1 if check_ori else 0, match.ctypes.data)
        return match, n

    def fuse_search(self, kps, desc, bounds, u, v, level, valid, mp_desc, scale_factors, th):
        kps = np.ascontiguousarray(kps, KP)
        desc = np.ascontiguousarray(desc, np.uint8)
        a = [np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32), np.ascontiguousarray(level, np.int32),
             np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(mp_desc, np.uint8), np.ascontiguousarray(scale_factors, np.float32)]
        bi, bd = np.full(len(a[0]), -1, np.int32), np.full(len(a[0]), -1, np.int32)
        self.L.orc_fuse_search(kps.ctypes.data, len(kps), desc.ctypes.data, *[int(b) for b in bounds], len(a[0]), *[x.ctypes.data for x in a], float(th),
                               bi.ctypes.data, bd.ctypes.data)
        return bi, bd

    # ---- the four ORBmatcher members without a caller in the reference (src/ORBmatcher.cc:409-713, :1507-1620) ----
    def window_search(self, kp1, desc1, has_mp1, kp2, desc2, bounds2, window, min_level, max_level, nnratio, check_ori):
        kp1, kp2 = np.ascontiguousarray(kp1, KP), np.ascontiguousarray(kp2, KP)
        d1, d2, hm = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8), np.ascontiguousarray(has_mp1, np.uint8)
        m21 = np.full(len(kp2), -1, np.int32)
        n = self.L.orc_window_search(kp1.ctypes.data, len(kp1), d1.ctypes.data, hm.ctypes.data, kp2.ctypes.data, len(kp2), d2.ctypes.data,
                                     *[int(b) for b in bounds2], int(window), int(min_level), int(max_level), float(nnratio), 1 if check_ori else 0,
                                     m21.ctypes.data)
        return m21, n

    def search_by_projection_frames(self, kp1, desc1, usable1, xyz1, cam, kp2, desc2, assigned2, window, nnratio):
        kp1, kp2 = np.ascontiguousarray(kp1, KP), np.ascontiguousarray(kp2, KP)
        d1, d2, us = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8), np.ascontiguousarray(usable1, np.uint8)
        x, cam = np.ascontiguousarray(xyz1, np.float32), np.ascontiguousarray(cam, np.float32)
        assert assigned2.dtype == np.int32
        return self.L.orc_search_by_projection_frames(kp1.ctypes.data, len(kp1), d1.ctypes.data, us.ctypes.data, x.ctypes.data, cam.ctypes.data,
                                                      kp2.ctypes.data, len(kp2), d2.ctypes.data, assigned2.ctypes.data, int(window), float(nnratio))

    def search_for_initialization(self, kp1, desc1, kp2, desc2, bounds2, prev_matched, window, nnratio, check_ori):
        kp1, kp2 = np.ascontiguousarray(kp1, KP), np.ascontiguousarray(kp2, KP)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        assert prev_matched.dtype == np.float32 and prev_matched.shape == (len(kp1), 2) and prev_matched.flags.c_contiguous
        m12 = np.full(len(kp1), -1, np.int32)
        n = self.L.orc_search_for_initialization(kp1.ctypes.data, len(kp1), d1.ctypes.data, kp2.ctypes.data, len(kp2), d2.ctypes.data,
                                                 *[int(b) for b in bounds2], prev_matched.ctypes.data, m12.ctypes.data, int(window), float(nnratio),
                                                 1 if check_ori else 0)
        return m12, n

    def search_by_projection_last(self, cam, kps, desc, assigned, usable_last, xyz_last, octave_last, angle_last, desc_last, scale_factors, th, check_ori):
        kps, cam = np.ascontiguousarray(kps, KP), np.ascontiguousarray(cam, np.float32)
        a = [np.ascontiguousarray(usable_last, np.uint8), np.ascontiguousarray(xyz_last, np.float32), np.ascontiguousarray(octave_last, np.int32),
             np.ascontiguousarray(angle_last, np.float32), np.ascontiguousarray(desc_last, np.uint8), np.ascontiguousarray(scale_factors, np.float32)]
        d = np.ascontiguousarray(desc, np.uint8)
        assert assigned.dtype == np.int32
        return self.L.orc_search_by_projection_last(cam.ctypes.data, kps.ctypes.data, len(kps), d.ctypes.data, assigned.ctypes.data, len(a[0]),
                                                    *[x.ctypes.data for x in a], float(th), 1 if check_ori else 0)

    def project_points(self, mode, cam, xyz, normal, min_distance, max_distance, usable, scale_factors, scale_factor=1.2, cos_limit=0.5):
        """cam: 23 floats = Rcw[9], tcw[3], Ow[3], fx, fy, cx, cy, minX, maxX, minY, maxY.  Returns (valid, u, v, level, view_cos)."""
        cam = np.ascontiguousarray(cam, np.float32)
        assert cam.shape == (23,)
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = len(xyz)
        nrm = np.ascontiguousarray(normal if normal is not None else np.zeros((n, 3)), np.float32).reshape(-1, 3)
        mn, mx = np.ascontiguousarray(min_distance, np.float32), np.ascontiguousarray(max_distance, np.float32)
        us = None if usable is None else np.ascontiguousarray(usable, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        valid, u, v = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32)
        level, vc = np.zeros(n, np.int32), np.zeros(n, np.float32)
        self.L.orc_project_points(int(mode), cam.ctypes.data, n, xyz.ctypes.data, nrm.ctypes.data, mn.ctypes.data, mx.ctypes.data,
                                  None if us is None else us.ctypes.data, sf.ctypes.data, len(sf), float(scale_factor), float(cos_limit),
                                  valid.ctypes.data, u.ctypes.data, v.ctypes.data, level.ctypes.data, vc.ctypes.data)
        return valid, u, v, level, vc

    def sim3_decompose(self, scw):
        scw = np.ascontiguousarray(scw, np.float32)
        r, t, o = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.L.orc_sim3_decompose(scw.ctypes.data, 4, r.ctypes.data, t.ctypes.data, o.ctypes.data)
        return r, t, o

    def sim3_relative(self, s12, r12, t12):
        r12, t12 = np.ascontiguousarray(r12, np.float32).reshape(9), np.ascontiguousarray(t12, np.float32).reshape(3)
        a, b, c = np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.zeros(3, np.float32)
        self.L.orc_sim3_relative(ctypes.c_float(s12), r12.ctypes.data, t12.ctypes.data, a.ctypes.data, b.ctypes.data, c.ctypes.data)
        return a, b, c

    def project_sim3(self, r_own, t_own, s_r, t, cam, xyz, min_distance, max_distance, usable, scale_factors):
        """min_distance / max_distance = mfMinDistance / mfMaxDistance; returns (valid, u, v, level)."""
        ro, to = np.ascontiguousarray(r_own, np.float32).reshape(9), np.ascontiguousarray(t_own, np.float32).reshape(3)
        sr, tt = np.ascontiguousarray(s_r, np.float32).reshape(9), np.ascontiguousarray(t, np.float32).reshape(3)
        cam = np.ascontiguousarray(cam, np.float32)
        assert cam.shape == (23,)
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = len(xyz)
        mn = (np.float32(0.8) * np.ascontiguousarray(min_distance, np.float32)).astype(np.float32)   # MapPoint::GetMinDistanceInvariance
        mx = (np.float32(1.2) * np.ascontiguousarray(max_distance, np.float32)).astype(np.float32)
        us = None if usable is None else np.ascontiguousarray(usable, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        valid, u, v, level = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        self.L.orc_project_sim3(ro.ctypes.data, to.ctypes.data, sr.ctypes.data, tt.ctypes.data, cam.ctypes.data, n, xyz.ctypes.data, mn.ctypes.data,
                                mx.ctypes.data, None if us is None else us.ctypes.data, sf.ctypes.data, len(sf), valid.ctypes.data, u.ctypes.data,
                                v.ctypes.data, level.ctypes.data)
        return valid, u, v, level

    def search_by_projection_sim3(self, kps, desc, bounds, matched, u, v, level, valid, mp_desc, scale_factors, th):
        kps, desc = np.ascontiguousarray(kps, KP), np.ascontiguousarray(desc, np.uint8)
        assert matched.dtype == np.int32
        a = [np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32), np.ascontiguousarray(level, np.int32),
             np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(mp_desc, np.uint8), np.ascontiguousarray(scale_factors, np.float32)]
        return self.L.orc_search_by_projection_sim3(kps.ctypes.data, len(kps), desc.ctypes.data, *[int(b) for b in bounds], matched.ctypes.data,
                                                    len(a[0]), *[x.ctypes.data for x in a], int(th))

    def search_by_sim3(self, kp1, desc1, bounds1, kp2, desc2, bounds2, proj12, mp_desc1, proj21, mp_desc2, sf1, sf2, th):
        kp1, kp2 = np.ascontiguousarray(kp1, KP), np.ascontiguousarray(kp2, KP)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        b1, b2 = np.ascontiguousarray(bounds1, np.int32), np.ascontiguousarray(bounds2, np.int32)

        def unpack(p):
            return [np.ascontiguousarray(p[1], np.float32), np.ascontiguousarray(p[2], np.float32), np.ascontiguousarray(p[3], np.int32),
                    np.ascontiguousarray(p[0], np.uint8)]
        a12, a21 = unpack(proj12), unpack(proj21)
        m1, m2 = np.ascontiguousarray(mp_desc1, np.uint8), np.ascontiguousarray(mp_desc2, np.uint8)
        s1, s2 = np.ascontiguousarray(sf1, np.float32), np.ascontiguousarray(sf2, np.float32)
        match12 = np.full(len(kp1), -1, np.int32)
        n = self.L.orc_search_by_sim3(kp1.ctypes.data, len(kp1), d1.ctypes.data, b1.ctypes.data, kp2.ctypes.data, len(kp2), d2.ctypes.data,
                                      b2.ctypes.data, *[x.ctypes.data for x in a12], m1.ctypes.data, *[x.ctypes.data for x in a21], m2.ctypes.data,
                                      s1.ctypes.data, s2.ctypes.data, ctypes.c_float(th), match12.ctypes.data)
        return match12, n

    def bow_transform(self, voc, desc, levelsup=4):
        """voc = dict(child_start, children, descriptor, word_id, weight, L, weighting, normalize).
        Returns (word_id, weight, node_id, (bow ids, bow values), {node: [features]})."""
        cs, ch = np.ascontiguousarray(voc["child_start"], np.int32), np.ascontiguousarray(voc["children"], np.int32)
        de, wi = np.ascontiguousarray(voc["descriptor"], np.uint8), np.ascontiguousarray(voc["word_id"], np.int32)
        we = np.ascontiguousarray(voc["weight"], np.float64)
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        wid, nid, ww = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float64)
        bid, bval = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        fnode, fstart, ffeat = np.zeros(max(n, 1), np.uint32), np.zeros(n + 2, np.int32), np.zeros(max(n, 1), np.int32)
        nb, nf = ctypes.c_int(), ctypes.c_int()
        self.L.orc_bow_transform(len(cs) - 1, cs.ctypes.data, ch.ctypes.data, de.ctypes.data, wi.ctypes.data, we.ctypes.data, int(voc["L"]),
                                 int(voc["weighting"]), int(voc["normalize"]), desc.ctypes.data, n, int(levelsup), wid.ctypes.data, ww.ctypes.data,
                                 nid.ctypes.data, bid.ctypes.data, bval.ctypes.data, ctypes.byref(nb), fnode.ctypes.data, fstart.ctypes.data,
                                 ffeat.ctypes.data, ctypes.byref(nf))
        groups = {int(fnode[j]): [int(x) for x in ffeat[fstart[j]:fstart[j + 1]]] for j in range(nf.value)}
        return wid, ww, nid, (bid[:nb.value].copy(), bval[:nb.value].copy()), groups

    def haloc_hash(self, proj, desc):
        proj, desc = np.ascontiguousarray(proj, np.float32), np.ascontiguousarray(desc, np.uint8)
        out = np.zeros(proj.shape[0] * 32, np.float32)
        self.L.orc_haloc_hash(proj.ctypes.data, proj.shape[0], proj.shape[1], desc.ctypes.data, len(desc), out.ctypes.data)
        return out

    def klt_pyramid(self, img, win=(21, 21), max_level=5):
        return OracleKltPyramid(self, img, win, max_level)

    def klt_track(self, p0, p1, prev_pts, next_pts0=None, win=(21, 21), max_level=5, max_count=30, epsilon=0.01, min_eig=1e-4):
        a = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        b = a.copy() if next_pts0 is None else np.ascontiguousarray(next_pts0, np.float32).reshape(-1, 2).copy()
        st, er = np.zeros(len(a), np.uint8), np.zeros(len(a), np.float32)
        self.L.orc_klt_track(p0.h, p1.h, a.ctypes.data, b.ctypes.data, len(a), win[0], win[1], max_level, max_count, float(epsilon), float(min_eig),
                             st.ctypes.data, er.ctypes.data)
        return b, st, er

    def klt_track_ex(self, p0, p1, prev_pts, next_pts0=None, win=(21, 21), max_level=5, max_count=30, epsilon=0.01, min_eig=1e-4, sum_mode=0):
        """klt_track with the association order of the float window sums selectable (0 = raster, OpenCV's generic loop; 1 = the HIP kernel's
        lane / butterfly order) and the per-point decision margin.  -> (next, status, err, margin)"""
        a = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        b = a.copy() if next_pts0 is None else np.ascontiguousarray(next_pts0, np.float32).reshape(-1, 2).copy()
        st, er, mg = np.zeros(len(a), np.uint8), np.zeros(len(a), np.float32), np.zeros(len(a), np.float32)
        self.L.orc_klt_track_ex(p0.h, p1.h, a.ctypes.data, b.ctypes.data, len(a), win[0], win[1], max_level, max_count, float(epsilon), float(min_eig),
                                st.ctypes.data, er.ctypes.data, int(sum_mode), mg.ctypes.data)
        return b, st, er, mg

    def undistort_points(self, pts, fx, fy, cx, cy, dist, fisheye=False):
        p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        d = np.ascontiguousarray(dist, np.float32)
        out = np.zeros_like(p)
        self.L.orc_undistort_points(p.ctypes.data, len(p), fx, fy, cx, cy, d.ctypes.data, len(d), 1 if fisheye else 0, out.ctypes.data)
        return out

    def compute_three_maxima(self, sizes):
        s = np.ascontiguousarray(sizes, np.int32)
        out = np.zeros(3, np.int32)
        self.L.orc_compute_three_maxima(s.ctypes.data, len(s), out.ctypes.data)
        return [int(x) for x in out]

    def grider_fast(self, img, num_features, grid_x, grid_y, threshold, nms=True):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros(w * h // 4 + 16, KP)
        n = self.L.orc_grider_fast(img.ctypes.data, w, h, img.strides[0], num_features, grid_x, grid_y, threshold, 1 if nms else 0,
                                   out.ctypes.data, len(out))
        return out[:n].copy()


class OracleKltPyramid:
    def __init__(self, o, img, win, max_level):
        self.L = o.L
        img = np.ascontiguousarray(img, np.uint8)
        self.h = self.L.orc_klt_pyramid(img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], win[0], win[1], max_level)
        self.levels = self.L.orc_klt_levels(self.h)

    def level(self, l):
        w, h = ctypes.c_int(), ctypes.c_int()
        self.L.orc_klt_level_dims(self.h, l, ctypes.byref(w), ctypes.byref(h))
        img, der = np.zeros((h.value, w.value), np.uint8), np.zeros((h.value, w.value, 2), np.int16)
        self.L.orc_klt_level(self.h, l, img.ctypes.data, der.ctypes.data)
        return img, der

    def __del__(self):
        try:
            self.L.orc_klt_pyramid_free(self.h)
        except Exception:
            pass


class OracleExtractor:
    def __init__(self, o, nfeatures, scaleFactor, nlevels, fastTh):
        self.o, self.L = o, o.L
        self.nlevels = nlevels
        self.h = self.L.orc_extractor_create(nfeatures, scaleFactor, nlevels, fastTh)
        self.scale = np.zeros(nlevels, np.float32)
        self.inv_scale = np.zeros(nlevels, np.float32)
        self.quota = np.zeros(nlevels, np.int32)
        self.umax = np.zeros(16, np.int32)
        self.L.orc_extractor_tables(self.h, self.scale.ctypes.data, self.inv_scale.ctypes.data, self.quota.ctypes.data, self.umax.ctypes.data)

    def __del__(self):
        try:
            self.L.orc_extractor_destroy(self.h)
        except Exception:
            pass

    def __call__(self, image, keypoints=None, grid_2d=None, min_px_dist=20, FullDetect=True, num_featsneeded=0, cap=20000):
        image = np.ascontiguousarray(image, np.uint8)
        h, w = image.shape
        n_in = 0 if keypoints is None else len(keypoints)
        kin = None if n_in == 0 else np.ascontiguousarray(keypoints, KP)
        rows = cols = 0
        if grid_2d is not None:
            assert grid_2d.dtype == np.int32 and grid_2d.flags.f_contiguous
            rows, cols = grid_2d.shape
        out_kp = np.zeros(cap, KP)
        out_desc = np.zeros((cap, 32), np.uint8)
        n = self.L.orc_extract(self.h, image.ctypes.data, w, h, image.strides[0], None if kin is None else kin.ctypes.data, n_in,
                               None if grid_2d is None else grid_2d.ctypes.data, rows, cols, int(min_px_dist), 1 if FullDetect else 0,
                               int(num_featsneeded), out_kp.ctypes.data, out_desc.ctypes.data, cap)
        assert n >= 0, "oracle capacity"
        return out_kp[:n].copy(), out_desc[:n].copy()

    def pattern(self):
        out = np.zeros(1024, np.int32)
        self.L.orc_extractor_pattern(self.h, out.ctypes.data)
        return out

    def set_blur_rounding(self, rounding):
        self.L.orc_extractor_set_blur_rounding(self.h, int(rounding))

    def level_dims(self, level):
        w, h = ctypes.c_int(), ctypes.c_int()
        self.L.orc_level_dims(self.h, level, ctypes.byref(w), ctypes.byref(h))
        return w.value, h.value

    def level_plane(self, level, blurred=False):
        w, h = self.level_dims(level)
        out = np.zeros((h + 32, w + 32), np.uint8)
        self.L.orc_level_plane(self.h, level, 1 if blurred else 0, out.ctypes.data)
        return out

    def level_candidates(self, level):
        n = self.L.orc_level_candidates(self.h, level, None, 0)
        out = np.zeros(max(n, 1), KP)
        self.L.orc_level_candidates(self.h, level, out.ctypes.data, n)
        return out[:n]

    def level_keypoints(self, level):
        n = self.L.orc_level_keypoints(self.h, level, None, 0)
        out = np.zeros(max(n, 1), KP)
        self.L.orc_level_keypoints(self.h, level, out.ctypes.data, n)
        return out[:n]

    def ic_angle(self, plane, x, y, pad=16):
        plane = np.ascontiguousarray(plane, np.uint8)
        ph, pw = plane.shape
        return float(self.L.orc_ic_angle(self.h, plane.ctypes.data, pw - 2 * pad, ph - 2 * pad, pad, float(x), float(y)))

    def descriptor(self, plane, x, y, angle_deg, pad=16):
        plane = np.ascontiguousarray(plane, np.uint8)
        ph, pw = plane.shape
        d = np.zeros(32, np.uint8)
        self.L.orc_descriptor(self.h, plane.ctypes.data, pw - 2 * pad, ph - 2 * pad, pad, float(x), float(y), float(angle_deg), d.ctypes.data)
        return d

    def octree(self, cand_xyr, W, H, N):
        """cand_xyr: (P,3) ints (x, y, response) relative to minBorder, in candidate order."""
        P = len(cand_xyr)
        kp = np.zeros(max(P, 1), KP)
        kp["x"][:P] = cand_xyr[:, 0]
        kp["y"][:P] = cand_xyr[:, 1]
        kp["response"][:P] = cand_xyr[:, 2]
        out = np.zeros(N + P + 8, KP)
        n = self.L.orc_octree(self.h, kp.ctypes.data, P, 13, 13 + W, 13, 13 + H, N, out.ctypes.data, len(out))
        r = out[:n]
        return np.stack([r["x"], r["y"], r["response"]], 1).astype(np.int64)
