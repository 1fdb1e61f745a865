#!/usr/bin/env python3
"""Developer soak: the matcher entry points against the oracle on random sizes / parameters.  Usage: soak_matcher.py [n] [seed]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rand_kps(rng, n, w, h, uvo):
    kp = np.zeros(n, uvo.KEYPOINT_DTYPE)
    kp["x"], kp["y"] = rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)
    kp["octave"] = rng.integers(0, 8, n)
    kp["angle"] = rng.uniform(0, 360, n)
    return kp


def noisy(rng, base, p):
    return np.packbits(np.unpackbits(base, axis=1) ^ (rng.random((len(base), 256)) < p), axis=1)


def groups(rng, n, nn):
    g = {}
    ids = rng.choice(np.arange(1, 5 * nn + 5), nn, replace=False)
    for i in rng.permutation(n):
        if rng.random() < 0.9:
            g.setdefault(int(ids[rng.integers(0, nn)]), []).append(int(i))
    return g


def main():
    n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    torch.zeros(1, device="cuda")
    uvo = importlib.import_module("u-vip-slam_amd")
    import oracle_lib
    o = oracle_lib.Oracle()
    rng = np.random.default_rng(seed)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    bad = 0
    for t in range(n_trials):
        w, h = int(rng.integers(100, 1300)), int(rng.integers(100, 800))
        n, M = int(rng.integers(0, 2500)), int(rng.integers(0, 6000))
        ratio, ori = float(rng.choice([0.6, 0.75, 0.9, 1.0])), bool(rng.integers(0, 2))
        m = uvo.ORBmatcher(ratio, ori, max_query=4096, max_map_points=8192)
        kp = rand_kps(rng, n, w, h, uvo)
        de = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        src = rng.integers(0, max(n, 1), M)
        mpd = noisy(rng, de[src], 0.07) if n else rng.integers(0, 256, (M, 32), dtype=np.uint8)
        rnd = rng.random(M) < 0.3
        mpd[rnd] = rng.integers(0, 256, (int(rnd.sum()), 32), dtype=np.uint8)
        u = (kp["x"][src] + rng.normal(0, 3, M)).astype(np.float32) if n else rng.uniform(0, w, M).astype(np.float32)
        v = (kp["y"][src] + rng.normal(0, 3, M)).astype(np.float32) if n else rng.uniform(0, h, M).astype(np.float32)
        lvl = rng.integers(0, 8, M).astype(np.int32)
        valid = (rng.random(M) < 0.85).astype(np.uint8)
        vc = np.where(rng.random(M) < 0.5, 0.999, 0.9).astype(np.float32)
        bounds = (0, 0, w, h)
        pre = np.full(n, -1, np.int32)
        if n:
            pre[rng.integers(0, n, n // 20)] = 999999
        what = []
        # SearchByProjection (frame, map points)
        a, b = pre.copy(), pre.copy()
        th = float(rng.choice([1.0, 3.0, 8.0]))
        na, nb = m.SearchByProjection(kp, de, bounds, a, u, v, lvl, vc, valid, mpd, sf, th), o.search_by_projection(kp, de, bounds, b, u, v, lvl, vc, valid, mpd, sf, th, ratio)
        if na != nb or not np.array_equal(a, b):
            what.append("sbp")
        # SearchByProjection (frame, key frame)
        a, b = pre.copy(), pre.copy()
        ang = rng.uniform(0, 360, M).astype(np.float32)
        od = int(rng.choice([50, 64, 100]))
        na, nb = m.SearchByProjectionKF(kp, de, bounds, a, u, v, lvl, valid, mpd, ang, sf, th * 3, od), o.search_by_projection_kf(kp, de, bounds, b, u, v, lvl, valid, mpd, ang, sf, th * 3, od, ori)
        if na != nb or not np.array_equal(a, b):
            what.append("sbp_kf")
        # Fuse core
        fa, fb = m.FuseSearch(kp, de, bounds, u, v, lvl, valid, mpd, sf, th), o.fuse_search(kp, de, bounds, u, v, lvl, valid, mpd, sf, th)
        if not (np.array_equal(fa[0], fb[0]) and np.array_equal(fa[1], fb[1])):
            what.append("fuse")
        # loop-closing forms: exclusive projection search, and (further down, with the second set) SearchBySim3
        a, b = pre.copy(), pre.copy()
        thi = int(rng.choice([3, 10]))
        na, nb = m.SearchByProjectionSim3(kp, de, bounds, a, u, v, lvl, valid, mpd, sf, thi), o.search_by_projection_sim3(kp, de, bounds, b, u, v, lvl, valid, mpd, sf, thi)
        if na != nb or not np.array_equal(a, b):
            what.append("sbp_sim3")
        # SearchReferencePointsInFrustum as one call: random pose, points scattered in front of the camera
        if M:
            ax = rng.normal(0, 0.1, 3)
            ang_ = np.linalg.norm(ax)
            kx = ax / ang_
            K_ = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
            R = (np.eye(3) + np.sin(ang_) * K_ + (1 - np.cos(ang_)) * K_ @ K_).astype(np.float32)
            tt = rng.normal(0, 0.3, 3).astype(np.float32)
            Ow = (-(R.T.astype(np.float64) @ tt.astype(np.float64))).astype(np.float32)
            fx, fy, cx, cy = 0.6 * w, 0.6 * w, w / 2, h / 2
            cam = uvo.CameraPose.make(R, tt, Ow, fx, fy, cx, cy, bounds)
            z = rng.uniform(1, 10, M)
            pc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
            xyz = ((pc - tt) @ R.astype(np.float64)).astype(np.float32)
            nrm = xyz - Ow + rng.normal(0, 2.0, (M, 3))
            nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
            dd = np.linalg.norm(xyz - Ow, axis=1)
            mxd = (dd * rng.uniform(0.6, 4.0, M)).astype(np.float32)
            mnd = (mxd / np.float32(rng.uniform(2.0, 5.0))).astype(np.float32)
            us = None if rng.random() < 0.5 else (rng.random(M) < 0.8).astype(np.uint8)
            a, b = pre.copy(), pre.copy()
            nm_a, iv = m.SearchPointsInFrustum(kp, de, a, cam, xyz, nrm, mnd, mxd, us, mpd, sf, 1.2, 0.5, th)
            pv, pu, pvv, pl, pvc = o.project_points(0, cam.as_array(), xyz, nrm, mnd, mxd, us, sf, 1.2, 0.5)
            nm_b = o.search_by_projection(kp, de, bounds, b, pu, pvv, pl, pvc, pv, mpd, sf, th, ratio)
            if nm_a != nm_b or not np.array_equal(a, b) or not np.array_equal(iv, pv):
                what.append("frustum_fused")
        # BoW searches + triangulation on a second random set
        n2 = int(rng.integers(0, 2000))
        kp2 = rand_kps(rng, n2, w, h, uvo)
        de2 = noisy(rng, de[rng.integers(0, max(n, 1), n2)], 0.05) if n else rng.integers(0, 256, (n2, 32), dtype=np.uint8)
        nn = int(rng.integers(1, 120))
        g1, g2 = groups(rng, n, nn), groups(rng, n2, nn)
        us1 = (rng.random(n) < 0.8).astype(np.uint8)
        us2 = (rng.random(n2) < 0.8).astype(np.uint8)
        for kf in (False, True):
            ga = m.SearchByBoW(uvo.FeatureVector(g1), de, kp["angle"], us1, uvo.FeatureVector(g2), de2, kp2["angle"], us2 if kf else None, kf_kf=kf)
            gb = o.search_by_bow(kf, g1, de, kp["angle"], us1, g2, de2, kp2["angle"], us2 if kf else None, ratio, ori)
            if ga[1] != gb[1] or not np.array_equal(ga[0], gb[0]):
                what.append("bow%d" % kf)
        F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-3, (3, 3)).astype(np.float32))
        s2 = (sf * sf * np.float32(rng.choice([1.0, 50.0, 2000.0]))).astype(np.float32)
        ta = m.SearchForTriangulation(uvo.FeatureVector(g1), kp, de, 1 - us1, uvo.FeatureVector(g2), kp2, de2, 1 - us2, F12, s2)
        tb = o.search_for_triangulation(g1, kp, de, 1 - us1, g2, kp2, de2, 1 - us2, F12, s2, ori)
        if ta[1] != tb[1] or not np.array_equal(ta[0], tb[0]):
            what.append("triang")
        # SearchBySim3: projections of set 1 into frame 2 and back (random targets near key points of the other frame)
        if n and n2:
            def proj(kp_to, cnt):
                nn_ = rng.integers(0, len(kp_to), cnt)
                return ((rng.random(cnt) < 0.85).astype(np.uint8), (kp_to["x"][nn_] + rng.normal(0, 3, cnt)).astype(np.float32),
                        (kp_to["y"][nn_] + rng.normal(0, 3, cnt)).astype(np.float32), rng.integers(0, 8, cnt).astype(np.int32)), nn_
            p12, i12 = proj(kp2, n)
            p21, i21 = proj(kp, n2)
            md1, md2 = noisy(rng, de2[i12], 0.08), noisy(rng, de[i21], 0.08)
            th3 = float(rng.choice([7.5, 15.0]))
            sa = m.SearchBySim3(kp, de, bounds, kp2, de2, bounds, p12, md1, p21, md2, sf, sf, th3)
            sb = o.search_by_sim3(kp, de, bounds, kp2, de2, bounds, p12, md1, p21, md2, sf, sf, th3)
            if sa[1] != sb[1] or not np.array_equal(sa[0], sb[0]):
                what.append("sim3")
        # knn-2 and the medoid pick
        if n and n2:
            ka, kb = m.knn2(de[:2000], de2), o.knn2(de[:2000], de2)
            if not (np.array_equal(ka[0], kb[0]) and np.array_equal(ka[2], kb[2])):
                what.append("knn2")
        if what:
            bad += 1
            print("MISMATCH trial", t, what, dict(w=w, h=h, n=n, M=M, n2=n2, ratio=ratio, ori=ori, th=th))
        m.close()
    print("trials", n_trials, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
