#!/usr/bin/env python3
"""Timings of the BASELINE.json configurations that are not the bench line (parity-test cases, measured for the record):
  configs[3] per-GPU shard: 1920x1080 @ 2000 features, batch 128, HBM-resident extraction (one GPU's share of the 1024-frame job)
  configs[4]: 752x480 @ 1000 features (fastTh 7) extract + isInFrustum + SearchByProjection against 5000 map points, per frame
Prints one JSON object."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    torch.zeros(1, device="cuda")
    uvo = importlib.import_module("u-vip-slam_amd")
    synth = importlib.import_module("u-vip-slam_amd.synth")
    out = {}
    # ---- configs[3] ----
    B, W, H = 128, 1920, 1080
    base = [synth.make_frame(2000 + i, W, H, n_shapes=1600) for i in range(4)]
    frames = np.stack([base[i % 4] for i in range(B)])
    d = torch.from_numpy(frames).cuda()
    ex = uvo.ORBextractor(2000, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=B)
    ex.set_pipeline(2)
    cap = ex.cap
    bufs = [(torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"), torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
             torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(2)]
    def step(i):
        kp, de, n = bufs[i % 2]
        ex.extract_batch_device(d.data_ptr(), B, W, H, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap)
    for i in range(4):
        step(i)
    ex.synchronize()
    t0 = time.perf_counter()
    steps = 10
    for i in range(steps):
        step(i)
    ex.synchronize()
    dt = time.perf_counter() - t0
    out["configs[3] per GPU: 1920x1080 @2000 feats, batch 128, extract only"] = {
        "frames_per_s": round(B * steps / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 3), "mean_keypoints": float(bufs[0][2].float().mean())}
    ex.close()
    del d
    # ---- configs[4] ----
    workloads = importlib.import_module("u-vip-slam_amd.workloads")
    W, H = workloads.EUROC_W, workloads.EUROC_H
    img = synth.make_frame(31337, W, H)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 7, max_width=W, max_height=H)
    m = uvo.ORBmatcher(0.8, max_query=4096, max_map_points=8192)
    kp, de = ex(img)
    sf = ex.mvScaleFactor.copy()
    n = len(kp)
    mp = workloads.config4_local_map(kp, de, sf)
    xyz, nrm, mnd, mxd, mp_desc = mp["xyz"], mp["normal"], mp["min_distance"], mp["max_distance"], mp["mp_desc"]
    cam = uvo.CameraPose.make(mp["R"], mp["t"], mp["Ow"], workloads.EUROC_FX, workloads.EUROC_FY, workloads.EUROC_CX, workloads.EUROC_CY, (0, 0, W, H))
    imu = workloads.ImuStress()
    stream = workloads.imu_stream(64)
    counter = [0]

    def imu_step():   # the host work between two frames: 10 samples through the restated IMUPreintegrator::update
        counter[0] += 1
        return imu.preintegrate(stream[counter[0] % len(stream)])

    def frame_two_calls():
        imu_step()
        k, d_ = ex(img)
        valid, u, v, level, vc = m.project_points(uvo.PROJECT_FRUSTUM, cam, xyz, nrm, mnd, mxd, None, sf, 1.2, 0.5)
        a = np.full(len(k), -1, np.int32)
        return m.SearchByProjection(k, d_, (0, 0, W, H), a, u, v, level, vc, valid, mp_desc, sf, 1.0)

    def frame_fused():
        imu_step()
        k, d_ = ex(img)
        a = np.full(len(k), -1, np.int32)
        return m.SearchPointsInFrustum(k, d_, a, cam, xyz, nrm, mnd, mxd, None, mp_desc, sf, 1.2, 0.5, 1.0)[0]
    res = {}
    for name, frame in (("two_calls", frame_two_calls), ("fused", frame_fused)):
        for _ in range(5):
            nm = frame()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            nm = frame()
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        res[name] = {"ms_per_frame_median": round(float(np.median(ts)), 3), "ms_p95": round(float(np.percentile(ts, 95)), 3), "matches": int(nm)}
    assert res["two_calls"]["matches"] == res["fused"]["matches"]
    res["keypoints"] = int(n)
    res["imu"] = "10 IMU samples per frame through the restated IMUPreintegrator::update on the host, inside the frame time"
    res["note"] = "fused = uvo_search_points_in_frustum (Tracking::SearchReferencePointsInFrustum as one call); two_calls = uvo_project_points + uvo_search_by_projection"
    out["configs[4]: 752x480 extract + isInFrustum + SearchByProjection vs 5000 map points (host buffers in/out)"] = res
    # ---- the LocalMapping loops: 20 single calls against the batched forms (src/LocalMapping.cc:1058-1080, :1228-1236) ----
    rng = np.random.default_rng(5)
    kp1, de1 = ex(img)
    sigma2 = (sf * sf).astype(np.float32)

    def groups(de):   # stand-in vocabulary nodes (tests/test_gpu_parity.py: _bow_groups)
        bits = np.unpackbits(de, axis=1)[:, [3, 41, 77, 130, 201, 250]]
        node = (bits * (1 << np.arange(6))).sum(1) % 40
        g = {}
        for i in range(len(de)):
            g.setdefault(int(node[i]) * 7 + 3, []).append(i)
        return g
    fv1 = uvo.FeatureVector(groups(de1))
    has1 = (rng.random(len(kp1)) < 0.25).astype(np.uint8)
    neigh = []
    for k in range(20):
        kp2, de2 = ex(synth.warp_frame(img, 900 + k))
        F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 3e-4, (3, 3)).astype(np.float32))
        neigh.append((uvo.FeatureVector(groups(de2)), kp2, de2, (rng.random(len(kp2)) < 0.3).astype(np.uint8), F12, (sigma2 * np.float32(40)).astype(np.float32)))
    mt = uvo.ORBmatcher(0.6, False)

    def tri_single():
        return sum(mt.SearchForTriangulation(fv1, kp1, de1, has1, *nb)[1] for nb in neigh)

    def tri_batch():
        mt.SearchForTriangulationBatch(fv1, kp1, de1, has1, neigh)
        return sum(mt.SearchForTriangulationNext(k, has1)[1] for k in range(20))
    M = 1000
    xyz2 = (rng.normal(0, 1, (M, 3)) * [3, 2, 1.5] + [0, 0, 6]).astype(np.float32)
    nrm2 = xyz2 / np.linalg.norm(xyz2, axis=1, keepdims=True)
    mn2 = (np.linalg.norm(xyz2, axis=1) * 0.5).astype(np.float32)
    mx2 = (mn2 * 6).astype(np.float32)
    md2 = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    targets = [(nb[1], nb[2], cam, sf) for nb in neigh]

    def fuse_single():
        tot = 0
        for kp2, de2, c, _ in targets:
            valid, u, v, level, _ = mt.project_points(uvo.PROJECT_FUSE, c, xyz2, nrm2, mn2, mx2, None, sf)
            tot += int((mt.FuseSearch(kp2, de2, (0, 0, W, H), u, v, level, valid, md2, sf, 3.0)[0] >= 0).sum())
        return tot

    def fuse_batch():
        return int((mt.FuseBatch(targets, xyz2, nrm2, mn2, mx2, None, md2, 3.0)[0] >= 0).sum())
    lm = {}
    for name, fn in (("SearchForTriangulation x 20, single calls", tri_single), ("SearchForTriangulation x 20, batched", tri_batch),
                     ("Fuse x 20 targets, single calls (project + search)", fuse_single), ("Fuse x 20 targets, batched", fuse_batch)):
        for _ in range(3):
            r = fn()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        lm[name] = {"ms_median": round(float(np.median(ts)) * 1e3, 3), "result": int(r)}
    assert lm["SearchForTriangulation x 20, single calls"]["result"] == lm["SearchForTriangulation x 20, batched"]["result"]
    assert lm["Fuse x 20 targets, single calls (project + search)"]["result"] == lm["Fuse x 20 targets, batched"]["result"]
    lm["host_waits"] = {"SearchForTriangulation": "20 -> 1", "Fuse (project + window count + resolve)": "60 -> 1"}
    out["LocalMapping loops, 752x480 key frames of ~1000 key points, 20 neighbours"] = lm
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
