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
    W, H = 752, 480
    rng = np.random.default_rng(7)
    img = synth.make_frame(31337, W, H)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 7, max_width=W, max_height=H)
    m = uvo.ORBmatcher(0.8, max_query=4096, max_map_points=8192)
    kp, de = ex(img)
    sf = ex.mvScaleFactor.copy()
    n, M = len(kp), 5000
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    R, t, Ow = np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    src = rng.integers(0, n, M)
    z = rng.uniform(2, 12, M)
    xyz = np.stack([(kp["x"][src] - cx) / fx * z, (kp["y"][src] - cy) / fy * z, z], 1).astype(np.float32)
    nrm = (xyz / np.linalg.norm(xyz, axis=1, keepdims=True)).astype(np.float32)
    dist = np.linalg.norm(xyz, axis=1)
    mxd = (dist * sf[kp["octave"][src]]).astype(np.float32)
    mnd = (mxd / sf[7]).astype(np.float32)
    mp_desc = de[src].copy()
    flip = rng.random((M, 256)) < 0.06
    mp_desc = np.packbits(np.unpackbits(mp_desc, axis=1) ^ flip, axis=1)
    cam = uvo.CameraPose.make(R, t, Ow, fx, fy, cx, cy, (0, 0, W, H))
    def frame_two_calls():
        k, d_ = ex(img)
        valid, u, v, level, vc = m.project_points(uvo.PROJECT_FRUSTUM, cam, xyz, nrm, mnd, mxd, None, sf, 1.2, 0.5)
        a = np.full(len(k), -1, np.int32)
        return m.SearchByProjection(k, d_, (0, 0, W, H), a, u, v, level, vc, valid, mp_desc, sf, 1.0)

    def frame_fused():
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
    res["note"] = "fused = uvo_search_points_in_frustum (Tracking::SearchReferencePointsInFrustum as one call); two_calls = uvo_project_points + uvo_search_by_projection"
    out["configs[4]: 752x480 extract + isInFrustum + SearchByProjection vs 5000 map points (host buffers in/out)"] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
