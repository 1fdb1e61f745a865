#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-buffer boundary: frames in page-locked host memory -> uvo_extract_batch_submit / _wait with two
batches in flight (upload and download of one batch overlap the kernels of the other) -> keypoints and descriptors back in host
memory.  Prints one JSON object; the synchronous uvo_extract_batch on pageable memory is timed beside it."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    torch.zeros(1, device="cuda")
    uvo = importlib.import_module("u-vip-slam_amd")
    synth = importlib.import_module("u-vip-slam_amd.synth")
    W, H, B, steps = 640, 512, 256, 20
    base = [synth.make_frame(100 + i, W, H) for i in range(16)]
    frames = np.stack([base[i % 16] for i in range(B)])
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=B)
    for _ in range(2):
        ex.extract_batch(frames)
    t0 = time.perf_counter()
    for _ in range(3):
        ex.extract_batch(frames)
    sync_fps = 3 * B / (time.perf_counter() - t0)
    ex.set_pipeline(2)
    cap = ex.cap
    bufs = []
    for k in range(2):
        img = uvo.pinned_empty((B, H, W), np.uint8)
        img[:] = frames
        bufs.append((img, uvo.pinned_empty((B, cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((B, cap, 32), np.uint8), uvo.pinned_empty((B,), np.int32)))
    tick = [ex.submit(*bufs[0]), ex.submit(*bufs[1])]
    ex.wait(tick[0]), ex.wait(tick[1])
    tick = [ex.submit(*bufs[0]), None]
    t0 = time.perf_counter()
    for i in range(1, steps + 1):
        tick[i & 1] = ex.submit(*bufs[i & 1])
        ex.wait(tick[(i - 1) & 1])
    dt = time.perf_counter() - t0
    ex.wait(tick[steps & 1])
    mb_in, mb_out = B * W * H / 1e6, B * cap * 60 / 1e6
    print(json.dumps({"workload": "640x512, 1000 feats, batch 256, FullDetect, host buffers in and out",
                      "async_pinned_depth2_frames_per_s": round(steps * B / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 3),
                      "MB_in_per_batch": round(mb_in, 1), "MB_out_per_batch": round(mb_out, 1),
                      "sync_pageable_frames_per_s": round(sync_fps, 1), "mean_keypoints": float(bufs[0][3].mean())}, indent=1))


if __name__ == "__main__":
    main()
