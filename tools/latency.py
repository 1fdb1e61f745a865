#!/usr/bin/env python3
"""Per-frame latency of the batch=1 path (BASELINE.json configs[1]): 640x512, 1000 features, 8 levels.
  host   : uvo_extract() with host image in / host keypoints+descriptors out (PCIe both ways, what Tracking.cc would call)
  device : uvo_extract_batch_device(batch=1) + stream sync, image and outputs resident in HBM
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    uvo = importlib.import_module("u-vip-slam_amd")
    synth = importlib.import_module("u-vip-slam_amd.synth")
    img = synth.make_frame(1000)
    NB = int(os.environ.get("UVO_LAT_BATCH", "1"))   # frames per device-resident call (the host-buffer loop stays one frame)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512, max_batch=NB)
    # A/B knobs: UVO_LAT_PYR_FORM = 0 auto / 1 per-level launches / 2 k_pyr_tiles; UVO_LAT_PYR_GROUPS = "first:txXty[w],..." forced level groups
    if os.environ.get("UVO_LAT_PYR_FORM"):
        ex.tune(uvo.UVO_TUNE_PYR_FORM, int(os.environ["UVO_LAT_PYR_FORM"]))
    if os.environ.get("UVO_LAT_ZERO_COPY"):
        ex.tune(uvo.UVO_TUNE_ZERO_COPY_OUT, int(os.environ["UVO_LAT_ZERO_COPY"]))
    if os.environ.get("UVO_LAT_SPIN"):
        ex.tune(uvo.UVO_TUNE_SPIN_WAIT, int(os.environ["UVO_LAT_SPIN"]))
    if os.environ.get("UVO_LAT_OCT_WIDE"):
        ex.tune(uvo.UVO_TUNE_OCT_WIDE_MAX, int(os.environ["UVO_LAT_OCT_WIDE"]))
    if os.environ.get("UVO_LAT_FEW"):
        ex.tune(uvo.UVO_TUNE_FEW_FRAMES, int(os.environ["UVO_LAT_FEW"]))
    if os.environ.get("UVO_LAT_PYR_GROUPS"):
        for g in os.environ["UVO_LAT_PYR_GROUPS"].split(","):
            first, grid = g.split(":")
            wide = (1 << 24 if grid.endswith("w") else 0) | (1 << 25 if grid.endswith("r") else 0)
            tx, ty = grid.rstrip("wr").split("x")
            ex.tune(uvo.UVO_TUNE_PYR_TILE_GROUP, wide | int(first) << 16 | int(tx) << 8 | int(ty))
    trace = bool(os.environ.get("UVO_LAT_TRACE"))   # under rocprofv3 --kernel-trace: the device-resident loop only (tools/latency_trace.py reads the timeline)
    for _ in range(20):
        kp, de = ex(img)
    t = []
    for _ in range(20 if trace else 300):
        t0 = time.perf_counter()
        kp, de = ex(img)
        t.append(time.perf_counter() - t0)
    host = np.array(t) * 1e3
    # the call Tracking makes (src/Tracking.cc:896-946): 400 tracked keypoints fill the occupancy grid, the extractor tops up to 1000 (top-up mode)
    ex_t = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512, max_batch=1, max_input_keypoints=400)
    tracked = kp[::3][:400].copy()
    for _ in range(20):
        kq, dq = ex_t.extract_tracked(img, tracked, 20, 600)
    t = []
    for _ in range(20 if trace else 300):
        t0 = time.perf_counter()
        kq, dq = ex_t.extract_tracked(img, tracked, 20, 600)
        t.append(time.perf_counter() - t0)
    topup = np.array(t) * 1e3
    if os.environ.get("UVO_LAT_TOPUP_PROFILE"):
        ex_t.profile(True)
        for _ in range(50):
            ex_t.extract_tracked(img, tracked, 20, 600)
        print({k: round(v[0] / 50 * 1e3, 1) for k, v in sorted(ex_t.kernel_times().items())}, file=sys.stderr)
    ex_t.close()
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(np.stack([synth.make_frame(1000 + b) for b in range(NB)])).to(dev)
    cap = ex.cap
    d_kp = torch.zeros((NB, cap, 7), dtype=torch.float32, device=dev)
    d_de = torch.zeros((NB, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(NB, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for _ in range(20):
        ex.extract_batch_device(d_img.data_ptr(), NB, 640, 512, d_kp.data_ptr(), d_de.data_ptr(), d_n.data_ptr(), cap)
    ex.synchronize()
    t, tq = [], []
    for _ in range(300):
        t0 = time.perf_counter()
        ex.extract_batch_device(d_img.data_ptr(), NB, 640, 512, d_kp.data_ptr(), d_de.data_ptr(), d_n.data_ptr(), cap)
        t1 = time.perf_counter()
        ex.synchronize()
        t.append(time.perf_counter() - t0)
        tq.append(t1 - t0)
    devt = np.array(t) * 1e3
    enq = np.array(tq) * 1e3      # the host's share: all launches enqueued (the wait for the stream follows)
    if trace:
        print(json.dumps({"device_ms_median": round(float(np.median(devt)), 4), "enqueue_ms_median": round(float(np.median(enq)), 4), "host_ms_median": round(float(np.median(host)), 4), "topup_host_ms_median": round(float(np.median(topup)), 4), "topup_new_keypoints": int(len(kq))}))
        return
    ex.profile(True)
    for _ in range(50):
        ex.extract_batch_device(d_img.data_ptr(), NB, 640, 512, d_kp.data_ptr(), d_de.data_ptr(), d_n.data_ptr(), cap)
    kt = ex.kernel_times()
    print(json.dumps({"workload": "configs[1]: batch=1, 640x512, 1000 feats, 8 levels, fastTh 20", "keypoints": int(len(kp)),
                      "pyr_form": os.environ.get("UVO_LAT_PYR_FORM", "0"), "pyr_groups": os.environ.get("UVO_LAT_PYR_GROUPS", ""),
                      "host_ms_median": round(float(np.median(host)), 4), "topup_host_ms_median": round(float(np.median(topup)), 4), "topup_new_keypoints": int(len(kq)), "host_ms_p95": round(float(np.percentile(host, 95)), 4),
                      "device_ms_median": round(float(np.median(devt)), 4), "enqueue_ms_median": round(float(np.median(enq)), 4), "device_ms_p95": round(float(np.percentile(devt, 95)), 4),
                      "kernel_us": {k: round(v[0] / 50 * 1e3, 1) for k, v in sorted(kt.items())}}))


if __name__ == "__main__":
    main()
