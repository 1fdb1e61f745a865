#!/usr/bin/env python3
"""Per-level share of FAST cells that fall back to the literal threshold 7 (src/ORBextractor.cc:797) on the benchmark's frames, and what
each UVO_TUNE_FAST_MODE costs there.   python tools/fast_state_probe.py [--config 2|3] [--noise sensor|cumulative] [--contrast 1.0]"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--noise", default="sensor")
    ap.add_argument("--contrast", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=None)
    args = ap.parse_args()
    import torch
    torch.zeros(1, device="cuda")
    uvo = importlib.import_module("u-vip-slam_amd")
    synth = importlib.import_module("u-vip-slam_amd.synth")
    W, H, NF, B, NS = (640, 512, 1000, 256, 400) if args.config == 2 else (1920, 1080, 2000, 128, 2500)
    B = args.batch or B
    frames = synth.make_sequence(0, B, W, H, n_shapes=NS, noise=args.noise)
    if args.contrast != 1.0:
        frames = (frames.astype(np.float32) * args.contrast + 110 * (1 - args.contrast)).astype(np.uint8)
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(frames).to(dev)
    out = {"config": args.config, "noise": args.noise, "contrast": args.contrast, "batch": B}
    for mode, mid in (("two_pass", uvo.UVO_FAST_MODE_TWO_PASS), ("single_pass", uvo.UVO_FAST_MODE_SINGLE_PASS), ("adaptive", uvo.UVO_FAST_MODE_ADAPTIVE)):
        ex = uvo.ORBextractor(NF, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=B)
        ex.tune(uvo.UVO_TUNE_FAST_MODE, mid)
        cap = ex.cap
        kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        de = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(B, dtype=torch.int32, device=dev)
        for _ in range(3):
            ex.extract_batch_device(d_img.data_ptr(), B, W, H, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap)
        ex.synchronize()
        ex.profile(True)
        for _ in range(5):
            ex.extract_batch_device(d_img.data_ptr(), B, W, H, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap)
        ex.synchronize()
        kt = {k: round(v[0] / 5, 4) for k, v in ex.kernel_times().items()}
        t, fb, cells = ex.fast_state()
        out[mode] = {"ms": {k: kt[k] for k in kt if k.startswith("k_fast") or k == "k_octree"}, "pass_threshold": t.tolist(),
                     "fallback_share": [round(float(a) / (b * B), 4) for a, b in zip(fb, cells)], "mean_keypoints": float(n.float().mean())}
        ex.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
