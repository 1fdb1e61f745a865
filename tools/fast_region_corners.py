#!/usr/bin/env python3
"""How many FAST corners a k_fast_score wavefront region holds on the benchmark frames (CPU only: the oracle's FAST on the oracle's pyramid
planes): per 248 x 24 region + its one-pixel halo ring, at the two thresholds a level's streaming pass runs at (fastTh = 20: two-pass form; 7:
single pass).  The wavefront's LDS corner list holds 320 records (csrc/fast.hip FL_CAP); a busier region spills to memory -- this is the
distribution behind DESIGN.md section 7.1 ("why the corner lists spill").   python tools/fast_region_corners.py > profiles/r06_fast_region_corner_counts.txt"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

synth = importlib.import_module("u-vip-slam_amd.synth")
o = oracle_lib.Oracle()
frames = synth.make_sequence(40, 6, 640, 512, n_shapes=400)   # six frames of the benchmark's generator (bench.py CONFIGS[2])
oe = o.extractor(1000, 1.2, 8, 20)
print("frames: synth.make_sequence(40, 6, 640, 512, n_shapes=400); regions of 248 x 24 pixels of the detection window + a one-pixel ring; LDS list = 320 records")
for t in (20, 7):
    cnt = []
    for img in frames:
        oe(img)
        for l in range(8):
            pl = oe.level_plane(l)                       # padded plane (16-pixel border)
            kp = o.fast(pl, t, nms=False)
            x, y = kp["x"].astype(int), kp["y"].astype(int)
            h, w = pl.shape
            x0, y0, ww, hh = 32, 32, w - 64, h - 64      # detection window in padded coordinates
            m = (x >= x0 - 1) & (x < x0 + ww + 1) & (y >= y0 - 1) & (y < y0 + hh + 1)
            x, y = x[m] - x0, y[m] - y0
            for sx in range((ww + 247) // 248):
                for sy in range((hh + 23) // 24):
                    cnt.append((l, int(((x >= sx * 248 - 1) & (x < sx * 248 + 249) & (y >= sy * 24 - 1) & (y < sy * 24 + 25)).sum())))
    c = np.array([v for _, v in cnt])
    print("threshold %2d: %d regions, corners per region mean %.0f, median %.0f, 90 %% %.0f, 99 %% %.0f, max %d; regions above 320 records: %.0f %%, above 426: %.0f %%, above 500: %.0f %%; corners per frame %.0f"
          % (t, len(c), c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max(), 100 * (c > 320).mean(), 100 * (c > 426).mean(), 100 * (c > 500).mean(), c.sum() / len(frames)))
    for l in range(8):
        cl = np.array([v for ll, v in cnt if ll == l])
        print("    level %d: %3d regions, mean %5.0f, max %4d, above 320: %3.0f %%" % (l, len(cl), cl.mean(), cl.max(), 100 * (cl > 320).mean()))
