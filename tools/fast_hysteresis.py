#!/usr/bin/env python3
"""What the one-batch lag of the adaptive FAST form costs (UVO_TUNE_FAST_MODE adaptive: a level's form is chosen from the previous batch
of the same pipeline lane): extraction time per batch of a steady textured stream, a steady low-contrast stream, and a stream that
alternates them, with one pipeline lane (every batch runs in the form the OTHER kind asked for) and with two (each lane sees one kind
only).  Batch 256 @ 640x512, HBM-resident.  python tools/fast_hysteresis.py > profiles/r04_fast_hysteresis.json"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
uvo = importlib.import_module("u-vip-slam_amd")
synth = importlib.import_module("u-vip-slam_amd.synth")
B, W, H = 256, 640, 512
tex = synth.make_sequence(0, B, W, H, n_shapes=400)
low = (tex.astype(np.float32) * 0.12 + 110 * 0.88).astype(np.uint8)
d = {"tex": torch.from_numpy(tex).to("cuda"), "low": torch.from_numpy(low).to("cuda")}
out = {"batch": B, "shape": [W, H], "unit": "ms per batch (extract only)"}
for depth in (1, 2):
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=B)
    ex.set_pipeline(depth)
    cap = ex.cap
    bufs = [(torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"), torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
             torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(depth)]

    def run(seq, n):
        for i in range(n):
            kp, de, cnt = bufs[i % depth]
            ex.extract_batch_device(d[seq[i % len(seq)]].data_ptr(), B, W, H, kp.data_ptr(), de.data_ptr(), cnt.data_ptr(), cap)
        ex.synchronize()

    r = {}
    for name, seq in (("steady textured", ["tex"]), ("steady low contrast", ["low"]), ("alternating", ["tex", "low"])):
        run(seq, 8)
        t0 = time.perf_counter()
        run(seq, 40)
        r[name] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
    r["alternating, were every batch in its own form"] = round((r["steady textured"] + r["steady low contrast"]) / 2, 4)
    r["cost of the lag"] = round(r["alternating"] - r["alternating, were every batch in its own form"], 4)
    out["pipeline depth %d" % depth] = r
    ex.close()
print(json.dumps(out, indent=1))
