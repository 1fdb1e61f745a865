#!/usr/bin/env python3
"""Per-kernel times of the pyramid launches (pipeline depth 1, HIP events), for a tail split: python tools/pyr_kt.py <tail> <run> [bands]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
uvo = importlib.import_module("u-vip-slam_amd")
synth = importlib.import_module("u-vip-slam_amd.synth")
tail, run = int(sys.argv[1]), int(sys.argv[2])
bands = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = 256
frames = synth.make_sequence(0, 32, 640, 512)
frames = np.concatenate([frames] * 9)[:B + 1]
d = torch.from_numpy(np.ascontiguousarray(frames)).to("cuda")
ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512, max_batch=B + 1)
ex.tune(uvo.UVO_TUNE_PYR_TAIL, tail); ex.tune(uvo.UVO_TUNE_PYR_RUN, run); ex.tune(uvo.UVO_TUNE_PYR_BANDS, bands)
ex.tune(uvo.UVO_TUNE_PYR_MODE, uvo.UVO_PYR_MODE_CHAIN if tail < 0 else uvo.UVO_PYR_MODE_SPLIT)
cap = ex.cap
kp = torch.zeros((B + 1, cap, 7), dtype=torch.float32, device="cuda"); de = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda"); n = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
for _ in range(3):
    ex.extract_batch_device(d.data_ptr(), B + 1, 640, 512, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap)
ex.synchronize()
ex.profile(True)
for _ in range(5):
    ex.extract_batch_device(d.data_ptr(), B + 1, 640, 512, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap)
ex.synchronize()
for k, (ms, nl) in sorted(ex.kernel_times().items()):
    print("%-20s %8.4f ms/step  %d launches" % (k, ms / 5, nl))
