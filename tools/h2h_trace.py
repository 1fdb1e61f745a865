#!/usr/bin/env python3
"""Developer probe for the host-to-host leg: streams N sharder jobs (batch 256 @ 640x512, two chunks per job, two jobs in flight, knn-2 on)
and nothing else -- run it under `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` and feed the CSVs to
tools/h2h_trace_summary.py to see how busy the copy engines and the CUs were.   python tools/h2h_trace.py [jobs]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.zeros(1, device="cuda")
uvo = importlib.import_module("u-vip-slam_amd")
synth = importlib.import_module("u-vip-slam_amd.synth")
B, W, H = 256, 640, 512
jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames = uvo.pinned_empty((B + 1, H, W), np.uint8)
frames[:] = synth.make_sequence(0, B + 1, W, H)
sh = uvo.Sharder(1000, 1.2, 8, 20, max_width=W, max_height=H, devices=[0], chunk_frames=B // 2, match=True)
cap = sh.cap
sets = [[uvo.pinned_empty((B, cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((B, cap, 32), np.uint8), uvo.pinned_empty((B,), np.int32)] +
        [uvo.pinned_empty((B, cap), t) for t in (np.int32, np.uint16, np.int32, np.uint16)] for _ in range(2)]
pending = None
for i in range(3):
    sh.run(frames, 0, B, *sets[0])
t0 = time.perf_counter()
for i in range(jobs):
    t = sh.submit(frames, 0, B, *sets[i % 2])
    if pending is not None:
        sh.wait(pending)
    pending = t
sh.wait(pending)
dt = time.perf_counter() - t0
print("jobs %d  ms/job %.3f  frames/s %.0f" % (jobs, dt / jobs * 1e3, jobs * B / dt))
sh.close()
