#!/usr/bin/env python3
"""Developer probe: where the host-to-host time of one sharder job goes (configs[2] batch)."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.zeros(1, device="cuda")
uvo = importlib.import_module("u-vip-slam_amd")
synth = importlib.import_module("u-vip-slam_amd.synth")
B, W, H = 256, 640, 512
frames = uvo.pinned_empty((B + 1, H, W), np.uint8)
frames[:] = synth.make_sequence(0, B + 1, W, H)
def run(chunk, match, reps=10):
    sh = uvo.Sharder(1000, 1.2, 8, 20, max_width=W, max_height=H, devices=[0], chunk_frames=chunk, match=match)
    cap = sh.cap
    kp, de, n = uvo.pinned_empty((B, cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((B, cap, 32), np.uint8), uvo.pinned_empty((B,), np.int32)
    m = [uvo.pinned_empty((B, cap), t) for t in (np.int32, np.uint16, np.int32, np.uint16)] if match else [None] * 4
    for _ in range(3):
        sh.run(frames, 0, B, kp, de, n, *m)
    t0 = time.perf_counter()
    for _ in range(reps):
        sh.run(frames, 0, B, kp, de, n, *m)
    dt = (time.perf_counter() - t0) / reps
    sh.close()
    return dt * 1e3
for chunk in (16, 32, 64, 128, 256):
    for match in (False,):
        print("chunk", chunk, "match", match, "ms/job %.3f" % run(chunk, match))
# raw copies
d = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
src = torch.from_numpy(frames[:B])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    d.copy_(src, non_blocking=True)
torch.cuda.synchronize()
print("H2D 84 MB: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))

# raw submit / wait, continuous (two batches in flight), per batch size
for bs in (64, 128, 256):
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=bs)
    ex.set_pipeline(2)
    cap = ex.cap
    bufs = [(uvo.pinned_empty((bs, cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((bs, cap, 32), np.uint8), uvo.pinned_empty((bs,), np.int32)) for _ in range(2)]
    nb = 40
    t = ex.submit(frames[:bs], *bufs[0])
    t0 = time.perf_counter()
    for i in range(1, nb):
        t2 = ex.submit(frames[(i * bs) % (B - bs + 1):(i * bs) % (B - bs + 1) + bs], *bufs[i % 2])
        ex.wait(t)
        t = t2
    ex.wait(t)
    dt = time.perf_counter() - t0
    print("continuous batch", bs, "frames/s %.0f  ms/batch %.3f" % ((nb - 1) * bs / dt, dt / (nb - 1) * 1e3))
    ex.close()
