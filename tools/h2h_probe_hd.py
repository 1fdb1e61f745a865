"""Streaming the asynchronous host form at 1920x1080: contiguous frames (one linear upload per batch) vs frames one row apart
(one 2-D upload per frame)."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.zeros(1, device="cuda")
uvo = importlib.import_module("u-vip-slam_amd")
synth = importlib.import_module("u-vip-slam_amd.synth")
W, H, bs = 1920, 1080, 64
base = synth.make_sequence(0, 8, W, H, n_shapes=2500)
for pad in (0, 1):
    store = uvo.pinned_empty((bs, H + pad, W), np.uint8)
    frames = store[:, :H, :]
    for i in range(bs):
        frames[i] = base[i % 8]
    ex = uvo.ORBextractor(2000, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=bs)
    ex.set_pipeline(2)
    cap = ex.cap
    bufs = [(uvo.pinned_empty((bs, cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((bs, cap, 32), np.uint8), uvo.pinned_empty((bs,), np.int32)) for _ in range(2)]
    t = ex.submit(frames, *bufs[0]); ex.wait(t)
    t0 = time.perf_counter()
    for i in range(6):
        t = ex.submit(frames, *bufs[0]); ex.wait(t)
    serial = (time.perf_counter() - t0) / 6 * 1e3
    nb = 20
    t = ex.submit(frames, *bufs[0])
    t0 = time.perf_counter()
    for i in range(1, nb):
        t2 = ex.submit(frames, *bufs[i % 2]); ex.wait(t); t = t2
    ex.wait(t)
    dt = time.perf_counter() - t0
    print("frames %s: serial submit+wait %.3f ms, continuous %.3f ms/batch = %.0f frames/s" % ("contiguous" if pad == 0 else "one row apart", serial, dt / (nb - 1) * 1e3, (nb - 1) * bs / dt))
    ex.close()
