"""Developer probe: streamed sharder jobs at 1920x1080 (one shard, chunk 64) with and without matching, against the raw
submit / wait form (tools/h2h_probe_hd.py)."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.zeros(1, device="cuda")
uvo = importlib.import_module("u-vip-slam_amd")
synth = importlib.import_module("u-vip-slam_amd.synth")
B, W, H = 128, 1920, 1080
frames = uvo.pinned_empty((B + 1, H, W), np.uint8)
base = synth.make_sequence(0, 8, W, H, n_shapes=2500)
for i in range(B + 1):
    frames[i] = base[i % 8]
for match in (False, True):
    for chunk in (32, 64):
        sh = uvo.Sharder(2000, 1.2, 8, 20, max_width=W, max_height=H, devices=[0], chunk_frames=chunk, match=match)
        cap = sh.cap
        outs = []
        for _ in range(2):
            o = [uvo.pinned_empty((B, cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((B, cap, 32), np.uint8), uvo.pinned_empty((B,), np.int32)]
            o += [uvo.pinned_empty((B, cap), t) for t in (np.int32, np.uint16, np.int32, np.uint16)] if match else []
            outs.append(o)
        sh.run(frames, 0, B, *outs[0])
        nj = 8
        t = sh.submit(frames, 0, B, *outs[0])
        t0 = time.perf_counter()
        for j in range(1, nj):
            t2 = sh.submit(frames, 0, B, *outs[j % 2]); sh.wait(t); t = t2
        sh.wait(t)
        dt = (time.perf_counter() - t0) / (nj - 1)
        print("match %s chunk %d: %.3f ms per %d-frame job = %.0f frames/s" % (match, chunk, dt * 1e3, B, B / dt))
        sh.close()
