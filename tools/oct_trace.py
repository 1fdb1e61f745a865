#!/usr/bin/env python3
"""Developer tool: per-phase timeline of one k_octree workgroup (level 0 of the middle frame of a 256-frame batch).
Needs a trace build:  UVO_EXTRA_FLAGS=-DUVO_OCT_TRACE python3 u-vip-slam_amd/build.py
Prints the source line of every phase end in octree_core.hpp with the time since the previous mark (wall_clock64, 100 MHz)."""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
uvo = importlib.import_module("u-vip-slam_amd")
import bench  # noqa: E402


def main():
    batch = int(os.environ.get("BATCH", "256"))
    W, H, NF = (1920, 1080, 2000) if os.environ.get("HD") else (640, 512, 1000)
    synth = importlib.import_module("u-vip-slam_amd.synth")
    ex = uvo.ORBextractor(NF, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=batch)
    frames = synth.make_sequence(0, batch, W, H, n_shapes=2500 if W > 1000 else 400)
    d_imgs = torch.from_numpy(frames).cuda()
    cap = ex.cap
    kp = torch.zeros((batch, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((batch, cap, 32), dtype=torch.uint8, device="cuda")
    nk = torch.zeros(batch, dtype=torch.int32, device="cuda")
    lib = uvo._lib if hasattr(uvo, "_lib") else uvo.lib
    lib.uvo_debug_oct_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvo_debug_oct_trace.restype = ctypes.c_int
    buf = np.zeros(2 * 1024, dtype=np.uint64)
    n = 0
    for it in range(3):
        ex.extract_batch_device(d_imgs.data_ptr(), batch, W, H, kp.data_ptr(), desc.data_ptr(), nk.data_ptr(), cap)
        ex.synchronize()
        n = lib.uvo_debug_oct_trace(buf.ctypes.data, 1024)
    bb = np.zeros(4096, dtype=np.uint64)
    lib.uvo_debug_oct_blocks.argtypes = [ctypes.c_void_p]
    lib.uvo_debug_oct_blocks(bb.ctypes.data)
    bb = bb.reshape(2048, 2).astype(np.int64)[: min(2048, batch * 8)]
    base = bb[:, 0].min()
    st = (bb[:, 0] - base) / 100.0
    en = (bb[:, 1] - base) / 100.0
    dur = en - st
    print("blocks: kernel span %.1f us; duration mean %.1f max %.1f; start max %.1f" % (en.max(), dur.mean(), dur.max(), st.max()))
    for lvl in range(8):
        d = dur[lvl::8]
        print("  level %d: dur mean %.1f max %.1f  start mean %.1f" % (lvl, d.mean(), d.max(), st[lvl::8].mean()))
    for t in (10, 50, 100, 200, 300, 400):
        print("  resident at %d us: %d" % (t, int(((st <= t) & (en > t)).sum())))
    rows = buf[: 2 * n].reshape(n, 2)
    if n == 0:
        print("no marks recorded")
        return
    t0 = rows[0, 1]
    if os.environ.get("SEQ"):   # the marks in the order they were taken (octree_core.hpp / octree_pyramid.hpp / octree.hip line : microseconds since the mark before)
        print("sequence:", " ".join("%d:%.2f" % (int(l), (int(t) - int(rows[i - 1, 1] if i else t0)) / 100.0) for i, (l, t) in enumerate(rows)))
    prev = t0
    agg = {}
    for line, t in rows:
        dt = (int(t) - int(prev)) / 100.0
        agg.setdefault(int(line), []).append(dt)
        prev = t
    print("marks", n, "total us", (int(rows[-1, 1]) - int(t0)) / 100.0)
    for line in sorted(agg):
        v = agg[line]
        print("line %4d  n=%3d  sum=%8.2f us  mean=%6.2f" % (line, len(v), sum(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
