#!/usr/bin/env python3
"""A/B timing of the pyramid stage on the GPU: the fused launch (k_pyramid) under its knobs against the per-level chain, HBM-resident
batches as bench.py runs them.  Prints per variant: the stage's kernel time alone on the chip (pipeline depth 1, HIP events) and the
extract-only throughput at pipeline depth 2.   python tools/pyr_ab.py [--config 2|3]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--variants", default="")
    args = ap.parse_args()
    import torch
    uvo = importlib.import_module("u-vip-slam_amd")
    synth = importlib.import_module("u-vip-slam_amd.synth")
    W, H, NFEAT, B, NS = (640, 512, 1000, 256, 400) if args.config == 2 else (1920, 1080, 2000, 128, 2500)
    B = args.batch or B
    dev = torch.device("cuda", 0)
    frames = synth.make_sequence(0, min(B + 1, 64), W, H, n_shapes=NS)
    frames = np.concatenate([frames] * ((B + 1 + len(frames) - 1) // len(frames)))[:B + 1]
    d0 = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
    ring = [d0, torch.flip(d0, dims=[2]).contiguous(), torch.flip(d0, dims=[1]).contiguous(), torch.flip(d0, dims=[1, 2]).contiguous()]
    ex = uvo.ORBextractor(NFEAT, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=B + 1)
    cap = ex.cap
    outs = [(torch.zeros((B + 1, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev),
             torch.zeros(B + 1, dtype=torch.int32, device=dev)) for _ in range(2)]
    cnt = [0]

    def step():
        o = outs[cnt[0] % 2]
        k = cnt[0] % 4
        cnt[0] += 1
        ex.extract_batch_device(ring[k].data_ptr(), B + 1, W, H, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), cap)

    def sync():
        ex.synchronize()
        torch.cuda.synchronize()

    variants = [("legacy", dict(legacy=1))]
    for tail in (3, 4, 2, 5, 8, 0):
        for nb in (1, 2):
            for run in (5, 3, 8):
                variants.append(("split t%d b%d n%d" % (tail, nb, run), dict(legacy=0, waves=0, bands=nb, rows=7, tail=tail, run=run)))
    if args.variants:
        keep = args.variants.split(",")
        variants = [v for v in variants if any(k in v[0] for k in keep)]
    res = []
    for name, v in variants:
        ex.tune(uvo.UVO_TUNE_PYR_MODE, uvo.UVO_PYR_MODE_CHAIN if v["legacy"] else uvo.UVO_PYR_MODE_SPLIT)
        if not v["legacy"]:
            ex.tune(uvo.UVO_TUNE_PYR_WAVES, v["waves"])
            ex.tune(uvo.UVO_TUNE_PYR_ROWS, v["rows"])
            ex.tune(uvo.UVO_TUNE_PYR_BANDS, v["bands"])
            ex.tune(uvo.UVO_TUNE_PYR_TAIL, v["tail"])
            ex.tune(uvo.UVO_TUNE_PYR_RUN, v["run"])
        ex.set_pipeline(1)
        for _ in range(3):
            step()
        sync()
        ex.profile(True)
        for _ in range(5):
            step()
        sync()
        kt = dict(ex.kernel_times())
        ex.profile(False)
        pyr_ms = sum(ms for k, (ms, n) in kt.items() if k in ("k_pyramid", "k_pad_level0", "k_resize_level", "k_pyr_stream", "k_pyr_stream1")) / 5
        ex.set_pipeline(2)
        for _ in range(4):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        # live: the stage with the other lane's kernels beside it
        ex.profile(True)
        for _ in range(10):
            step()
        sync()
        kl = dict(ex.kernel_times())
        ex.profile(False)
        live_ms = sum(ms for k, (ms, n) in kl.items() if k in ("k_pyramid", "k_pad_level0", "k_resize_level", "k_pyr_stream", "k_pyr_stream1")) / 10
        r = {"variant": name, "pyramid_alone_ms": round(pyr_ms, 4), "pyramid_live_ms": round(live_ms, 4), "extract_frames_per_s": round(B * args.steps / dt, 0),
             "ms_per_step": round(dt / args.steps * 1e3, 4)}
        print(json.dumps(r), flush=True)
        res.append(r)
    ex.close()


if __name__ == "__main__":
    main()
