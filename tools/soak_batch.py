#!/usr/bin/env python3
"""Developer soak of the batched HBM-resident path (bench.py's step: two pipeline lanes, knn-2 in the extracting lane's stream): random
batch sizes, frame sizes, feature counts, thresholds and sequence seeds, every frame and every pair against the oracle.
Usage: soak_batch.py [n_trials] [seed]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    torch.zeros(1, device="cuda")
    uvo = importlib.import_module("u-vip-slam_amd")
    synth = importlib.import_module("u-vip-slam_amd.synth")
    import oracle_lib
    import test_gpu_scale_parity as T
    o = oracle_lib.Oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(n_trials):
        B = int(rng.choice([3, 8, 17, 40, 64]))
        w, h = [(640, 512), (752, 480), (320, 240), (500, 375), (1024, 300)][int(rng.integers(0, 5))]
        nfeat = int(rng.choice([300, 1000, 1500]))
        th = int(rng.choice([7, 12, 20, 30]))
        noise = "cumulative" if rng.random() < 0.25 else "sensor"
        frames = synth.make_sequence(int(rng.integers(0, 1 << 20)), B, w, h, noise=noise) if noise != "sensor" else synth.make_sequence(int(rng.integers(0, 1 << 20)), B, w, h)
        stream = "lane" if rng.random() < 0.7 else "own"
        try:
            res = T._run_bench_path(uvo, frames, nfeat, th, passes=int(rng.integers(2, 6)), matcher_stream=stream)
            T._check_against_oracle(uvo, o, frames, nfeat, th, res, "trial %d" % t, 0)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", t, B, w, h, nfeat, th, noise, stream, str(e)[:200])
    print("trials", n_trials, "mismatches", bad)


if __name__ == "__main__":
    main()
