#!/usr/bin/env python3
"""Developer soak: CLAHE, the BoW transform, the projection prologues, haloc and the KLT pyramid on random sizes / parameters."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    torch.zeros(1, device="cuda")
    uvo = importlib.import_module("u-vip-slam_amd")
    synth = importlib.import_module("u-vip-slam_amd.synth")
    import oracle_lib
    import test_gpu_parity as tg
    o = oracle_lib.Oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    ex = uvo.ORBextractor(500, 1.2, 4, 0, 20, max_width=1000, max_height=800)
    m = uvo.ORBmatcher(0.8)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    for t in range(n_trials):
        what = []
        w, h = int(rng.integers(64, 1000)), int(rng.integers(64, 800))
        img = synth.make_frame(int(rng.integers(1 << 30)), w, h, n_shapes=max(10, w * h // 1200)) if t % 3 else rng.integers(0, 256, (h, w), dtype=np.uint8)
        img = (img * rng.uniform(0.1, 1.0)).astype(np.uint8)
        tiles = (int(rng.integers(1, 20)), int(rng.integers(1, 20)))
        clip = float(rng.choice([0.0, 1.0, 4.0, 40.0]))
        try:
            if not np.array_equal(ex.clahe(img, clip, tiles), o.clahe(img, clip, tiles)):
                what.append("clahe")
        except uvo.UvoError:
            pass  # tile grid too coarse for the REFLECT_101 extension: rejected
        # KLT pyramid
        win = (int(rng.choice([7, 15, 21, 31])),) * 2
        ml = int(rng.integers(0, 6))
        k = uvo.KLT(w, h, win, ml, max_points=64)
        nl = k.build_pyramid(0, img)
        po = o.klt_pyramid(img, win, ml)
        if nl != po.levels:
            what.append("klt_levels")
        else:
            for l in range(nl):
                gi, gd = k.read_level(0, l)
                oi, od = po.level(l)
                if not (np.array_equal(gi, oi) and np.array_equal(gd, od)):
                    what.append("klt_level%d" % l)
        k.close()
        # BoW
        voc = tg._random_vocabulary(rng, int(rng.integers(2, 12)), int(rng.integers(1, 5)), int(rng.integers(0, 4)), int(rng.integers(0, 3)))
        V = uvo.ORBVocabulary(voc["child_start"], voc["children"], voc["descriptor"], voc["word_id"], voc["weight"], voc["L"], voc["weighting"], voc["normalize"])
        feats = rng.integers(0, 256, (int(rng.integers(0, 1500)), 32), dtype=np.uint8)
        lu = int(rng.integers(0, 7))
        g, r = V.transform(feats, lu), o.bow_transform(voc, feats, lu)
        fvg = {int(g[4].node[j]): [int(x) for x in g[4].feat[g[4].start[j]:g[4].start[j + 1]]] for j in range(len(g[4].node))}
        if not (np.array_equal(g[0], r[0]) and np.array_equal(g[2], r[2]) and np.array_equal(g[3][0], r[3][0]) and
                np.array_equal(g[3][1].view(np.uint64), r[3][1].view(np.uint64)) and fvg == r[4]):
            what.append("bow")
        V.close()
        # projection prologues
        R, tt, Ow = tg._random_pose(rng)
        cam = uvo.CameraPose.make(R, tt, Ow, 458.654, 457.296, w / 2, h / 2, (0, 0, w, h))
        cam_o = np.concatenate([R.reshape(9), tt, Ow, np.float32([458.654, 457.296, w / 2, h / 2]), np.float32([0, w, 0, h])]).astype(np.float32)
        n = int(rng.integers(1, 5000))
        xyz = (rng.normal(0, 1, (n, 3)) * [4, 3, 4] + [0, 0, 6]).astype(np.float32)
        nrm = rng.normal(0, 1, (n, 3))
        nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
        d = np.linalg.norm(xyz - Ow, axis=1)
        mn = (d * rng.uniform(0.3, 1.4, n)).astype(np.float32)
        mx = (mn * rng.uniform(1.5, 6.0, n)).astype(np.float32)
        for mode in (0, 1, 2):
            a = m.project_points(mode, cam, xyz, nrm, mn, mx, None, sf, 1.2, 0.5)
            b = o.project_points(mode, cam_o, xyz, nrm, mn, mx, None, sf, 1.2, 0.5)
            if not all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y) for x, y in zip(a, b)):
                what.append("project%d" % mode)
        # haloc
        nd = int(rng.integers(0, 3000))
        proj = rng.normal(0, 1, (int(rng.integers(1, 5)), 6000)).astype(np.float32)
        dd = rng.integers(0, 256, (nd, 32), dtype=np.uint8)
        if not np.array_equal(m.haloc_hash(proj, dd).view(np.uint32), o.haloc_hash(proj, dd).view(np.uint32)):
            what.append("haloc")
        if what:
            bad += 1
            print("MISMATCH trial", t, what, dict(w=w, h=h, tiles=tiles, clip=clip, win=win, ml=ml))
    print("trials", n_trials, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
