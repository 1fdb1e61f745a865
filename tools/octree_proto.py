#!/usr/bin/env python3
"""Prototype of the pass-parallel formulation of ORBextractor::DistributeOctTree used by the HIP kernel
(u-vip-slam_amd/csrc/octree.hip), checked here against the literal std::list restatement in oracle/.

Every "pass" below only uses data-parallel primitives (histogram by key, exclusive scan, sort of <= N keys),
so it transcribes 1:1 into one workgroup.  Run:  python tools/octree_proto.py  (needs oracle/liborb_oracle.so)
"""
import ctypes
import sys

import numpy as np


def octree_parallel(xs, ys, resp, ords, W, H, N):
    """xs, ys: integer coords relative to minBorder; resp: scores; ords: candidate order keys (unique).
    returns list of point indices in the reference's output (list) order."""
    P = len(xs)
    nIni = int(np.floor(np.float32(W) / np.float32(H) + np.float32(0.5)))  # C round(): W/H > 0
    hX = np.float32(W) / np.float32(nIni)
    # A = active generation: arrays in list order (front -> back)
    A_box = []   # (ULx, URx, ULy, BRy)
    A_cnt = []
    root_of = (xs.astype(np.float32) / hX).astype(np.int32)
    counts = np.bincount(root_of, minlength=nIni)
    pos_of_root = {}
    for i in range(nIni):
        if counts[i] == 0:
            continue
        pos_of_root[i] = len(A_box)
        A_box.append((int(np.float32(hX) * np.float32(i)), int(np.float32(hX) * np.float32(i + 1)), 0, H))
        A_cnt.append(int(counts[i]))
    state = np.array([pos_of_root[r] for r in root_of], dtype=np.int64)  # node position in A, or -1 when frozen
    out = []  # (gen, pos, point) for frozen singles; final multis appended at the end
    gen = 0
    for a, c in enumerate(A_cnt):
        if c == 1:
            p = int(np.nonzero(state == a)[0][0])
            out.append((gen, a, p))
            state[p] = -1
    size = len(A_box)
    careful = False
    alive_prev = None  # (gen, box, cnt, positions) of unprocessed multis of the previous generation
    while True:
        prev_size = size
        exp = [a for a, c in enumerate(A_cnt) if c > 1]
        if not exp:
            break
        # per-node child histogram
        nE = len(exp)
        child_cnt = np.zeros((len(A_cnt), 4), dtype=np.int64)
        digit = np.zeros(P, dtype=np.int64)
        for p in range(P):
            a = state[p]
            if a < 0 or A_cnt[a] <= 1:
                continue
            ULx, URx, ULy, BRy = A_box[a]
            halfX = -((-(URx - ULx)) // 2)
            halfY = -((-(BRy - ULy)) // 2)
            d = (0 if xs[p] < ULx + halfX else 1) + (0 if ys[p] < ULy + halfY else 2)
            digit[p] = d
            child_cnt[a, d] += 1
        add = {a: int((child_cnt[a] > 0).sum()) - 1 for a in exp}
        if not careful:
            order = exp  # list order
            m = len(order)
        else:
            order = sorted(exp, key=lambda a: (-A_cnt[a], a))  # size desc, newest (front-most) first
            m = len(order)
            s = size
            for i, a in enumerate(order):
                s += add[a]
                if s >= N:
                    m = i + 1
                    break
        processed = order[:m]
        rank_of = {a: i for i, a in enumerate(processed)}
        # children in creation order: (processing rank, digit)
        keys = []
        for a in processed:
            for d in range(4):
                if child_cnt[a, d] > 0:
                    keys.append((rank_of[a], d, a))
        T = len(keys)
        newA_box = [None] * T
        newA_cnt = [0] * T
        newpos = {}
        for r, (pr, d, a) in enumerate(keys):
            pos = T - 1 - r
            ULx, URx, ULy, BRy = A_box[a]
            halfX = -((-(URx - ULx)) // 2)
            halfY = -((-(BRy - ULy)) // 2)
            bx = (ULx, ULx + halfX) if (d & 1) == 0 else (ULx + halfX, URx)
            by = (ULy, ULy + halfY) if (d & 2) == 0 else (ULy + halfY, BRy)
            newA_box[pos] = (bx[0], bx[1], by[0], by[1])
            newA_cnt[pos] = int(child_cnt[a, d])
            newpos[(a, d)] = pos
        size = prev_size - len(processed) + T
        unprocessed = [a for a in exp if a not in rank_of]
        gen += 1
        alive_prev = (gen - 1, A_box, A_cnt, unprocessed, state.copy())
        newstate = np.full(P, -1, dtype=np.int64)
        for p in range(P):
            a = state[p]
            if a < 0:
                continue
            if a in rank_of:
                np_ = newpos[(a, digit[p])]
                if newA_cnt[np_] == 1:
                    out.append((gen, np_, p))
                else:
                    newstate[p] = np_
            elif A_cnt[a] > 1:
                newstate[p] = -2 - a  # stays in an unprocessed node of the previous generation
        nToExpand = sum(1 for c in newA_cnt if c > 1)
        A_box, A_cnt = newA_box, newA_cnt
        state = newstate
        if size >= N or size == prev_size:
            break
        if unprocessed:
            raise AssertionError("truncated round must be final")
        if not careful and size + 3 * nToExpand > N:
            careful = True
    # emit surviving multi nodes with their best point (max response, first in candidate order)
    def best_of(mask):
        idx = np.nonzero(mask)[0]
        k = sorted(idx, key=lambda p: (-resp[p], ords[p]))
        return int(k[0])
    for a, c in enumerate(A_cnt):
        if c > 1:
            out.append((gen, a, best_of(state == a)))
    if alive_prev is not None:
        g0, _, cnt0, unproc, _ = alive_prev
        for a in unproc:
            out.append((g0, a, best_of(state == (-2 - a))))
    out.sort(key=lambda t: (-t[0], t[1]))
    return [p for (_, _, p) in out]


def main():
    L = ctypes.CDLL("oracle/liborb_oracle.so")
    L.orc_extractor_create.restype = ctypes.c_void_p
    L.orc_extractor_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    h = L.orc_extractor_create(1000, 1.2, 8, 20)
    L.orc_octree.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    ntest = 0
    for trial in range(400):
        W = int(rng.integers(30, 700))
        H = int(rng.integers(max(30, W // 3), min(700, 2 * W - 1)))
        if round(W / H) < 1:
            continue
        P = int(rng.integers(0, 900))
        N = int(rng.integers(1, 300))
        mode = trial % 4
        if mode == 0:
            xs = rng.integers(3, W - 3, size=P)
            ys = rng.integers(3, H - 3, size=P)
        elif mode == 1:  # clustered
            cx, cy = rng.integers(3, W - 3), rng.integers(3, H - 3)
            xs = np.clip(cx + rng.normal(0, 6, P).astype(int), 3, W - 4)
            ys = np.clip(cy + rng.normal(0, 6, P).astype(int), 3, H - 4)
        elif mode == 2:  # few distinct responses -> many ties
            xs = rng.integers(3, W - 3, size=P)
            ys = rng.integers(3, H - 3, size=P)
        else:  # lines
            xs = rng.integers(3, W - 3, size=P)
            ys = np.full(P, rng.integers(3, H - 3))
        pts = np.unique(np.stack([xs, ys], 1), axis=0)
        rng.shuffle(pts)
        P = len(pts)
        xs, ys = pts[:, 0].astype(np.int64), pts[:, 1].astype(np.int64)
        resp = rng.integers(1, 4 if mode == 2 else 200, size=P).astype(np.int64)
        kp = np.zeros((max(P, 1), 7), dtype=np.float32)
        kp[:P, 0] = xs
        kp[:P, 1] = ys
        kp[:P, 4] = resp
        kp[:P, 2] = np.arange(P)  # carry the index in `size`
        outk = np.zeros((N + 8 + P, 7), dtype=np.float32)
        n = L.orc_octree(h, kp.ctypes.data, P, 13, 13 + W, 13, 13 + H, N, outk.ctypes.data, len(outk))
        ref = [int(v) for v in outk[:n, 2]]
        got = octree_parallel(xs, ys, resp, np.arange(P), W, H, N) if P else []
        if ref != got:
            print("MISMATCH trial", trial, "W,H,P,N", W, H, P, N, "mode", mode)
            print(" ref", ref[:20], len(ref))
            print(" got", got[:20], len(got))
            sys.exit(1)
        ntest += 1
    print("ok", ntest, "cases")


if __name__ == "__main__":
    main()
