#!/usr/bin/env python3
"""Prototype of the closed-form ("count pyramid") formulation of ORBextractor::DistributeOctTree
(src/ORBextractor.cc:1006-1287) used by the HIP kernel, checked against the literal std::list restatement in oracle/.

A point's whole root-to-leaf path is a pure function of its coordinates and the root box (DivideNode halves a box at
ceil(size / 2)), so the node a point sits in at generation g is the length-g prefix of its path key.  With the number of points per
prefix known for every depth (one histogram at the deepest level + sums of four), everything the full passes (:1061-1132) decide --
list size after every pass, how many nodes are expandable, where the careful phase starts -- is arithmetic on those counts; no pass
over the points and no barrier per generation is needed until the careful rounds, which work on <= N nodes.

List order.  Children are pushed to the FRONT in the order n1..n4 while the parents are visited front to back, so the list order of a
generation made by full passes is: last digit descending, the digits above alternating, the root like digit 1.  With the even
digits complemented ("T key") generation g reads ascending in T for even g and descending for odd g.
Run:  python tools/octree_pyramid_proto.py [seed]   (needs oracle/liborb_oracle.so)
"""
import ctypes
import sys

import numpy as np

D_MAX = 12


def path_digits(xs, ys, W, H):
    """-> root[P], digits[P][D_MAX] (n1..n4 = 0..3: right -> +1, bottom -> +2), exactly as DivideNode walks."""
    P = len(xs)
    nIni = int(np.floor(np.float32(W) / np.float32(H) + np.float32(0.5)))
    hX = np.float32(W) / np.float32(nIni)
    root = (xs.astype(np.float32) / hX).astype(np.int64)
    ulx = (hX * root.astype(np.float32)).astype(np.int64)
    urx = (hX * (root + 1).astype(np.float32)).astype(np.int64)
    uly = np.zeros(P, np.int64)
    bry = np.full(P, H, np.int64)
    dig = np.zeros((P, D_MAX), np.int64)
    for g in range(D_MAX):
        hx = (urx - ulx + 1) // 2
        hy = (bry - uly + 1) // 2
        right = xs >= ulx + hx
        bottom = ys >= uly + hy
        dig[:, g] = right + 2 * bottom
        ulx = np.where(right, ulx + hx, ulx)
        urx = np.where(right, urx, ulx + hx)
        uly = np.where(bottom, uly + hy, uly)
        bry = np.where(bottom, bry, uly + hy)
    return nIni, root, dig


def tkey(root, dig, depth):
    """T key of the length-`depth` prefix: digit k (1-based) complemented when k is even; root is the most significant digit."""
    k = root.copy()
    for g in range(depth):
        d = dig[:, g]
        if (g + 1) % 2 == 0:
            d = 3 - d
        k = k * 4 + d
    return k


def octree_pyramid(xs, ys, resp, ords, W, H, N, stats=None):
    P = len(xs)
    if P == 0:
        return []
    nIni, root, dig = path_digits(xs, ys, W, H)
    keys = [tkey(root, dig, g) for g in range(D_MAX + 1)]          # per depth: T key of every point
    cnt = [dict(zip(*np.unique(k, return_counts=True))) for k in keys]   # per depth: T key -> points

    def exists(g, b):      # node b of depth g was created: its parent was split
        return g == 0 or cnt[g - 1][b >> 2] > 1

    def nodes_born(g):     # T keys of the generation-g nodes, in LIST order
        ks = sorted(b for b in cnt[g] if exists(g, b))
        return ks if g % 2 == 0 else ks[::-1]

    # ---- full passes in closed form ----
    out = []               # (gen, pos, depth, tkey): final nodes
    g = 0
    born = nodes_born(0)
    size = len(born)
    careful = False
    while True:
        for pos, b in enumerate(born):
            if cnt[g][b] == 1:
                out.append((g, pos, g, b))
        exp = [b for b in born if cnt[g][b] > 1]
        if not exp:
            A = []
            break
        prev = size
        nxt = nodes_born(g + 1)
        size = size - len(exp) + len(nxt)
        g += 1
        born = nxt
        if size >= N or size == prev:
            for pos, b in enumerate(born):
                if cnt[g][b] == 1:
                    out.append((g, pos, g, b))
            A = [(pos, b) for pos, b in enumerate(born) if cnt[g][b] > 1]
            break
        nexp = sum(1 for b in born if cnt[g][b] > 1)
        if size + 3 * nexp > N:
            careful = True
            for pos, b in enumerate(born):
                if cnt[g][b] == 1:
                    out.append((g, pos, g, b))
            A = [(pos, b) for pos, b in enumerate(born) if cnt[g][b] > 1]
            break
    full_gens = g
    rounds = 0
    # ---- careful rounds on the <= N expandable nodes A of generation g (pos, tkey) ----
    if careful:
        while True:
            rounds += 1
            prev = size
            depth = g      # generation == depth for every live node
            order = sorted(A, key=lambda t: (-cnt[depth][t[1]], t[0]))

            def kids(b):   # non-empty children in n1..n4 order -> their T keys
                res = []
                for d in range(4):
                    td = 3 - d if (depth + 1) % 2 == 0 else d
                    c = 4 * b + td
                    if c in cnt[depth + 1]:
                        res.append(c)
                return res
            m = len(order)
            s = size
            for i, (pos, b) in enumerate(order):
                s += len(kids(b)) - 1
                if s >= N:
                    m = i + 1
                    break
            processed, unprocessed = order[:m], order[m:]
            created = []
            for (pos, b) in processed:
                created += kids(b)
            T = len(created)
            size = prev - len(processed) + T
            newA = []
            for r, c in enumerate(created):
                pos = T - 1 - r
                if cnt[depth + 1][c] == 1:
                    out.append((g + 1, pos, depth + 1, c))
                else:
                    newA.append((pos, c))
            for (pos, b) in unprocessed:
                out.append((g, pos, depth, b))
            g += 1
            A = sorted(newA)
            if size >= N or size == prev or not A:
                break
            assert not unprocessed
    for (pos, b) in A:
        out.append((g, pos, g, b))
    # ---- one point per final node: max response, first in candidate order ----
    out.sort(key=lambda t: (-t[0], t[1]))
    res = []
    for (gen, pos, depth, b) in out:
        idx = np.nonzero(keys[depth] == b)[0]
        best = sorted(idx, key=lambda p: (-resp[p], ords[p]))[0]
        res.append(int(best))
    if stats is not None:
        stats.append((P, N, full_gens, rounds, max(t[2] for t in out)))
    return res


def main():
    L = ctypes.CDLL("oracle/liborb_oracle.so")
    L.orc_extractor_create.restype = ctypes.c_void_p
    L.orc_extractor_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    h = L.orc_extractor_create(1000, 1.2, 8, 20)
    L.orc_octree.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    ntest = 0
    for trial in range(600):
        W = int(rng.integers(30, 1900))
        H = int(rng.integers(max(30, W // 4), min(1100, 2 * W - 1)))
        if round(W / H) < 1:
            continue
        P = int(rng.integers(0, 1500))
        N = int(rng.integers(1, 450))
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
        got = octree_pyramid(xs, ys, resp, np.arange(P), W, H, N) if P else []
        if ref != got:
            print("MISMATCH trial", trial, "W,H,P,N", W, H, P, N, "mode", mode)
            print(" ref", ref[:20], len(ref))
            print(" got", got[:20], len(got))
            sys.exit(1)
        ntest += 1
    print("ok", ntest, "cases")


if __name__ == "__main__":
    main()
