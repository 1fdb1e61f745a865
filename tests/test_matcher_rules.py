"""Matcher half of the oracle: hand-checkable rules of src/ORBmatcher.cc and src/FrameKTL.cc."""
import numpy as np

import oracle_lib


def test_descriptor_distance_is_popcount(oracle):
    rng = np.random.default_rng(0)
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0 and oracle.descriptor_distance(z, o) == 256 and oracle.descriptor_distance(o, o) == 0
    for _ in range(500):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    one = z.copy()
    one[31] = 0x80
    assert oracle.descriptor_distance(z, one) == 1


def test_knn2_ties_keep_lower_index_and_short_sets(oracle):
    q = np.zeros((2, 32), np.uint8)
    t = np.zeros((4, 32), np.uint8)
    t[0, 0] = 0b11       # d=2
    t[1, 0] = 0b01       # d=1
    t[2, 0] = 0b10       # d=1  (tie with 1 -> index 1 is best, 2 is second)
    t[3, 0] = 0b111      # d=3
    idx0, d0, idx1, d1 = oracle.knn2(q, t)
    assert idx0.tolist() == [1, 1] and idx1.tolist() == [2, 2] and d0.tolist() == [1, 1] and d1.tolist() == [1, 1]
    idx0, d0, idx1, d1 = oracle.knn2(q, t[:1])
    assert idx0.tolist() == [0, 0] and idx1.tolist() == [-1, -1] and d1.tolist() == [-1, -1]
    mask = np.array([[0, 0, 0, 1], [0, 0, 0, 0]], np.uint8)
    idx0, d0, idx1, d1 = oracle.knn2(q, t, mask)
    assert idx0.tolist() == [3, -1] and idx1.tolist() == [-1, -1]


def _kps(xy, octave):
    k = np.zeros(len(xy), oracle_lib.KP)
    k["x"], k["y"], k["octave"] = [p[0] for p in xy], [p[1] for p in xy], octave
    return k


def test_features_in_area_against_linear_scan(oracle):
    rng = np.random.default_rng(1)
    n = 1500
    kp = _kps(rng.uniform(0, 752, (n, 2)).astype(np.float32) * [1, 480 / 752], rng.integers(0, 8, n))
    bounds = (0, 0, 752, 480)
    invw, invh = np.float32(64) / np.float32(752), np.float32(48) / np.float32(480)
    px = np.floor((kp["x"] - np.float32(0)) * invw + np.float32(0.5)).astype(int)  # roundf for non-negative values
    py = np.floor((kp["y"] - np.float32(0)) * invh + np.float32(0.5)).astype(int)
    ingrid = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    for _ in range(200):
        x, y = np.float32(rng.uniform(-20, 780)), np.float32(rng.uniform(-20, 500))
        r = np.float32(rng.uniform(1, 60))
        lo = int(rng.integers(-1, 7))
        hi = lo if rng.random() < 0.3 else (lo + 1 if lo >= 0 else int(rng.integers(-1, 2)))
        got = oracle.features_in_area(kp, bounds, x, y, r, lo, hi).tolist()
        c0x, c1x = max(0, int(np.floor((x - r) * invw))), min(63, int(np.ceil((x + r) * invw)))
        c0y, c1y = max(0, int(np.floor((y - r) * invh))), min(47, int(np.ceil((y + r) * invh)))
        ref = []
        if c0x < 64 and c1x >= 0 and c0y < 48 and c1y >= 0:
            for ix in range(c0x, c1x + 1):
                for iy in range(c0y, c1y + 1):
                    for i in np.nonzero(ingrid & (px == ix) & (py == iy))[0]:
                        o = kp["octave"][i]
                        if lo == -1 and hi == -1:
                            pass
                        elif lo == hi:
                            if o != lo:
                                continue
                        elif o < lo or o > hi:
                            continue
                        if abs(kp["x"][i] - x) > r or abs(kp["y"][i] - y) > r:
                            continue
                        ref.append(int(i))
        assert got == ref


def test_grid_uses_round_not_floor(oracle):
    # a keypoint at x = 11.74*0.6 cells ... : cell width 752/64 = 11.75 px; x = 6.0 -> round(0.51) = cell 1, floor would say 0
    kp = _kps([(6.0, 5.0)], [0])
    # query window that covers only cell column 0 (x - r >= 0, x + r < 11.75 -> ceil gives 1, so column 1 is visited too)
    assert oracle.features_in_area(kp, (0, 0, 752, 480), 3.0, 5.0, 4.0, -1, -1).tolist() == [0]
    # a keypoint whose rounded cell is 64 (x = 751.9) is never inserted (PosInGrid returns false)
    kp = _kps([(751.9, 5.0)], [0])
    assert oracle.features_in_area(kp, (0, 0, 752, 480), 750.0, 5.0, 5.0, -1, -1).tolist() == []


def _sbp(oracle, kp, desc, mp, th=1.0, ratio=0.8, assigned=None):
    px, py, lvl, vc, inv, md = mp
    a = np.full(len(kp), -1, np.int32) if assigned is None else assigned
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    n = oracle.search_by_projection(kp, desc, (0, 0, 752, 480), a, px, py, lvl, vc, inv, md, sf, th, ratio)
    return n, a


def test_search_by_projection_rules(oracle):
    d = np.zeros((4, 32), np.uint8)
    d[1, 0] = 0xFF           # 8 bits away from descriptor 0
    d[2, :13] = 0xFF         # 104 bits away: above TH_HIGH = 100
    kp = _kps([(100, 100), (101, 100), (300, 300), (102, 101)], [2, 2, 2, 1])
    d[3, :2] = 0xFF          # 16 bits
    one = lambda **k: (np.array([k["x"]], np.float32), np.array([k["y"]], np.float32), np.array([k["l"]], np.int32),
                       np.array([k.get("vc", 0.9)], np.float32), np.array([k.get("iv", 1)], np.uint8), k["d"][None, :])
    # best = kp0 (0), second = kp1 (8) on the same octave: 0 > 0.8*8 is false -> accepted
    n, a = _sbp(oracle, kp, d, one(x=100.5, y=100, l=2, d=d[0]))
    assert n == 1 and a.tolist() == [0, -1, -1, -1]
    # descriptor half way: best 4 bits (kp0), second 4 bits (kp1) same octave -> 4 > 0.8*4 -> rejected by the ratio test
    q = np.zeros(32, np.uint8)
    q[0] = 0x0F
    n, a = _sbp(oracle, kp, d, one(x=100.5, y=100, l=2, d=q))
    assert n == 0
    # same distances but the runner-up sits on another octave -> no ratio test (:116)
    kp2 = kp.copy()
    kp2["octave"][1] = 1
    n, a = _sbp(oracle, kp2, d, one(x=100.5, y=100, l=2, d=q))
    assert n == 1 and a[0] == 0
    # TH_HIGH: only candidate is 104 bits away
    n, a = _sbp(oracle, kp, d, one(x=300, y=300, l=2, d=d[0]))
    assert n == 0
    # not in view -> skipped
    n, a = _sbp(oracle, kp, d, one(x=100.5, y=100, l=2, d=d[0], iv=0))
    assert n == 0
    # level filter: predicted level 4 looks at octaves 3..4 only
    n, a = _sbp(oracle, kp, d, one(x=100.5, y=100, l=4, d=d[0]))
    assert n == 0
    # window radius: viewCos > 0.998 -> 2.5*scale[2] = 3.6 px, else 4*1.44 = 5.76 px
    n, a = _sbp(oracle, kp, d, one(x=106, y=100, l=2, d=d[0], vc=0.999))
    assert n == 0
    n, a = _sbp(oracle, kp, d, one(x=106, y=100, l=2, d=d[0], vc=0.9))   # kp1 (5 px) and kp3 (4 px, octave 1) now inside
    assert n == 1 and a.tolist() == [-1, 0, -1, -1]
    # greedy exclusivity: two identical map points -> the first takes kp0; the second skips it and takes kp1 (8 bits), its
    # runner-up kp3 sits on another octave so no ratio test applies
    two = (np.array([100.5, 100.5], np.float32), np.array([100, 100], np.float32), np.array([2, 2], np.int32), np.array([0.9, 0.9], np.float32),
           np.array([1, 1], np.uint8), np.stack([d[0], d[0]]))
    n, a = _sbp(oracle, kp, d, two)
    assert n == 2 and a.tolist() == [0, 1, -1, -1]
    # keypoints that already hold a map point are skipped (:91)
    pre = np.array([77, -1, -1, -1], np.int32)
    n, a = _sbp(oracle, kp, d, one(x=100.5, y=100, l=2, d=d[0]), assigned=pre)
    assert n == 1 and a.tolist()[:2] == [77, 0]


def test_distinctive_descriptor_by_hand(oracle):
    """src/MapPoint.cc:250-263: median = sorted row [int(0.5*(N-1))] with the self distance 0 included; first index wins ties."""
    z = np.zeros(32, np.uint8)
    def bits(n):
        d = z.copy()
        d[: n // 8] = 0xFF
        if n % 8:
            d[n // 8] = (1 << (n % 8)) - 1
        return d
    # four descriptors on a line: 0, 10, 20, 100 bits set (nested) -> distances are differences
    D = np.stack([bits(0), bits(10), bits(20), bits(100)])
    # rows sorted: [0,10,20,100] [0,10,10,90] [0,10,20,80] [0,80,90,100]; index int(0.5*3) = 1 -> medians 10,10,10,80 -> first = 0
    assert oracle.distinctive_descriptor(D) == (0, 10)
    # N = 3: index 1 of the sorted row: [0,10,20] -> 10, [0,10,10] -> 10, [0,10,20] -> 10 -> first
    assert oracle.distinctive_descriptor(D[:3]) == (0, 10)
    # N = 5 with a clear centre: index 2
    D5 = np.stack([bits(0), bits(40), bits(50), bits(60), bits(100)])
    # row of bits(50): [0,10,10,50,50] -> median 10; others: bits(40): [0,10,20,40,60] -> 20; bits(60): 20; ends: 50
    assert oracle.distinctive_descriptor(D5) == (2, 10)
    assert oracle.distinctive_descriptor(D[:1]) == (0, 0)
