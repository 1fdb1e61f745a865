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


# ---- rotation histogram / ComputeThreeMaxima (src/ORBmatcher.cc:1748-1789) ---------------------------------------------
def test_compute_three_maxima_by_hand(oracle):
    s = [0] * 30
    assert oracle.compute_three_maxima(s) == [-1, -1, -1]
    s[4], s[7], s[9], s[11] = 50, 20, 6, 5
    assert oracle.compute_three_maxima(s) == [4, 7, 9]
    s[9] = 4                                 # bin 11 (5) is now third: 5 < 0.1f * 50 is false -> kept
    assert oracle.compute_three_maxima(s) == [4, 7, 11]
    s[11] = 3                                # third = 4 < 0.1 * 50 -> dropped
    assert oracle.compute_three_maxima(s) == [4, 7, -1]
    s[7] = 4                                 # second < 0.1 * max -> second and third dropped
    assert oracle.compute_three_maxima(s)[1:] == [-1, -1]
    t = [0] * 30
    t[2] = t[5] = t[8] = 10                  # ties: strict '>' keeps the first seen on top
    assert oracle.compute_three_maxima(t) == [2, 5, 8]


def _desc_with_distance(rng, base, d):
    out = np.unpackbits(base).copy()
    flip = rng.choice(256, d, replace=False)
    out[flip] ^= 1
    return np.packbits(out)


def test_search_by_bow_rules_by_hand(oracle):
    """Thresholds (<= TH_LOW for KF-Frame, < TH_LOW for KF-KF), ratio against the second best of the same node,
    first-come exclusivity, usable flags."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    far = base ^ np.uint8(255)
    d1 = np.stack([base, base, base])                                  # three queries, all the same descriptor
    d2 = np.stack([_desc_with_distance(rng, base, 50), _desc_with_distance(rng, base, 10), far, _desc_with_distance(rng, base, 30)])
    ang1, ang2 = np.zeros(3, np.float32), np.zeros(4, np.float32)
    g1 = {7: [0, 1], 9: [2]}
    g2 = {7: [1, 0], 8: [2], 9: [3]}
    # KF-Frame, ratio 0.9: q0 -> t1 (10 < 0.9*50); q1: t1 taken, best t0 at 50 <= TH_LOW, second INT_MAX -> t0; q2 (node 9) -> t3
    m, n = oracle.search_by_bow(False, g1, d1, ang1, [1, 1, 1], g2, d2, ang2, None, 0.9, False)
    assert m.tolist() == [1, 0, 3] and n == 3
    # KF-KF: same but '< TH_LOW' rejects the distance-50 match
    m, n = oracle.search_by_bow(True, g1, d1, ang1, [1, 1, 1], g2, d2, ang2, [1, 1, 1, 1], 0.9, False)
    assert m.tolist() == [1, -1, 3] and n == 2
    # unusable query 0 (no map point): query 1 now gets t1; unusable target 3 in KF-KF mode
    m, n = oracle.search_by_bow(True, g1, d1, ang1, [0, 1, 1], g2, d2, ang2, [1, 1, 1, 0], 0.9, False)
    assert m.tolist() == [-1, 1, -1] and n == 1
    # ratio: 10 < 0.15 * 50 fails
    m, n = oracle.search_by_bow(False, {7: [0]}, d1[:1], ang1[:1], [1], g2, d2, ang2, None, 0.15, False)
    assert m.tolist() == [-1] and n == 0


def test_rotation_bins_reach_only_0_to_12(oracle):
    """bin = round(rot / 30) with rot in [0, 360): bins 0..12 (upstream quirk); the three fullest survive."""
    rng = np.random.default_rng(6)
    n = 40
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d1, d2 = base.copy(), base.copy()                                   # query i matches target i exactly
    groups = {i: [i] for i in range(n)}
    ang1 = np.zeros(n, np.float32)
    ang2 = np.zeros(n, np.float32)
    # rot = ang1 - ang2 (+360): 20 matches at rot 0 (bin 0), 10 at 90 (bin 3), 6 at 180 (bin 6), 4 at 359 (bin 12)
    ang2[20:30], ang2[30:36], ang2[36:40] = 270, 180, 1
    m, nm = oracle.search_by_bow(False, groups, d1, ang1, np.ones(n), groups, d2, ang2, None, 0.9, True)
    assert nm == 36 and (m[:36] == np.arange(36)).all() and (m[36:] == -1).all()


def test_search_for_triangulation_by_hand(oracle):
    """Sorted by (distance, index), cut at 2 * best, first candidate passing the epipolar test wins."""
    rng = np.random.default_rng(7)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    kp1 = _kps([(100, 100)], [0])
    kp2 = _kps([(100, 140), (100, 100.5), (300, 100), (100, 99.5)], [0, 0, 0, 0])
    d1 = base[None]
    d2 = np.stack([_desc_with_distance(rng, base, 10), _desc_with_distance(rng, base, 15), _desc_with_distance(rng, base, 20),
                   _desc_with_distance(rng, base, 21)])
    # F12 such that the epipolar line of (x1, y1) in image 2 is y = y1:  l = x1' F12 = (0, 1, -y1)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    sig = np.ones(8, np.float32)
    g = {3: [0]}
    g2 = {3: [0, 1, 2, 3]}
    # best = 10 (t0, off the line by 40 px) -> DistTh 20: t1 (15, 0.5 px off: 0.25 < 3.84) is the first to pass
    m, n = oracle.search_for_triangulation(g, kp1, d1, [0], g2, kp2, d2, [0, 0, 0, 0], F12, sig, False)
    assert m.tolist() == [1] and n == 1
    # t1 already has a map point: t2 (20 <= DistTh) lies on the line; t3 (21) would be past the cut
    m, n = oracle.search_for_triangulation(g, kp1, d1, [0], g2, kp2, d2, [0, 1, 0, 0], F12, sig, False)
    assert m.tolist() == [2]
    m, n = oracle.search_for_triangulation(g, kp1, d1, [0], g2, kp2, d2, [0, 1, 1, 0], F12, sig, False)
    assert m.tolist() == [-1] and n == 0                              # t3 is beyond 2 * best
    m, n = oracle.search_for_triangulation(g, kp1, d1, [1], g2, kp2, d2, [0, 0, 0, 0], F12, sig, False)
    assert m.tolist() == [-1]                                          # keypoint 1 already has a map point


def test_projection_kf_and_fuse_by_hand(oracle):
    rng = np.random.default_rng(8)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    kp = _kps([(100, 100), (103, 100), (100, 104), (180, 100)], [1, 2, 4, 1])
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    desc = np.stack([_desc_with_distance(rng, base, 40), _desc_with_distance(rng, base, 20), _desc_with_distance(rng, base, 5),
                     _desc_with_distance(rng, base, 0)])
    bounds = (0, 0, 752, 480)
    u, v = [100.0, 100.0], [100.0, 100.0]
    mpd = np.stack([base, base])
    # level 2, th 3 -> radius 4.32 px, levels 1..3: kp0 (40), kp1 (20) are candidates (kp2 is level 4, kp3 far) -> kp1; the
    # second map point then takes kp0
    a = np.full(4, -1, np.int32)
    n = oracle.search_by_projection_kf(kp, desc, bounds, a, u, v, [2, 2], [1, 1], mpd, [0, 0], sf, 3.0, 100, False)
    assert n == 2 and a.tolist() == [1, 0, -1, -1]
    a = np.full(4, -1, np.int32)
    n = oracle.search_by_projection_kf(kp, desc, bounds, a, u, v, [2, 2], [1, 1], mpd, [0, 0], sf, 3.0, 30, False)
    assert n == 1 and a.tolist() == [-1, 0, -1, -1]                    # ORBdist 30: the distance-40 match is refused
    # Fuse: levels [l-1, l], TH_LOW, no exclusivity -> both map points get kp1
    bi, bd = oracle.fuse_search(kp, desc, bounds, u, v, [2, 2], [1, 1], mpd, sf, 3.0)
    assert bi.tolist() == [1, 1] and bd.tolist() == [20, 20]
    bi, bd = oracle.fuse_search(kp, desc, bounds, u, v, [5, 2], [1, 0], mpd, sf, 3.0)
    assert bi.tolist() == [2, -1] and bd.tolist() == [5, -1]          # level 5 -> levels 4..5, radius 7.46


def _cam(R=np.eye(3), t=(0, 0, 0), Ow=(0, 0, 0), fx=500.0, fy=500.0, cx=376.0, cy=240.0, bounds=(0, 752, 0, 480)):
    return np.concatenate([np.asarray(R, np.float32).reshape(9), np.asarray(t, np.float32), np.asarray(Ow, np.float32),
                           np.asarray([fx, fy, cx, cy], np.float32), np.asarray(bounds, np.float32)])


def test_projection_prologues_by_hand(oracle):
    """isInFrustum / PredictScale, the SearchByProjection(F, pKF) prologue and the Fuse prologue on an identity pose."""
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    cam = _cam()
    #          on axis       right of axis   behind      outside image   too far        grazing normal
    P = [[0, 0, 10.0], [1.0, 0, 10.0], [0, 0, -5.0], [20.0, 0, 10.0], [0, 0, 100.0], [0, 0, 10.0]]
    Pn = [[0, 0, 1.0]] * 5 + [[1.0, 0, 0.2]]
    mn, mx = np.full(6, 5.0, np.float32), np.full(6, 20.0, np.float32)
    valid, u, v, lvl, vc = oracle.project_points(0, cam, P, Pn, mn, mx, None, sf)
    assert valid.tolist() == [1, 1, 0, 0, 0, 0]
    assert (u[0], v[0], u[1]) == (376.0, 240.0, 426.0) and vc[0] == 1.0
    # PredictScale: ratio = mfMax / dist = 2 -> ceil(log 2 / log 1.2) = ceil(3.80) = 4
    assert lvl[0] == 4 and lvl[1] == 4
    # mode 1 has no depth-sign, distance or normal test; level = lower_bound(dist / (0.8 * mfMin)): 10 / 4 = 2.5 -> 1.2^6 = 2.99 -> 6
    valid, u, v, lvl, _ = oracle.project_points(1, cam, P, None, mn, mx, None, sf)
    assert valid.tolist() == [1, 1, 1, 0, 1, 1] and lvl[0] == 6 and lvl[4] == 7 and (u[2], v[2]) == (376.0, 240.0)
    # mode 2 (Fuse): depth, image, distance window [4, 24], dot(PO, Pn) >= 0.5 * dist
    valid, u, v, lvl, _ = oracle.project_points(2, cam, P, Pn, mn, mx, [1, 1, 1, 1, 1, 1], sf)
    assert valid.tolist() == [1, 1, 0, 0, 0, 0] and lvl[0] == 6
    valid, *_ = oracle.project_points(2, cam, P, Pn, mn, mx, [0, 1, 1, 1, 1, 1], sf)
    assert valid.tolist() == [0, 1, 0, 0, 0, 0]


def test_sim3_forms_by_hand(oracle):
    """Loop-closing forms: the Scw decomposition (:299-303), the relative transforms of SearchBySim3 (:1284-1287), its per-point
    prologue, the exclusive search of SearchByProjection(pKF, Scw, ...) and the mutual-agreement rule."""
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    # Scw = [2 R | 2 t] with R = rotation by 90 deg about z, t = (1, 2, 3): scale 2 comes out, Ow = -R^T t
    R = np.float32([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
    Scw = np.eye(4, dtype=np.float32)
    Scw[:3, :3], Scw[:3, 3] = 2 * R, [2, 4, 6]
    r, t, o = oracle.sim3_decompose(Scw)
    np.testing.assert_array_equal(r.reshape(3, 3), R)
    assert t.tolist() == [1.0, 2.0, 3.0] and o.tolist() == [-2.0, 1.0, -3.0]
    # relative transforms: s12 = 2, R12 = R, t12 = (1, 0, 0): sR21 = R^T / 2, t21 = -(R^T / 2) t12 = (0, 0.5, 0)
    a, b, c = oracle.sim3_relative(2.0, R, [1, 0, 0])
    np.testing.assert_array_equal(a, 2 * R)
    np.testing.assert_array_equal(b, R.T / 2)
    assert c.tolist() == [0.0, 0.5, 0.0]
    # prologue with identity transforms: like the Fuse prologue without the normal test, distance = norm of the camera-frame point
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    P = [[0, 0, 10.0], [1.0, 0, 10.0], [0, 0, -5.0], [20.0, 0, 10.0], [0, 0, 100.0]]
    mn, mx = np.full(5, 5.0, np.float32), np.full(5, 20.0, np.float32)
    valid, u, v, lvl = oracle.project_sim3(I, z, I, z, _cam(), P, mn, mx, None, sf)
    assert valid.tolist() == [1, 1, 0, 0, 0] and (u[0], v[0], u[1]) == (376.0, 240.0, 426.0) and lvl[0] == 6
    # the chain really is other <- own <- world: own = translate by (0, 0, 5), other = scale by 2 -> depth 30, u = 376 + 500 * 2 / 30
    valid, u, v, lvl = oracle.project_sim3(I, [0, 0, 5.0], 2 * I, z, _cam(), [[1.0, 0, 10.0]], [10.0], [40.0], None, sf)
    assert valid.tolist() == [1] and u[0] == np.float32(376.0) + np.float32(500.0) * (np.float32(2.0) * np.float32(np.float32(1.0) / np.float32(30.0)))
    # exclusive search: key points 0 (d 40), 1 (d 20) at level 1/2, both map points prefer kp1; the second must settle for kp0
    rng = np.random.default_rng(9)
    kp = _kps([(100, 100), (103, 100), (100, 104), (180, 100)], [1, 2, 4, 1])
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    desc = np.stack([_desc_with_distance(rng, base, 40), _desc_with_distance(rng, base, 20), _desc_with_distance(rng, base, 5),
                     _desc_with_distance(rng, base, 0)])
    bounds = (0, 0, 752, 480)
    mpd = np.stack([base, base])
    matched = np.full(4, -1, np.int32)
    n = oracle.search_by_projection_sim3(kp, desc, bounds, matched, [100.0, 100.0], [100.0, 100.0], [2, 2], [1, 1], mpd, sf, 3)
    assert n == 2 and matched.tolist() == [1, 0, -1, -1]
    matched = np.int32([-1, 7, -1, -1])                                   # kp1 already holds a point: both candidates see only kp0
    n = oracle.search_by_projection_sim3(kp, desc, bounds, matched, [100.0, 100.0], [100.0, 100.0], [2, 2], [1, 1], mpd, sf, 3)
    assert n == 1 and matched.tolist() == [0, 7, -1, -1]
    # SearchBySim3: two key frames with the same two key points; direction 1->2 maps point 0 -> kp 0 and point 1 -> kp 0 (TH_HIGH,
    # no exclusivity), direction 2->1 maps kp 0's point -> 0: only the mutual pair (0, 0) survives
    kpA = _kps([(100, 100), (300, 100)], [1, 1])
    dA = np.stack([base, _desc_with_distance(rng, base, 90)])
    proj12 = ([1, 1], [100.0, 100.0], [100.0, 100.0], [1, 1])
    proj21 = ([1, 0], [100.0, 0.0], [100.0, 0.0], [1, 0])
    m12, nf = oracle.search_by_sim3(kpA, dA, bounds, kpA, dA, bounds, proj12, np.stack([base, base]), proj21, np.stack([base, base]), sf, sf, 7.5)
    assert nf == 1 and m12.tolist() == [0, -1]
    # beyond TH_HIGH nothing is kept
    far = np.stack([_desc_with_distance(rng, base, 101), _desc_with_distance(rng, base, 120)])
    m12, nf = oracle.search_by_sim3(kpA, dA, bounds, kpA, dA, bounds, proj12, far, proj21, far, sf, sf, 7.5)
    assert nf == 0 and m12.tolist() == [-1, -1]


# ---- DBoW2 transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1125-1258) ---------------------------------------------
def _toy_vocabulary(weighting=0, normalize=1):
    """Root 0 -> {1, 2}; 1 -> leaves {3, 4}; 2 -> leaf {5} and inner 6 -> leaves {7, 8}.  Descriptors: byte 0 carries the value."""
    d = np.zeros((9, 32), np.uint8)
    d[1, 0], d[2, 0] = 0x00, 0xFF
    d[3, 0], d[4, 0] = 0x00, 0x0F
    d[5, 0], d[6, 0] = 0xFF, 0xF0
    d[7, 0], d[8, 0] = 0xF0, 0xF1
    return dict(child_start=[0, 2, 4, 6, 6, 6, 6, 8, 8, 8], children=[1, 2, 3, 4, 5, 6, 7, 8], descriptor=d,
                word_id=[-1, -1, -1, 0, 1, 2, -1, 3, 4], weight=[0, 0, 0, 1.0, 2.0, 0.0, 0, 0.5, 4.0], L=3, weighting=weighting,
                normalize=normalize)


def test_bow_transform_by_hand(oracle):
    voc = _toy_vocabulary()
    f = np.zeros((5, 32), np.uint8)
    f[:, 0] = [0x01, 0x0E, 0xFF, 0xF1, 0x0F]
    # f0 = 0x01: level 1 -> node 1 (d 1 vs 7); level 2: node 3 (d 1) vs node 4 (0x0F: d 3) -> word 0
    # f1 = 0x0E: node 1 (3 vs 5); node 3 (d 3) vs node 4 (d 1) -> word 1
    # f2 = 0xFF: node 2; node 5 (d 0) vs node 6 (d 4) -> word 2, weight 0: stopped, enters neither container
    # f3 = 0xF1: node 2 (d 3 vs 5 for node 1); node 5 (0xFF: d 3) vs node 6 (0xF0: d 1) -> 6; node 7 (d 1) vs node 8 (d 0) -> word 4
    # f4 = 0x0F: node 1 (d 4) vs node 2 (d 4): tie -> first child, node 1; node 3 (d 4) vs node 4 (d 0) -> word 1
    wid, ww, nid, (bid, bval), fv = oracle.bow_transform(voc, f, levelsup=2)     # nid level = L - 2 = 1
    assert wid.tolist() == [0, 1, 2, 4, 1] and ww.tolist() == [1.0, 2.0, 0.0, 4.0, 2.0]
    assert nid.tolist() == [1, 1, 2, 2, 1]
    assert bid.tolist() == [0, 1, 4]
    np.testing.assert_array_equal(bval, np.array([1.0, 4.0, 4.0]) / 9.0)            # TF-IDF accumulates, then L1
    assert fv == {1: [0, 1, 4], 2: [3]}
    # levelsup = 1 -> nid level 2: f2 / f3 differ (5 is a leaf at level 2, 6 an inner node)
    _, _, nid, _, fv = oracle.bow_transform(voc, f, levelsup=1)
    assert nid.tolist() == [3, 4, 5, 6, 4] and fv == {3: [0], 4: [1, 4], 6: [3]}
    # levelsup >= L -> root for everybody
    _, _, nid, _, fv = oracle.bow_transform(voc, f, levelsup=3)
    assert nid.tolist() == [0, 0, 0, 0, 0] and fv == {0: [0, 1, 3, 4]}
    # BINARY weighting without normalisation: first occurrence only
    _, _, _, (bid, bval), _ = oracle.bow_transform(_toy_vocabulary(3, 0), f, levelsup=2)
    assert bid.tolist() == [0, 1, 4] and bval.tolist() == [1.0, 2.0, 4.0]
    # TF without normalisation divides by the number of distinct words
    _, _, _, (bid, bval), _ = oracle.bow_transform(_toy_vocabulary(1, 0), f, levelsup=2)
    np.testing.assert_array_equal(bval, np.array([1.0, 4.0, 4.0]) / 3.0)


def test_haloc_hash_by_hand(oracle):
    """hash[i*32 + c] = sum_m r_i[m] * desc[m][c] / rows (src/hash.cpp:70-82), fp32 accumulation in row order."""
    desc = np.zeros((3, 32), np.uint8)
    desc[:, 0], desc[:, 5] = [10, 20, 30], [1, 0, 255]
    proj = np.array([[1.0, 0.5, -1.0, 99.0], [0.0, 2.0, 0.0, 99.0]], np.float32)       # 4th entry unused: 3 rows only
    h = oracle.haloc_hash(proj, desc)
    assert h.shape == (64,)
    assert h[0] == np.float32(np.float32(10 + 10 - 30) / np.float32(3)) and h[5] == np.float32(np.float32(1 - 255) / np.float32(3))
    assert h[32] == np.float32(np.float32(40) / np.float32(3)) and h[32 + 5] == 0 and h[1] == 0
    assert (oracle.haloc_hash(proj, desc[:0]) == 0).all()


# ---- the four ORBmatcher members without a caller in the reference (src/ORBmatcher.cc:409-713, :1507-1620) ----------
def _desc(bits):
    """32-byte descriptor with the first `bits` bits set."""
    d = np.zeros(256, np.uint8)
    d[:bits] = 1
    return np.packbits(d)


def test_window_search_rule_exclusivity_and_level_filter(oracle):
    """:409-516 -- accept iff best <= nnratio * second (second = INT_MAX when alone) and best <= TH_HIGH; a target taken by an earlier
    F1 keypoint is skipped; F1 keypoints without a map point or outside [minScaleLevel, maxScaleLevel] are not searched."""
    bounds = (0, 0, 640, 480)
    kp2 = _kps([(100, 100), (104, 100), (300, 300)], [0, 0, 1])
    kp2["angle"] = [10, 10, 10]
    d2 = np.stack([_desc(0), _desc(30), _desc(0)])
    kp1 = _kps([(101, 100), (102, 100), (300, 301), (300, 299), (500, 50)], [0, 0, 1, 1, 0])
    kp1["angle"] = [10, 10, 10, 10, 10]
    d1 = np.stack([_desc(10), _desc(2), _desc(120), _desc(5), _desc(0)])
    has = np.array([1, 1, 1, 1, 1], np.uint8)
    # query 0: d(t0) = 10, d(t1) = 20 -> 10 <= 0.9 * 20 -> takes t0; query 1: t0 is taken, t1 alone at d = 28 -> accepted (second = INT_MAX);
    # query 2 (level 1): t2 at 120 > TH_HIGH -> rejected; query 3: t2 at 5 -> accepted; query 4: empty window
    m21, n = oracle.window_search(kp1, d1, has, kp2, d2, bounds, 10, -1, 0x7fffffff, 0.9, False)
    assert m21.tolist() == [0, 1, 3] and n == 3
    # ratio: with nnratio 0.4 query 0 fails (10 > 0.4 * 20) and leaves t0 free for query 1 (d = 2, second 28)
    m21, n = oracle.window_search(kp1, d1, has, kp2, d2, bounds, 10, -1, 0x7fffffff, 0.4, False)
    assert m21.tolist() == [1, -1, 3] and n == 2
    # keypoints without a map point are skipped; maxScaleLevel = 0 drops the level-1 queries, minScaleLevel = 1 the level-0 ones
    m21, n = oracle.window_search(kp1, d1, np.array([0, 1, 1, 1, 1], np.uint8), kp2, d2, bounds, 10, -1, 0x7fffffff, 0.9, False)
    assert m21.tolist() == [1, -1, 3]
    assert oracle.window_search(kp1, d1, has, kp2, d2, bounds, 10, -1, 0, 0.9, False)[0].tolist() == [0, 1, -1]
    assert oracle.window_search(kp1, d1, has, kp2, d2, bounds, 10, 1, 0x7fffffff, 0.9, False)[0].tolist() == [-1, -1, 3]
    # rotation check: three matches in bin 0 and one at 90 degrees (bin 3): 1 >= 0.1 * 3, so nothing is removed ...
    kp1b = kp1.copy()
    kp1b["angle"][3] = 100
    assert oracle.window_search(kp1b, d1, has, kp2, d2, bounds, 10, -1, 0x7fffffff, 0.9, True)[1] == 3


def test_search_for_initialization_takes_over_a_target_only_with_a_smaller_distance(oracle):
    """:598-713 -- vMatchedDistance: a later F1 keypoint replaces the holder of a target iff its distance is strictly smaller; the loser
    drops to -1; candidates matched at <= the query's distance are invisible to it (they do not even count as second best); only
    level-0 keypoints are searched; matched keypoints get the target's position as their new vbPrevMatched."""
    bounds = (0, 0, 640, 480)
    kp2 = _kps([(100, 100), (106, 100)], [0, 0])
    d2 = np.stack([_desc(0), _desc(200)])
    kp1 = _kps([(100, 101), (101, 100), (100, 99), (99, 100), (100, 100)], [0, 0, 0, 0, 1])
    d1 = np.stack([_desc(20), _desc(10), _desc(10), _desc(15), _desc(0)])
    prev = np.array([[100, 100]] * 5, np.float32)
    m12, n = oracle.search_for_initialization(kp1, d1, kp2, d2, bounds, prev, 10, 0.9, False)
    # q0 takes t0 at 20; q1 (10 < 20) takes it over; q2 (10, not < 10) and q3 (15) do not see t0 at all and t1 is too far (190/185 > TH_LOW);
    # q4 is on level 1
    assert m12.tolist() == [-1, 0, -1, -1, -1] and n == 1
    assert prev.tolist() == [[100, 100]] * 5                     # q1's new prev = t0's position = what it was
    prev = np.array([[100, 100]] * 5, np.float32)
    prev[1] = [103, 100]
    m12, n = oracle.search_for_initialization(kp1, d1, kp2, d2, bounds, prev, 10, 0.9, False)
    assert m12[1] == 0 and prev[1].tolist() == [100, 100]       # :705-708
    # the rotation histogram counts every accept, the displaced one included: q0 (bin 0) and q1 (bin 3) -> both bins survive
    kp1r = kp1.copy()
    kp1r["angle"][1] = 90
    prev = np.array([[100, 100]] * 5, np.float32)
    m12, n = oracle.search_for_initialization(kp1r, d1, kp2, d2, bounds, prev, 10, 0.9, True)
    assert m12.tolist() == [-1, 0, -1, -1, -1] and n == 1


def test_projection_searches_between_two_frames(oracle):
    """SearchByProjection(F1, F2, windowSize) :519-594 and SearchByProjection(CurrentFrame, LastFrame, th) :1507-1620 on an identity pose:
    a point (X, Y, Z) lands at (fx X / Z + cx, fy Y / Z + cy); pre-assigned keypoints are skipped; the second form tests the image
    bounds and searches levels [octave - 1, octave + 1] within th * scale[octave]."""
    fx = fy = 100.0
    cx, cy = 320.0, 240.0
    cam = np.concatenate([np.eye(3).reshape(9), np.zeros(3), np.zeros(3), [fx, fy, cx, cy], [0, 640, 0, 480]]).astype(np.float32)
    kp2 = _kps([(320, 240), (420, 240), (326, 240)], [0, 1, 0])
    d2 = np.stack([_desc(0), _desc(0), _desc(40)])
    kp1 = _kps([(0, 0), (0, 0), (0, 0)], [0, 1, 0])
    d1 = np.stack([_desc(3), _desc(3), _desc(3)])
    xyz = np.array([[0, 0, 2], [2, 0, 2], [0, 0, -2]], np.float32)   # -> (320, 240), (420, 240), (320, 240): no depth test in this member
    a2 = np.array([-1, -1, -1], np.int32)
    n = oracle.search_by_projection_frames(kp1, d1, np.ones(3, np.uint8), xyz, cam, kp2, d2, a2, 10, 0.9)
    # point 0 -> t0 (3 <= 0.9 * 37); point 1 (level 1) -> t1; point 2 projects to the same pixel, t0 is taken, t2 alone at 37 -> accepted
    assert n == 3 and a2.tolist() == [0, 1, 2]
    a2 = np.array([7, -1, -1], np.int32)                              # t0 already holds a map point
    n = oracle.search_by_projection_frames(kp1, d1, np.array([1, 0, 0], np.uint8), xyz, cam, kp2, d2, a2, 10, 0.9)
    assert n == 1 and a2.tolist() == [7, -1, 0]
    # last-frame form: the point behind the camera still projects (no depth test) but a point outside the image is dropped
    sf = np.float32([1.0, 1.2, 1.44])
    a = np.array([-1, -1, -1], np.int32)
    xyz2 = np.array([[0, 0, 2], [2, 0, 2], [8, 0, 2]], np.float32)   # the third lands at u = 720 > maxX
    n = oracle.search_by_projection_last(cam, kp2, d2, a, np.ones(3, np.uint8), xyz2, np.array([0, 1, 0], np.int32), np.zeros(3, np.float32), d1, sf, 7.0,
                                         False)
    assert n == 2 and a.tolist() == [0, 1, -1]                        # t2 (6 px away, d = 37) loses to t0 (d = 3) for point 0
    a = np.array([-1, -1, -1], np.int32)
    n = oracle.search_by_projection_last(cam, kp2, d2, a, np.ones(3, np.uint8), xyz2, np.array([0, 1, 0], np.int32), np.zeros(3, np.float32), d1, sf, 3.0,
                                         False)
    assert n == 2 and a.tolist() == [0, 1, -1]
