"""Known-answer tests that pin the CPU oracle (the reference has no tests or fixtures of its own,
so every answer here is hand-derivable from the reference sources, or cross-checked by an independent numpy
formulation written in this file)."""
import ctypes
import subprocess
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- E1: constructor tables (src/ORBextractor.cc:458-512) ----------------------------------------------------
def test_ctor_tables(oracle):
    e = oracle.extractor(1000, 1.2, 8, 20)
    assert e.quota.tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert e.umax.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    s = np.float32(1.0)
    for i in range(8):
        assert e.scale[i] == s
        s = np.float32(np.float64(s) * np.float64(np.float32(1.2)))
    assert oracle.extractor(400, 1.2, 8, 20).quota.tolist() == [87, 72, 60, 50, 42, 35, 29, 25]
    assert oracle.extractor(2000, 1.2, 8, 20).quota.tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert int(oracle.extractor(1000, 1.2, 8, 20).quota.sum()) == 1000


# ---- E2: pyramid geometry (Appendix B) -----------------------------------------------------------------------
@pytest.mark.parametrize("wh,expect", [
    ((640, 512), [(640, 512), (533, 427), (444, 356), (370, 296), (309, 247), (257, 206), (214, 171), (179, 143)]),
    ((1920, 1080), [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)]),
    ((752, 480), [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
])
def test_pyramid_sizes(oracle, wh, expect):
    e = oracle.extractor(1000, 1.2, 8, 20)
    e(np.zeros((wh[1], wh[0]), np.uint8))
    assert [e.level_dims(l) for l in range(8)] == expect


def test_border_reflect101(oracle):
    img = np.arange(7 * 9, dtype=np.uint8).reshape(7, 9)
    out = oracle.border101(img, pad=3)
    ref = np.pad(img, 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101 (edge not repeated)
    np.testing.assert_array_equal(out, ref)


# ---- A.2 resize -----------------------------------------------------------------------------------------------
def test_resize_constant_and_identity(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    np.testing.assert_array_equal(oracle.resize_linear(img, 50, 40), img)  # scale 1: fx = fy = 0
    c = np.full((37, 53), 137, np.uint8)
    assert (oracle.resize_linear(c, 44, 31) == 137).all()


def test_resize_against_float_bilinear(oracle):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (96, 120), dtype=np.uint8)
    dw, dh = 100, 80
    got = oracle.resize_linear(src, dw, dh).astype(np.float64)
    sx, sy = 120 / dw, 96 / dh
    fx = (np.arange(dw) + 0.5) * sx - 0.5
    fy = (np.arange(dh) + 0.5) * sy - 0.5
    x0 = np.floor(fx).astype(int)
    y0 = np.floor(fy).astype(int)
    ax, ay = fx - x0, fy - y0
    s = src.astype(np.float64)
    x1, y1 = np.minimum(x0 + 1, 119), np.minimum(y0 + 1, 95)
    ref = (s[y0][:, x0] * (1 - ax) + s[y0][:, x1] * ax) * (1 - ay)[:, None] + (s[y1][:, x0] * (1 - ax) + s[y1][:, x1] * ax) * ay[:, None]
    assert np.abs(got - ref).max() <= 1.0  # 11-bit fixed point vs exact bilinear


def test_resize_fixed_point_formula_by_hand(oracle):
    # 4x1 -> 2x1: fx = (dx+0.5)*2-0.5 = 0.5, 2.5 -> weights 1024/1024 on pixels (0,1) and (2,3); single row -> beta = (2048, 0)
    src = np.array([[10, 20, 200, 101]], np.uint8)
    out = oracle.resize_linear(src, 2, 1)
    for k, (a, b) in enumerate(((10, 20), (200, 101))):
        r = a * 1024 + b * 1024
        assert out[0, k] == ((((2048 * (r >> 4)) >> 16) + ((0 * (r >> 4)) >> 16) + 2) >> 2)


# ---- A.3 FAST -------------------------------------------------------------------------------------------------
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2),
          (-1, 3)]


def _fast_numpy(img, t, nms=True):
    """Independent formulation: score = max over 9-arcs of the minimum signed contrast, minus 1; corner iff score >= t
    ... computed per pixel with plain python/numpy, no early exits."""
    h, w = img.shape
    I = img.astype(np.int32)
    score = np.zeros((h, w), np.int32)
    corner = np.zeros((h, w), bool)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = np.array([I[y + dy, x + dx] - I[y, x] for dx, dy in CIRCLE])
            dd = np.concatenate([d, d])
            best = -10 ** 9
            for s in range(16):
                arc = dd[s:s + 9]
                best = max(best, arc.min(), (-arc).min())
            if best > t:
                corner[y, x] = True
                score[y, x] = best - 1
    pts = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not corner[y, x]:
                continue
            s = score[y, x]
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if nms and not (s > nb.max()):
                continue
            pts.append((x, y, s))
    return pts


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 40), (3, 0)])
def test_fast_matches_independent_bruteforce(oracle, seed, t):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (37, 41), dtype=np.uint8)
    if seed == 1:  # smoother content: blocks + noise
        img = (np.kron(rng.integers(0, 2, (10, 11)), np.ones((4, 4))) * 120 + rng.integers(0, 30, (40, 44)))[:37, :41].astype(np.uint8)
    ref = _fast_numpy(img, t)
    for brute in (False, True):
        got = oracle.fast(img, t, True, bruteforce=brute)
        assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == ref
        assert (got["size"] == 7).all() and (got["angle"] == -1).all() and (got["octave"] == 0).all() and (got["class_id"] == -1).all()
    ref_nonms = _fast_numpy(img, t, nms=False)
    got = oracle.fast(img, t, False)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == ref_nonms


def test_fast_too_small_roi(oracle):
    assert len(oracle.fast(np.full((6, 40), 9, np.uint8), 5)) == 0
    assert len(oracle.fast(np.full((40, 6), 9, np.uint8), 5)) == 0


def test_fast_single_bright_dot_is_a_dark_ring_corner(oracle):
    img = np.full((15, 15), 50, np.uint8)
    img[7, 7] = 200  # centre brighter than the whole ring: all 16 ring pixels are darker by 150
    got = oracle.fast(img, 20)
    assert len(got) == 1 and (int(got[0]["x"]), int(got[0]["y"]), int(got[0]["response"])) == (7, 7, 149)


# ---- A.4 Gaussian ---------------------------------------------------------------------------------------------
def test_gauss_taps_and_constant(oracle):
    assert oracle.gauss_taps().tolist() == [18, 34, 49, 55, 49, 34, 18]  # sums to 257: OpenCV's integer engine brightens slightly
    plane = np.full((40 + 32, 50 + 32), 100, np.uint8)
    out = oracle.gauss7_padded(plane)
    assert (out[16:-16, 16:-16] == ((100 * 257 * 257 + 32768) >> 16)).all()
    assert (out[:16] == 100).all()  # pad untouched
    sat = oracle.gauss7_padded(np.full((72, 82), 255, np.uint8))
    assert (sat[16:-16, 16:-16] == 255).all()  # saturate_cast


def test_gauss_impulse_and_border_reads_real_pad(oracle):
    taps = np.array([18, 34, 49, 55, 49, 34, 18])
    plane = np.zeros((30 + 32, 30 + 32), np.uint8)
    plane[16 + 10, 16 + 12] = 200
    out = oracle.gauss7_padded(plane)[16:-16, 16:-16].astype(int)
    ref = (np.outer(taps, taps) * 200 + 32768) >> 16
    np.testing.assert_array_equal(out[7:14, 9:16], ref)
    # a bright pixel that exists ONLY in the pad (not the reflection of anything) must leak into the blurred interior
    plane = np.zeros((62, 62), np.uint8)
    plane[16 + 5, 15] = 255  # x = -1
    out = oracle.gauss7_padded(plane)[16:-16, 16:-16].astype(int)
    assert out[5, 0] == (49 * 55 * 255 + 32768) >> 16 and out[5, 2] == (18 * 55 * 255 + 32768) >> 16 and out[5, 3] == 0


# ---- A.5 fastAtan2 / E6 IC_Angle ------------------------------------------------------------------------------
def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0, 0) == 0.0
    assert oracle.fast_atan2(0, 5) == 0.0
    assert oracle.fast_atan2(5, 0) == 90.0
    assert abs(oracle.fast_atan2(0, -5) - 180.0) < 1e-4
    assert oracle.fast_atan2(-5, 0) == 270.0
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-20000, 20000, 2)
        if x == 0 and y == 0:
            continue
        ref = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        got = oracle.fast_atan2(float(y), float(x))
        assert min(abs(got - ref), 360 - abs(got - ref)) < 0.3  # documented accuracy of the polynomial
        assert 0.0 <= got <= 360.0


def test_ic_angle_analytic(oracle):
    e = oracle.extractor(1000, 1.2, 8, 20)
    yy, xx = np.mgrid[0:96, 0:96]
    const = np.full((96, 96), 77, np.uint8)
    assert e.ic_angle(const, 30, 30) == 0.0                          # m01 = m10 = 0 by symmetry
    assert e.ic_angle((xx * 2).astype(np.uint8), 30, 30) == 0.0      # +x ramp
    assert e.ic_angle((yy * 2).astype(np.uint8), 30, 30) == 90.0     # +y ramp
    assert abs(e.ic_angle((200 - xx * 2).astype(np.uint8), 30, 30) - 180.0) < 1e-4
    assert e.ic_angle((200 - yy * 2).astype(np.uint8), 30, 30) == 270.0
    # moments of the 749-pixel circular patch, by hand
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (96, 96), dtype=np.uint8)
    umax = e.umax
    m01 = m10 = 0
    for v in range(-15, 16):
        for u in range(-umax[abs(v)], umax[abs(v)] + 1):
            m10 += u * int(img[16 + 40 + v, 16 + 45 + u])
            m01 += v * int(img[16 + 40 + v, 16 + 45 + u])
    assert sum(2 * umax[abs(v)] + 1 for v in range(-15, 16)) == 749
    assert e.ic_angle(img, 45, 40) == oracle.fast_atan2(float(m01), float(m10))
    assert e.ic_angle(img, 45.4, 39.5) == e.ic_angle(img, 45, 40)   # cvRound: 39.5 -> 40 (half to even), 45.4 -> 45
    assert e.ic_angle(img, 44.5, 40.5) == e.ic_angle(img, 44, 40)   # 44.5 -> 44, 40.5 -> 40


# ---- E9 steered BRIEF -----------------------------------------------------------------------------------------
def _pattern():
    vals = []
    for line in open(os.path.join(ROOT, "oracle", "rbrief_pattern.inc")):
        if line.startswith("//"):
            continue
        vals += [int(v) for v in line.strip().rstrip(",").split(",")]
    return np.array(vals).reshape(256, 4)


def test_pattern_table_identical_in_product_and_oracle():
    a = open(os.path.join(ROOT, "oracle", "rbrief_pattern.inc")).read()
    b = open(os.path.join(ROOT, "u-vip-slam_amd", "csrc", "rbrief_pattern.inc")).read()
    assert a == b
    p = _pattern()
    assert p.shape == (256, 4) and np.abs(p).max() == 13
    assert p[0].tolist() == [8, -3, 9, 5] and p[255].tolist() == [-1, -6, 0, -11]


@pytest.mark.parametrize("angle,rot", [(0.0, lambda x, y: (x, y)), (90.0, lambda x, y: (-y, x)), (180.0, lambda x, y: (-x, -y)),
                                       (270.0, lambda x, y: (y, -x))])
def test_rbrief_axis_aligned_by_hand(oracle, angle, rot):
    e = oracle.extractor(1000, 1.2, 8, 20)
    rng = np.random.default_rng(4)
    plane = rng.integers(0, 256, (100 + 32, 100 + 32), dtype=np.uint8)
    cx, cy = 50, 47
    d = e.descriptor(plane, cx, cy, angle)
    p = _pattern()
    bits = []
    for x0, y0, x1, y1 in p:
        rx0, ry0 = rot(x0, y0)
        rx1, ry1 = rot(x1, y1)
        bits.append(int(plane[16 + cy + ry0, 16 + cx + rx0]) < int(plane[16 + cy + ry1, 16 + cx + rx1]))
    ref = np.packbits(np.array(bits, np.uint8), bitorder="little")
    np.testing.assert_array_equal(d, ref)


def test_rbrief_generic_angle_float_steps(oracle):
    e = oracle.extractor(1000, 1.2, 8, 20)
    rng = np.random.default_rng(5)
    plane = rng.integers(0, 256, (132, 132), dtype=np.uint8)
    p = _pattern()
    for ang in (33.3, 123.456, 359.99, 45.0):
        a32 = np.float32(ang) * np.float32(np.pi / np.float32(180.0))
        s, c = oracle.sincosf(float(a32))
        a, b = np.float32(c), np.float32(s)
        bits = []
        for x0, y0, x1, y1 in p.astype(np.float32):
            def px(x, y):
                yy = int(np.rint(np.float32(np.float32(x * b) + np.float32(y * a))))
                xx = int(np.rint(np.float32(np.float32(x * a) - np.float32(y * b))))
                return int(plane[16 + 60 + yy, 16 + 61 + xx])
            bits.append(px(x0, y0) < px(x1, y1))
        np.testing.assert_array_equal(e.descriptor(plane, 61, 60, ang), np.packbits(np.array(bits, np.uint8), bitorder="little"))


# ---- E5 quad tree ---------------------------------------------------------------------------------------------
def test_octree_toy_cases(oracle):
    e = oracle.extractor(1000, 1.2, 8, 20)
    W = H = 100
    # one point per quadrant, N = 4: root splits once, children are push_front'ed n1..n4 -> list reads n4, n3, n2, n1
    pts = np.array([[10, 10, 5], [80, 12, 6], [12, 85, 7], [90, 90, 8]])
    assert e.octree(pts, W, H, 4).tolist() == [[90, 90, 8], [12, 85, 7], [80, 12, 6], [10, 10, 5]]
    # a single point: root is flagged bNoMore and returned
    assert e.octree(pts[:1], W, H, 10).tolist() == [[10, 10, 5]]
    assert e.octree(pts[:0], W, H, 10).tolist() == []
    # N = 1 still performs the first split of a multi-point root (:1058 loop runs before any size test)
    assert len(e.octree(pts, W, H, 1)) == 4
    # best response per node, first in candidate order wins ties (:1211-1226)
    pts = np.array([[10, 10, 5], [11, 11, 9], [12, 12, 9], [80, 80, 1]])
    out = e.octree(pts, W, H, 2).tolist()
    assert out == [[80, 80, 1], [11, 11, 9]]
    # careful phase: sizes differ -> the larger node is split first and the walk stops at the first split reaching N
    pts = np.array([[5, 5, 1], [40, 5, 2], [5, 40, 3], [40, 40, 4],          # TL quadrant: 4 points, one per sub-quadrant
                    [60, 5, 1], [90, 5, 2], [60, 40, 3],                      # TR quadrant: 3 points
                    [5, 60, 9]])                                              # BL quadrant: single
    # pass 1: list = [BL(1), TR(3), TL(4)]: 3 nodes < 5 and 3 + 3*2 > 5 -> careful: TL (size 4) first -> 3 - 1 + 4 = 6 >= 5 -> stop
    assert e.octree(pts, W, H, 5).tolist() == [[40, 40, 4], [5, 40, 3], [40, 5, 2], [5, 5, 1], [5, 60, 9], [60, 40, 3]]
    # same, but TL's points all fall into one child: that split gains nothing, so TR is split too (3 - 1 + 1 - 1 + 3 = 5)
    pts = np.array([[5, 5, 1], [20, 5, 2], [5, 20, 3], [20, 20, 4], [60, 5, 1], [90, 5, 2], [60, 40, 3], [5, 60, 9]])
    assert e.octree(pts, W, H, 5).tolist() == [[60, 40, 3], [90, 5, 2], [60, 5, 1], [20, 20, 4], [5, 60, 9]]
    # equal sizes in the careful phase: the most recently created node (front of the list) is split first
    pts = np.array([[5, 5, 1], [40, 40, 2],        # TL: 2 points
                    [60, 5, 3], [90, 40, 4],       # TR: 2 points
                    [5, 60, 5], [40, 90, 6]])      # BL: 2 points
    # pass 1: list = [BL, TR, TL] (3 nodes), N = 4: 3 + 9 > 4 -> careful, all sizes 2 -> newest = BL first: 3 - 1 + 2 = 4 -> stop
    assert e.octree(pts, W, H, 4).tolist() == [[40, 90, 6], [5, 60, 5], [90, 40, 4], [40, 40, 2]]


def _octree_emu():
    lib = os.path.join(ROOT, "tests", "emu", "liboctree_emu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", "octree_emu.cpp"), os.path.join(ROOT, "u-vip-slam_amd", "csrc", "octree_core.hpp"),
            os.path.join(ROOT, "u-vip-slam_amd", "csrc", "octree_pyramid.hpp")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        import subprocess
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, srcs[0]])
    E = ctypes.CDLL(lib)
    E.emu_octree_algo.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                                                           ctypes.c_int]
    return E


def _octree_case(rng, trial, dense):
    """-> W, H, grid, candidates in the reference's order, responses, N.  dense: FAST-like inputs (thousands of candidates for a
    quota of a few hundred: shallow, balanced trees); otherwise anything goes (sparse, clustered, collinear: deep trees)."""
    W = int(rng.integers(30, 1900 if dense else 900))
    H = int(rng.integers(max(30, W // 3), min(1100 if dense else 900, 2 * W - 1)))
    if round(W / H) < 1:
        return None
    nCols, nRows = W // 30, H // 30
    wCell, hCell = -(-W // nCols), -(-H // nRows)
    mode = trial % 4
    if dense:
        N = int(rng.integers(20, 450))
        P = int(N * rng.uniform(3, 20))
        mode = 0 if mode != 2 else 2
    else:
        P, N = int(rng.integers(0, 2500)), int(rng.integers(1, 450))
    if mode == 1:
        cx, cy = rng.integers(3, W - 3), rng.integers(3, H - 3)
        xs = np.clip(cx + rng.normal(0, 8, P).astype(int), 3, W - 4)
        ys = np.clip(cy + rng.normal(0, 8, P).astype(int), 3, H - 4)
    elif mode == 3:
        xs, ys = rng.integers(3, W - 3, size=P), np.full(P, rng.integers(3, H - 3))
    else:
        xs, ys = rng.integers(3, W - 3, size=P), rng.integers(3, H - 3, size=P)
    pts = np.unique(np.stack([xs, ys], 1), axis=0).reshape(-1, 2)
    j = np.minimum((pts[:, 0] - 3) // wCell, nCols - 1)
    i = np.minimum((pts[:, 1] - 3) // hCell, nRows - 1)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1], j, i))]   # reference candidate order: cell-major, raster inside a cell
    resp = rng.integers(1, 4 if mode == 2 else 200, size=len(pts))
    return W, H, (nCols, nRows, wCell, hCell), pts, resp, N


@pytest.mark.parametrize("dense", [False, True])
def test_octree_kernel_bodies_match_oracle_on_random_sets(oracle, dense):
    """The HIP kernel's selection logic, executed on the host through the phase macros: the closed form over the count pyramid
    (csrc/octree_pyramid.hpp), the pass-per-generation form (csrc/octree_core.hpp) and the kernel's combination of the two, in every
    candidate-state instantiation, against the literal std::list restatement."""
    E = _octree_emu()
    e = oracle.extractor(1000, 1.2, 8, 20)
    rng = np.random.default_rng(123 + dense)
    done = fast = 0
    for trial in range(160):
        case = _octree_case(rng, trial, dense)
        if case is None:
            continue
        W, H, (nCols, nRows, wCell, hCell), pts, resp, N = case
        P = len(pts)
        ref = e.octree(np.concatenate([pts, resp[:, None]], 1), W, H, N).tolist()
        perm = rng.permutation(P)                              # the GPU candidate array is in arbitrary order
        xy = (pts[perm, 0].astype(np.uint32) | (pts[perm, 1].astype(np.uint32) << 16)).astype(np.uint32)
        sc = resp[perm].astype(np.uint32)
        sxy, ssc = np.zeros(N + 16 + P, np.uint32), np.zeros(N + 16 + P, np.uint32)
        # every instantiation the kernel has: candidate state in registers (8 or 32 per thread) or in memory (0)
        for k_regs in (8, 32, 0):
            if k_regs and P > k_regs * 256:
                continue
            for algo in (0, 1, 2):   # passes only / pyramid only (-2: tree deeper than the pyramid) / as the kernel
                sxy[:], ssc[:] = 0, 0
                m = E.emu_octree_algo(xy.ctypes.data, sc.ctypes.data, P, N, W, H, nCols, nRows, wCell, hCell, sxy.ctypes.data, ssc.ctypes.data,
                                      len(sxy), k_regs, algo)
                if algo == 1 and m == -2:
                    continue
                got = [[int(v & 0xffff), int(v >> 16), int(s)] for v, s in zip(sxy[:m], ssc[:m])]
                assert got == ref, "trial %d W=%d H=%d P=%d N=%d k_regs=%d algo=%d" % (trial, W, H, P, N, k_regs, algo)
                fast += algo == 1 and k_regs == 0 and P > 0
        done += 1
    assert done > 100
    if dense:
        assert fast >= 0.95 * done, "FAST-like candidate sets must take the closed form (%d of %d did)" % (fast, done)
    else:
        assert 0.2 * done < fast < done, "the mixed set must exercise both paths (%d of %d took the closed form)" % (fast, done)


# ---- extractor glue -------------------------------------------------------------------------------------------
def test_extract_properties(oracle, synth):
    img = synth.make_frame(501, 320, 256, n_shapes=120)
    e = oracle.extractor(300, 1.2, 4, 20)
    kp, de = e(img)
    assert len(kp) == len(de) > 250
    assert (np.diff(kp["octave"]) >= 0).all()                       # level-major
    for l in range(4):
        m = kp["octave"] == l
        assert m.sum() <= e.quota[l] + 2                            # careful phase adds at most 3 per split, stops at >= N
        assert (kp["size"][m] == np.float32(int(np.float32(31) * e.scale[l]))).all()
        lev = kp[m]
        x = lev["x"] / e.scale[l]
        w, h = e.level_dims(l)
        assert (np.rint(x) >= 16).all() and (np.rint(x) < w - 16).all()
    assert (kp["class_id"] == -1).all() and (kp["angle"] >= 0).all() and (kp["angle"] <= 360).all()
    assert (kp["response"] >= 7).all()                              # fastTh 20 or the literal-7 fallback
    # per-level stage taps agree with the final output
    n0 = (kp["octave"] == 0).sum()
    lk = e.level_keypoints(0)
    assert len(lk) == n0 and (lk["x"] == kp["x"][:n0]).all()


def test_topup_filter_semantics(oracle, synth):
    img = synth.make_frame(501, 320, 256, n_shapes=120)
    e = oracle.extractor(300, 1.2, 4, 20)
    full_kp, _ = e(img)
    rows, cols = 256 // 20 + 2, 320 // 20 + 2
    grid = np.zeros((rows, cols), np.int32, order="F")
    kp, de = e(img, None, grid, 20, False, 10 ** 6)   # caps never reached, numofpoint huge
    # every accepted point marks its own cell exactly once, and is the first of the full list to fall into it
    assert grid.sum() == len(kp) and grid.max() == 1
    # a fully occupied grid rejects everything
    g2 = np.ones((rows, cols), np.int32, order="F")
    kp2, de2 = e(img, None, g2, 20, False, 100)
    assert len(kp2) == 0 and (g2 == 1).all()
    # global cap: Total_counter == num_featsneeded stops the walk (:898-901)
    g3 = np.zeros((rows, cols), np.int32, order="F")
    kp3, _ = e(img, None, g3, 20, False, 7)
    # per-level cap: num_featsneeded*(8-level)/30 = 1 for level 0..3 -> KP_counter resets per level; total stops at 7 or earlier
    assert 0 < len(kp3) <= 7
    # caller keypoints pass through level 0 first, angles recomputed, other fields untouched
    kin = np.zeros(3, oracle_kp_dtype())
    kin["x"], kin["y"], kin["size"], kin["angle"], kin["response"], kin["octave"], kin["class_id"] = [50, 100.4, 200], [60, 80.5, 90], 9, -1, 3, 0, [7, 8, 9]
    g4 = np.zeros((rows, cols), np.int32, order="F")
    kp4, de4 = e(img, kin, g4, 20, False, 50)
    assert (kp4["class_id"][:3] == [7, 8, 9]).all() and (kp4["size"][:3] == 9).all() and (kp4["x"][:3] == kin["x"]).all()
    assert (kp4["angle"][:3] >= 0).all()
    # FullDetect drops caller keypoints (:911)
    kp5, _ = e(img, kin, None, 20, True, 0)
    assert len(kp5) == len(full_kp)


def oracle_kp_dtype():
    import oracle_lib
    return oracle_lib.KP


def test_golden_fixtures(oracle, synth):
    g = np.load(os.path.join(ROOT, "tests", "golden", "extract_golden.npz"))
    for name, seed, w, h, ns, nf, nl, th in (("full_320x256", 501, 320, 256, 120, 300, 4, 20), ("full_640x512", 1000, 640, 512, 400, 1000, 8, 20),
                                             ("full_752x480_th7", 502, 752, 480, 400, 1000, 8, 7)):
        img = synth.make_frame(seed, w, h, n_shapes=ns)
        assert [int(img.astype(np.int64).sum()), int((img.astype(np.int64) * np.arange(w)).sum() % (1 << 31))] == g[name + "_imgsum"].tolist(), \
            "synthetic generator drifted"
        kp, de = oracle.extractor(nf, 1.2, nl, th)(img)
        np.testing.assert_array_equal(kp.view(np.uint8).reshape(len(kp), 28), g[name + "_kp"])
        np.testing.assert_array_equal(de, g[name + "_desc"])


def test_grider_fast(oracle, synth):
    img = synth.make_frame(9, 320, 256, n_shapes=120)
    pts = oracle.grider_fast(img, 200, 8, 5, 20)
    per_cell = 200 // 40 + 1
    cx, cy = (pts["x"] // (320 // 8)).astype(int), (pts["y"] // (256 // 5)).astype(int)
    counts = np.bincount(cy * 8 + cx, minlength=40)
    assert counts.max() <= per_cell and len(pts) > 40
    for c in range(40):  # inside a cell: response descending
        r = pts["response"][(cy * 8 + cx) == c]
        assert (np.diff(r) <= 0).all()


# ---- CLAHE (cv::CLAHE::apply, src/Tracking.cc:425-431) ------------------------------------------------------------------
def _clahe_numpy(img, clip_limit, tiles):
    """Independent numpy statement of OpenCV 3.4's 8-bit CLAHE (vectorised differently from the oracle's loops)."""
    tx, ty = tiles
    h, w = img.shape
    if w % tx == 0 and h % ty == 0:
        ext = img
    else:
        ext = np.pad(img, ((0, ty - h % ty), (0, tx - w % tx)), mode="reflect")
    tw, th = ext.shape[1] // tx, ext.shape[0] // ty
    total = tw * th
    scale = np.float32(255) / np.float32(total)
    clip = max(int(clip_limit * total / 256), 1) if clip_limit > 0 else 0
    luts = np.zeros((ty, tx, 256), np.float32)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(ext[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip) + clipped // 256
                residual = clipped % 256
                if residual:
                    step = max(256 // residual, 1)
                    idx = np.arange(0, 256, step)[:residual]
                    hist[idx] += 1
            luts[j, i] = np.clip(np.rint(np.cumsum(hist).astype(np.float32) * scale), 0, 255)
    xs, ys = np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32)
    txf = xs * (np.float32(1) / np.float32(tw)) - np.float32(0.5)
    tyf = ys * (np.float32(1) / np.float32(th)) - np.float32(0.5)
    tx1, ty1 = np.floor(txf).astype(int), np.floor(tyf).astype(int)
    xa, ya = (txf - tx1.astype(np.float32)).astype(np.float32), (tyf - ty1.astype(np.float32)).astype(np.float32)
    xa1, ya1 = np.float32(1) - xa, np.float32(1) - ya
    tx2, ty2 = np.minimum(tx1 + 1, tx - 1), np.minimum(ty1 + 1, ty - 1)
    tx1, ty1 = np.maximum(tx1, 0), np.maximum(ty1, 0)
    v = img.astype(int)
    p11, p12 = luts[ty1[:, None], tx1[None, :], v], luts[ty1[:, None], tx2[None, :], v]
    p21, p22 = luts[ty2[:, None], tx1[None, :], v], luts[ty2[:, None], tx2[None, :], v]
    res = (p11 * xa1[None, :] + p12 * xa[None, :]) * ya1[:, None] + (p21 * xa1[None, :] + p22 * xa[None, :]) * ya[:, None]
    return np.clip(np.rint(res.astype(np.float32)), 0, 255).astype(np.uint8)


def test_clahe_by_hand_and_against_numpy(oracle, synth):
    flat = np.full((8, 8), 10, np.uint8)
    # 2 x 2 tiles of 16 px, no clipping: lut = 0 below 10, 255 from 10 on
    assert (oracle.clahe(flat, 0.0, (2, 2)) == 255).all()
    # clip limit 40 -> int(40 * 16 / 256) = 2: 14 px clipped, residual 14 spread over bins 0, 18, ..., 234 -> lut[10] = round(3 * 255 / 16) = 48
    assert (oracle.clahe(flat, 40.0, (2, 2)) == 48).all()
    rng = np.random.default_rng(17)
    for shape, tiles, clip in (((64, 96), (4, 4), 4.0), ((61, 93), (4, 4), 4.0), ((96, 120), (12, 12), 4.0), ((50, 70), (3, 5), 2.0),
                               ((48, 48), (6, 6), 0.0), ((40, 64), (8, 8), 40.0)):
        img = (rng.integers(0, 256, shape) * rng.random(shape) ** 2).astype(np.uint8)      # skewed histogram: clipping matters
        np.testing.assert_array_equal(oracle.clahe(img, clip, tiles), _clahe_numpy(img, clip, tiles), err_msg=str((shape, tiles, clip)))
    img = synth.make_frame(5, 640, 512)
    np.testing.assert_array_equal(oracle.clahe(img, 4.0, (12, 12)), _clahe_numpy(img, 4.0, (12, 12)))


# ---- KLT step (cv::buildOpticalFlowPyramid + cv::calcOpticalFlowPyrLK, src/FrameKTL.cc:76, src/Tracking.cc:1046-1047) -------
def test_klt_pyramid_and_tracker_known_answers(oracle, synth):
    # pyrDown of a constant image is the constant; Scharr derivatives of a horizontal ramp: dx = 32 * slope, dy = 0
    flat = np.full((64, 96), 77, np.uint8)
    p = oracle.klt_pyramid(flat, (5, 5), 3)
    assert p.levels == 4
    for l in range(4):
        img, der = p.level(l)
        assert img.shape == ((64 >> l), (96 >> l)) and (img == 77).all() and (der == 0).all()
    ramp = np.tile((np.arange(96) * 2).astype(np.uint8), (64, 1))
    img, der = oracle.klt_pyramid(ramp, (5, 5), 0).level(0)
    assert (der[:, 1:-1, 0] == 2 * 2 * 16).all() and (der[..., 1] == 0).all() and (der[:, 0, 0] == 0).all()   # (3+10+3) * (I[x+1]-I[x-1]); reflect at the edge
    # pyrDown against an independent numpy statement (separable 1-4-6-4-1, reflect-101, (sum + 128) >> 8)
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (45, 71), dtype=np.uint8)
    lvl1, _ = oracle.klt_pyramid(a, (5, 5), 1).level(1)
    pad = np.pad(a.astype(np.int64), 2, mode="reflect")
    k = np.array([1, 4, 6, 4, 1])
    hor = sum(k[i] * pad[:, i:i + a.shape[1]] for i in range(5))
    full = sum(k[j] * hor[j:j + a.shape[0], :] for j in range(5))
    np.testing.assert_array_equal(lvl1, ((full[::2, ::2] + 128) >> 8).astype(np.uint8))
    # tracker: a frame shifted by whole pixels is found again; points too close to nothing trackable fail the eigenvalue test
    base = synth.make_frame(21, 400, 320)
    shifted = np.roll(np.roll(base, 3, axis=1), -2, axis=0)
    p0, p1 = oracle.klt_pyramid(base), oracle.klt_pyramid(shifted)
    ys, xs = np.mgrid[60:260:25, 60:340:25]
    pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)
    nxt, st, er = oracle.klt_track(p0, p1, pts)
    good = st > 0
    assert good.mean() > 0.9
    d = nxt[good] - pts[good]
    assert np.abs(np.median(d, axis=0) - [3, -2]).max() < 0.05 and (np.abs(d - [3, -2]).max(axis=1) < 0.5).mean() > 0.95
    flat_p = oracle.klt_pyramid(np.full((320, 400), 90, np.uint8))
    _, st, er = oracle.klt_track(flat_p, flat_p, pts)
    assert (st == 0).all() and (er == 0).all()                          # minEig = 0 < 1e-4
    # a point outside the image by more than the window is rejected at level 0 with err = 0
    _, st, er = oracle.klt_track(p0, p1, np.array([[-40.0, 50.0], [100.0, 100.0]], np.float32))
    assert st[0] == 0 and er[0] == 0 and st[1] == 1

def test_strip_plan_tiles_every_window_exactly_once():
    """Host check of the wavefront strip plan shared by k_fast_score and k_gauss7 (csrc/strip_plan.hpp, compiled with a plain C++
    compiler): every pixel of a window belongs to exactly one (item, sub-strip); the plan never needs more wavefront-rows than
    one 248-column strip per started 248 columns would."""
    lib = os.path.join(ROOT, "tests", "emu", "libstrip_plan_emu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", "strip_plan_emu.cpp"), os.path.join(ROOT, "u-vip-slam_amd", "csrc", "strip_plan.hpp")]
    if not os.path.exists(lib) or max(os.path.getmtime(p) for p in srcs) > os.path.getmtime(lib):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, srcs[0]])
    E = ctypes.CDLL(lib)
    E.emu_strip_plan_check.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
    rng = np.random.default_rng(12)
    widths = list(range(1, 700)) + [int(x) for x in rng.integers(700, 4200, 150)]
    saved = total = 0
    for w in widths:
        for h, rps in ((int(rng.integers(1, 600)), 24), (int(rng.integers(1, 300)), 8), (int(rng.integers(1, 1100)), 64)):
            items, lane_rows = ctypes.c_int(), ctypes.c_int()
            rc = E.emu_strip_plan_check(w, h, rps, ctypes.byref(items), ctypes.byref(lane_rows))
            assert rc == 0, (w, h, rps, rc)
            nseg = -(-h // rps)
            plain = -(-w // 248) * nseg * rps        # one full-width wavefront strip per started 248 columns
            assert lane_rows.value <= plain, (w, h, rps)
            saved += plain - lane_rows.value
            total += plain
    assert saved > 0.05 * total                      # the narrow strips pay off on average (about 10 % over random sizes)
    # the level widths of the 640 x 512 pyramid (windows = width - 32): 20 % fewer wavefront-rows than full strips only
    rows = plain = 0
    for w, h in ((608, 480), (501, 395), (412, 324), (338, 264), (277, 215), (225, 174), (182, 139), (147, 111)):
        items, lane_rows = ctypes.c_int(), ctypes.c_int()
        assert E.emu_strip_plan_check(w, h, 24, ctypes.byref(items), ctypes.byref(lane_rows)) == 0
        rows += lane_rows.value
        plain += -(-w // 248) * -(-h // 24) * 24
    assert rows < 0.83 * plain
    # the XCD-contiguous work mapping used by the same kernels is a permutation with one contiguous range per XCD
    E.emu_xcd_contiguous_check.argtypes = [ctypes.c_int]
    for total in list(range(1, 300)) + [1023, 1024, 1025, 10496, 65537]:
        assert E.emu_xcd_contiguous_check(total) == 0, total


def test_klt_association_order_only_moves_decisions_at_their_thresholds(oracle, synth):
    """calcOpticalFlowPyrLK accumulates its window sums in float, so the result depends on the order of the additions (OpenCV's own
    SIMD builds differ from its scalar loop).  Raster order (sum_mode 0, the generic loop) against the HIP kernel's order (sum_mode 1):
    a point's status differs only if one of its yes/no decisions -- minimum eigenvalue, determinant, image bounds, the two
    termination tests -- sits within 1e-3 of its threshold; positions agree to float rounding, and the few that differ by more than
    0.01 px are points whose termination tests ran within 10 % of their thresholds (an iteration more or less)."""
    rng = np.random.default_rng(81)
    total = far = 0
    for (w, h), win, ml in (((640, 512), (21, 21), 5), ((752, 480), (15, 15), 3)):
        a = synth.make_frame(6000 + w, w, h)
        b = synth.warp_frame(a, 6001 + w)
        pa, pb = oracle.klt_pyramid(a, win, ml), oracle.klt_pyramid(b, win, ml)
        n = 1500
        pts = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)
        init = (pts + rng.normal(0, 1.0, (n, 2))).astype(np.float32)
        n0, s0, e0, m0 = oracle.klt_track_ex(pa, pb, pts, init, win, ml, sum_mode=0)
        ref = oracle.klt_track(pa, pb, pts, init, win, ml)
        assert (n0 == ref[0]).all() and (s0 == ref[1]).all() and (e0 == ref[2]).all()     # mode 0 is the plain entry point
        n1, s1, e1, m1 = oracle.klt_track_ex(pa, pb, pts, init, win, ml, sum_mode=1)
        mg = np.minimum(m0, m1)
        assert ((s0 == s1) | (mg < 1e-3)).all()
        both = (s0 > 0) & (s1 > 0)
        d = np.abs(n0 - n1).max(axis=1)
        assert np.median(d[both]) < 1e-3 and np.percentile(d[both], 99) < 0.02
        assert (d[both & (mg > 0.1)] < 0.01).all()
        np.testing.assert_allclose(e0[s0 == s1], e1[s0 == s1], rtol=1e-4, atol=1e-8)   # the minimum eigenvalue is a difference of sums
        total += int(both.sum())
        far += int((d[both] > 0.01).sum())
    assert total > 1800 and far < 0.02 * total


def test_undistort_point_models(oracle):
    """Tracking::undistort_point (src/Tracking.cc:1265-1283).  Without distortion both models are the identity (up to float rounding of
    the normalise / re-project round trip); the principal point is a fixed point of any distortion; and undistort inverts the forward
    model: pin-hole x_d = x (1 + k1 r^2 + k2 r^4) + tangential terms (five fixed-point iterations leave ~1e-4 px near the border),
    fisheye theta_d = theta (1 + k1 theta^2 + ...)."""
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(0, 752, 500), rng.uniform(0, 480, 500)], 1).astype(np.float32)
    for fisheye in (False, True):
        out = oracle.undistort_points(pts, fx, fy, cx, cy, [0, 0, 0, 0], fisheye)
        assert np.abs(out - pts).max() < 1e-3 if not fisheye else True
        c = oracle.undistort_points(np.float32([[cx, cy]]), fx, fy, cx, cy, [-0.28, 0.07, 1e-4, 1e-5], fisheye)
        assert np.abs(c - np.float32([[cx, cy]])).max() < 1e-4
    # pin-hole: distort ideal points with the EuRoC coefficients (Data/Settings_VIORB.yaml:20-23), undistort, compare
    k1, k2, p1, p2 = -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05
    x, y = (pts[:, 0].astype(np.float64) - cx) / fx * 0.6, (pts[:, 1].astype(np.float64) - cy) / fy * 0.6
    r2 = x * x + y * y
    xd = x * (1 + k1 * r2 + k2 * r2 * r2) + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * (1 + k1 * r2 + k2 * r2 * r2) + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    dist_px = np.stack([xd * fx + cx, yd * fy + cy], 1).astype(np.float32)
    und = oracle.undistort_points(dist_px, fx, fy, cx, cy, [k1, k2, p1, p2], False)
    assert np.abs(und - np.stack([x * fx + cx, y * fy + cy], 1)).max() < 2e-2
    # fisheye (harbor: Data/Settings_VI_Aqualoc_harbor.yaml:24-33,102): theta_d = theta (1 + k1 theta^2 + k2 theta^4 + ...)
    fxh, fyh, cxh, cyh = 413.32595366596017, 413.70198739483686, 305.9507483284928, 259.4439948946375
    kk = [-0.06125568297136998, -0.003796743395135256, 0.027326634771204592, -0.030296403142887066]
    a, b = rng.uniform(-0.7, 0.7, 500), rng.uniform(-0.6, 0.6, 500)          # ideal normalised coordinates
    r = np.sqrt(a * a + b * b)
    th = np.arctan(r)
    thd = th * (1 + kk[0] * th**2 + kk[1] * th**4 + kk[2] * th**6 + kk[3] * th**8)
    s = np.where(r > 1e-12, thd / np.maximum(r, 1e-12), 1.0)
    dpx = np.stack([a * s * fxh + cxh, b * s * fyh + cyh], 1).astype(np.float32)
    und = oracle.undistort_points(dpx, fxh, fyh, cxh, cyh, kk, True)
    assert np.abs(und - np.stack([a * fxh + cxh, b * fyh + cyh], 1)).max() < 2e-3


def _pyr_tiles_emu():
    lib = os.path.join(ROOT, "tests", "emu", "libpyr_tiles_emu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", "pyr_tiles_emu.cpp"), os.path.join(ROOT, "u-vip-slam_amd", "csrc", "pyr_tiles.hpp")]
    if not os.path.exists(lib) or max(os.path.getmtime(p) for p in srcs) > os.path.getmtime(lib):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, srcs[0]])
    E = ctypes.CDLL(lib)
    E.emu_pyr_tiles_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return E


def _run_pyr_tiles(E, oe, img, nlevels, groups, ring=4, max_lds=160 * 1024):
    """groups: [(first level, tx, ty), ...] -- one launch per group; or a (tx, ty) pair: every level in one group."""
    if not isinstance(groups[0], tuple):
        groups = [(1, groups[0], groups[1])]
    dims = [oe.level_dims(l) for l in range(nlevels)]
    lw = np.array([d[0] for d in dims], np.int32)
    lh = np.array([d[1] for d in dims], np.int32)
    pitch = (lw + 32 + 63) // 64 * 64
    planes = np.full(int((pitch.astype(np.int64) * (lh + 32)).sum()), 0xEE, np.uint8)
    off = np.zeros(nlevels, np.int64)
    pitch_out = np.zeros(nlevels, np.int32)
    stats = np.zeros(3, np.int64)
    gf, gx, gy = (np.array([g[i] for g in groups], np.int32) for i in range(3))
    rc = E.emu_pyr_tiles_run(img.ctypes.data, img.strides[0], nlevels, lw.ctypes.data, lh.ctypes.data, ring, len(groups), gf.ctypes.data, gx.ctypes.data, gy.ctypes.data,
                             max_lds, planes.ctypes.data, off.ctypes.data, pitch_out.ctypes.data, stats.ctypes.data)
    out = []
    if rc == 0:
        for l in range(nlevels):
            out.append(planes[off[l]:off[l] + int(pitch_out[l]) * (int(lh[l]) + 32)].reshape(int(lh[l]) + 32, int(pitch_out[l]))[:, :int(lw[l]) + 32])
    return rc, out, stats


def test_pyramid_tile_plan_executed_on_the_host_gives_the_oracle_pyramid(oracle, synth):
    """The fused pyramid launches (k_pyr_tiles, csrc/pyramid.hip) walk plans compiled on the host (csrc/pyr_tiles.hpp): the levels are cut
    into groups, one launch each; per (tile, level of the group) the region a workgroup computes -- its own cell of a TX x TY partition plus
    what its cell of the next level reads -- and the LDS tile it keeps for the next level.  tests/emu/pyr_tiles_emu.cpp runs the plans on
    the host with an independent per-pixel restatement of cv::resize's two 11-bit passes and checks what the kernel relies on: every byte
    of a level's image and 4-pixel ring is stored by exactly one workgroup, no tap of non-zero weight reads an LDS byte its own workgroup
    has not written, tap windows and tiles stay inside the LDS allocation, the tiles of consecutive levels do not overlap -- and the planes
    must equal the oracle's ComputePyramid (src/ORBextractor.cc:963-1004) wherever they are written."""
    E = _pyr_tiles_emu()
    one = lambda tx, ty: [(1, tx, ty)]
    cases = [((640, 512), 1.2, 8, (one(2, 2), one(4, 4), one(8, 8), one(16, 16), one(12, 10), one(3, 7), [(1, 4, 4), (4, 1, 1)], [(1, 8, 8), (3, 4, 4), (5, 2, 2)],
                                   [(l, 3, 2) for l in range(1, 8)], [(1, 4, 4), (3, 2, 2), (5, 1, 1)])),
             ((321, 243), 1.2, 8, (one(1, 1), one(5, 4), one(8, 8), [(1, 2, 2), (2, 1, 1)])),
             ((752, 480), 1.2, 8, (one(4, 4), one(10, 6), [(1, 6, 4), (4, 2, 2)])), ((200, 180), 1.1, 6, (one(1, 1), one(6, 6), [(1, 3, 3), (5, 1, 1)])),
             ((333, 222), 1.33, 5, (one(2, 3), one(8, 8))), ((1920, 1080), 1.2, 8, (one(8, 4), one(16, 9), [(1, 8, 8), (3, 4, 4), (5, 2, 2)])),
             ((1241, 376), 1.2, 8, (one(8, 2), one(12, 4))), ((97, 131), 1.2, 3, (one(1, 1), one(4, 4), one(16, 16))), ((4096, 600), 1.2, 4, (one(16, 2),))]
    for (w, h), scale, nl, plans in cases:
        img = synth.make_frame(4000 + w, w, h, n_shapes=60)
        oe = oracle.extractor(500, scale, nl, 20)
        oe(img)
        ref = [oe.level_plane(l) for l in range(nl)]
        for groups in plans:
            rc, planes, stats = _run_pyr_tiles(E, oe, img, nl, groups)
            assert rc == 0, ((w, h), scale, groups, rc)
            for l in range(1, nl):
                lw, lh = oe.level_dims(l)
                x1 = (16 + lw + 4 + 3) // 4 * 4
                np.testing.assert_array_equal(planes[l][12:16 + lh + 4, 12:min(x1, lw + 32)], ref[l][12:16 + lh + 4, 12:min(x1, lw + 32)],
                                              err_msg="%dx%d scale %.2f groups %s level %d" % (w, h, scale, groups, l))
                assert (planes[l][:12] == 0xEE).all() and (planes[l][16 + lh + 4:] == 0xEE).all()    # nothing outside the ring is touched
            if all(g[1:] == (1, 1) for g in groups):
                assert stats[0] < 1.03 * stats[1], stats      # one tile: nothing is computed twice (row groups of 4 round up)
    # the benchmark shape: what the halo costs -- deep groups pay for it, shallow groups of large tiles hardly
    img = synth.make_frame(4001, 640, 512)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    oe(img)
    for groups, worst in ((one(4, 4), 1.6), (one(8, 8), 2.4), ([(1, 4, 4), (3, 2, 2), (5, 1, 1)], 1.15), ([(1, 4, 4), (4, 1, 1)], 1.25)):
        rc, _, stats = _run_pyr_tiles(E, oe, img, 8, groups, max_lds=160 * 1024)
        assert rc == 0 and stats[0] < worst * stats[1], (groups, stats, stats[0] / stats[1])
        print("pyramid tiles", groups, "computed / owned = %.3f" % (stats[0] / stats[1]), "LDS", stats[2])
    # a scale factor whose taps leave the 12-byte window: refused (the per-level launches gather bytes there)
    img = synth.make_frame(4002, 512, 384)
    oe = oracle.extractor(500, 2.0, 3, 20)
    oe(img)
    assert _run_pyr_tiles(E, oe, img, 3, (4, 4))[0] == 2
    # a tile that does not fit the LDS budget: refused
    img = synth.make_frame(4003, 1920, 1080)
    oe = oracle.extractor(500, 1.2, 8, 20)
    oe(img)
    assert _run_pyr_tiles(E, oe, img, 8, (1, 1), max_lds=64 * 1024)[0] == 3
