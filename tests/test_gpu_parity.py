"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact bar: pyramid / blurred planes, FAST candidates (as sets), keypoints (x, y, size, angle, response, octave,
class_id as raw bytes) and 256-bit descriptors."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kp_bytes(kp):
    return np.ascontiguousarray(kp).view(np.uint8).reshape(len(kp), -1)


def _assert_same_features(kp_g, de_g, kp_o, de_o, what=""):
    assert len(kp_g) == len(kp_o), "%s: %d keypoints on GPU, %d in oracle" % (what, len(kp_g), len(kp_o))
    if len(kp_g) == 0:
        return
    bad = np.nonzero((_kp_bytes(kp_g) != _kp_bytes(kp_o)).any(1))[0]
    assert len(bad) == 0, "%s: %d keypoints differ, first %d: gpu=%s oracle=%s" % (what, len(bad), bad[0], kp_g[bad[0]], kp_o[bad[0]])
    badd = np.nonzero((de_g != de_o).any(1))[0]
    assert len(badd) == 0, "%s: %d descriptors differ, first at %d" % (what, len(badd), badd[0])


@pytest.fixture(scope="module")
def frames(synth):
    return [synth.make_frame(1000 + i) for i in range(3)]


@pytest.mark.parametrize("fast_th", [20, 7])
def test_stages_and_full_detect_640x512(uvo, oracle, frames, fast_th):
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, fast_th, max_width=640, max_height=512)
    oe = oracle.extractor(1000, 1.2, 8, fast_th)
    np.testing.assert_array_equal(ex.mvScaleFactor, oe.scale)
    np.testing.assert_array_equal(ex.mvInvScaleFactor, oe.inv_scale)
    np.testing.assert_array_equal(ex.mnFeaturesPerLevel, oe.quota)
    np.testing.assert_array_equal(ex.umax, oe.umax)
    for fi, img in enumerate(frames):
        kp_g, de_g = ex(img)
        kp_o, de_o = oe(img)
        for l in range(8):
            assert ex.level_dims(l) == oe.level_dims(l)
            np.testing.assert_array_equal(ex.read_plane(l), oe.level_plane(l), err_msg="pyramid level %d frame %d" % (l, fi))
            c_g = ex.read_candidates(l)
            c_o = oe.level_candidates(l)
            set_g = sorted(map(tuple, c_g.tolist()))
            set_o = sorted(zip(c_o["x"].astype(int).tolist(), c_o["y"].astype(int).tolist(), c_o["response"].astype(int).tolist()))
            assert set_g == set_o, "FAST candidates level %d frame %d: %d vs %d" % (l, fi, len(set_g), len(set_o))
            # the oracle blurs only levels that kept keypoints; compare the interior + the 2-px ring the descriptor can reach
            bo = oe.level_plane(l, blurred=True)
            bg = ex.read_plane(l, blurred=True)
            if (kp_o["octave"] == l).any():
                np.testing.assert_array_equal(bg[14:-14, 14:-14], bo[14:-14, 14:-14], err_msg="blurred level %d frame %d" % (l, fi))
        _assert_same_features(kp_g, de_g, kp_o, de_o, "frame %d fastTh %d" % (fi, fast_th))
    ex.close()


def test_batch_equals_single(uvo, oracle, synth):
    imgs = synth.make_batch(6, 640, 512, seed0=2000)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512, max_batch=6)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    res = ex.extract_batch(imgs)
    for i, (kp_g, de_g) in enumerate(res):
        kp_o, de_o = oe(imgs[i])
        _assert_same_features(kp_g, de_g, kp_o, de_o, "batch frame %d" % i)
    ex.close()


@pytest.mark.parametrize("shape,nfeat,th", [((480, 752), 1000, 7), ((376, 1241), 1500, 12), ((600, 420), 500, 20), ((96, 128), 300, 10)])
def test_other_resolutions(uvo, oracle, synth, shape, nfeat, th):
    h, w = shape
    img = synth.make_frame(77, w, h, n_shapes=max(40, w * h // 800))
    nlev = 8 if min(h, w) >= 300 else 3
    ex = uvo.ORBextractor(nfeat, 1.2, nlev, 0, th, max_width=w, max_height=h)
    oe = oracle.extractor(nfeat, 1.2, nlev, th)
    kp_g, de_g = ex(img)
    kp_o, de_o = oe(img)
    _assert_same_features(kp_g, de_g, kp_o, de_o, "%dx%d" % (w, h))
    ex.close()


@pytest.mark.parametrize("scale,nlev", [(1.1, 6), (1.3, 5), (1.5, 4), (2.0, 3)])
def test_other_scale_factors(uvo, oracle, synth, scale, nlev):
    """Pyramid scale factors on both sides of the 12-byte-window limit of k_resize_level (byte-gather path above ~1.33)."""
    w, h = 640, 512
    img = synth.make_frame(91, w, h)
    ex = uvo.ORBextractor(800, scale, nlev, 0, 15, max_width=w, max_height=h)
    oe = oracle.extractor(800, scale, nlev, 15)
    kp_g, de_g = ex(img)
    kp_o, de_o = oe(img)
    for l in range(nlev):
        np.testing.assert_array_equal(ex.read_plane(l), oe.level_plane(l), err_msg="scale %.2f level %d" % (scale, l))
    _assert_same_features(kp_g, de_g, kp_o, de_o, "scale %.2f" % scale)
    ex.close()


def test_degenerate_images(uvo, oracle):
    ex = uvo.ORBextractor(500, 1.2, 8, 0, 20, max_width=640, max_height=512)
    oe = oracle.extractor(500, 1.2, 8, 20)
    rng = np.random.default_rng(5)
    flat = np.full((512, 640), 128, np.uint8)
    noise = rng.integers(0, 256, (512, 640), dtype=np.uint8)            # corner everywhere: stresses candidate capacity + NMS ties
    checker = ((np.indices((512, 640)).sum(0) // 2) % 2 * 255).astype(np.uint8)
    low = (rng.integers(0, 12, (512, 640)) + 100).astype(np.uint8)     # only the threshold-7 fallback can fire
    for name, img in (("flat", flat), ("noise", noise), ("checker", checker), ("lowcontrast", low)):
        kp_g, de_g = ex(img)
        kp_o, de_o = oe(img)
        _assert_same_features(kp_g, de_g, kp_o, de_o, name)
    ex.close()


def _fast_mode_images(synth):
    """Frames that put the per-cell fallback of src/ORBextractor.cc:792-799 in every state: no cell falls back (textured), all do (flat,
    low contrast), some do (textured left half / low-contrast right half; a dim frame), cells whose only corners >= fastTh are
    equal-score neighbours that suppress each other (so `FAST(cell, 20)` is empty although pixels score >= 20, and in the second
    call those pixels still suppress their weaker neighbours), and dense noise."""
    rng = np.random.default_rng(11)
    H, W = 512, 640
    tex = synth.make_frame(4242, W, H)
    low = (rng.integers(0, 12, (H, W)) + 100).astype(np.uint8)
    half = tex.copy()
    half[:, W // 2:] = low[:, W // 2:]
    dim = (tex.astype(np.float32) * 0.22 + 90).astype(np.uint8)   # contrast cut to a fifth: most corners score below 20
    # pairs of identical bright 2x1 blobs on a flat background: both pixels of a pair get the same score and annihilate
    ties = np.full((H, W), 100, np.uint8)
    ties += rng.integers(0, 9, (H, W)).astype(np.uint8)           # weak corners (score >= 7) around them
    for y in range(40, H - 40, 37):
        for x in range(40, W - 40, 41):
            ties[y, x] = ties[y, x + 1] = 190
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    flat = np.full((H, W), 77, np.uint8)
    return (("textured", tex), ("lowcontrast", low), ("half", half), ("dim", dim), ("ties", ties), ("noise", noise), ("flat", flat))


@pytest.mark.parametrize("mode", ["two_pass", "single_pass", "adaptive"])
def test_fast_modes_give_the_reference_candidates(uvo, oracle, synth, mode):
    """UVO_TUNE_FAST_MODE: the threshold-adaptive two-pass form (stream at fastTh + sparse literal-7 pass over the empty cells), the
    single pass at 7 with the per-cell vote, and the adaptive choice between them must all produce the reference's per-cell
    `FAST(cell, fastTh)`, else `FAST(cell, 7)` candidates (as sets, per level) and the same final keypoints / descriptors."""
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512)
    ex.tune(uvo.UVO_TUNE_FAST_MODE, {"two_pass": uvo.UVO_FAST_MODE_TWO_PASS, "single_pass": uvo.UVO_FAST_MODE_SINGLE_PASS,
                                     "adaptive": uvo.UVO_FAST_MODE_ADAPTIVE}[mode])
    oe = oracle.extractor(1000, 1.2, 8, 20)
    seen_fallback = False
    for rep in range(2 if mode == "adaptive" else 1):   # adaptive: the second round starts from whatever the first left behind
        for name, img in _fast_mode_images(synth):
            kp_g, de_g = ex(img)
            kp_o, de_o = oe(img)
            for l in range(8):
                c_g = ex.read_candidates(l)
                c_o = oe.level_candidates(l)
                set_g = sorted(map(tuple, c_g.tolist()))
                set_o = sorted(zip(c_o["x"].astype(int).tolist(), c_o["y"].astype(int).tolist(), c_o["response"].astype(int).tolist()))
                assert set_g == set_o, "%s / %s: FAST candidates of level %d: %d vs %d" % (mode, name, l, len(set_g), len(set_o))
            _assert_same_features(kp_g, de_g, kp_o, de_o, "%s / %s" % (mode, name))
            t, fb, cells = ex.fast_state()
            assert (fb >= 0).all() and (fb <= cells).all(), (name, fb, cells)
            if name in ("lowcontrast", "flat"):
                assert (fb == cells).all(), "every cell of a %s frame falls back: %s of %s" % (name, fb, cells)
            if name == "textured":
                assert fb.sum() * 10 < cells.sum(), "a textured frame should leave few cells empty at fastTh: %s of %s" % (fb, cells)
            seen_fallback = seen_fallback or fb.sum() > 0
            if mode == "two_pass":
                assert (t == 20).all()
            elif mode == "single_pass":
                assert (t == 7).all()
            else:   # hysteresis of k_octree's last workgroup: > 22 % fall-back cells -> one pass, < 14 % -> two
                for l in range(8):
                    if fb[l] * 100 > cells[l] * 22:
                        assert t[l] == 7, (name, l, fb[l], cells[l], t[l])
                    elif fb[l] * 100 < cells[l] * 14:
                        assert t[l] == 20, (name, l, fb[l], cells[l], t[l])
    assert seen_fallback
    ex.close()


@pytest.mark.parametrize("shape,nfeat,th,nlev", [((480, 752), 1000, 12, 8), ((1080, 1920), 2000, 20, 8), ((200, 1300), 600, 30, 4), ((96, 128), 300, 10, 3), ((85, 118), 200, 15, 2)])
def test_two_pass_fast_at_other_geometries(uvo, oracle, synth, shape, nfeat, th, nlev):
    """The sparse per-cell pass on other cell sizes / grids (two quad-tree roots, a very wide image, a tiny one, one whose levels are a
    single row of cells up to 65 px high: the large LDS geometry of k_fast_cells), on a frame whose
    contrast fades from left to right so that every level has cells of both kinds; both forced modes against the oracle."""
    h, w = shape
    img = synth.make_frame(123, w, h, n_shapes=max(40, w * h // 800)).astype(np.float32)
    fade = np.linspace(1.0, 0.05, w, dtype=np.float32)[None, :]
    img = (img * fade + 110 * (1 - fade)).astype(np.uint8)
    oe = oracle.extractor(nfeat, 1.2, nlev, th)
    kp_o, de_o = oe(img)
    for mode in (uvo.UVO_FAST_MODE_TWO_PASS, uvo.UVO_FAST_MODE_SINGLE_PASS):
        ex = uvo.ORBextractor(nfeat, 1.2, nlev, 0, th, max_width=w, max_height=h)
        ex.tune(uvo.UVO_TUNE_FAST_MODE, mode)
        kp_g, de_g = ex(img)
        for l in range(nlev):
            c_g, c_o = ex.read_candidates(l), oe.level_candidates(l)
            assert sorted(map(tuple, c_g.tolist())) == sorted(zip(c_o["x"].astype(int).tolist(), c_o["y"].astype(int).tolist(), c_o["response"].astype(int).tolist())), \
                "mode %d level %d" % (mode, l)
        _assert_same_features(kp_g, de_g, kp_o, de_o, "%dx%d mode %d" % (w, h, mode))
        t, fb, cells = ex.fast_state()
        assert (fb <= cells).all() and (cells.sum() < 50 or 0 < fb.sum() < cells.sum()), (fb, cells)   # (the tiny image has 10 cells)
        ex.close()


@pytest.mark.parametrize("th", [1, 2, 4, 6, 8, 9, 13, 21, 60, 127, 128, 200, 254, 255, 300])
def test_fast_threshold_sweep_with_extreme_pixels(uvo, oracle, synth, th):
    """The streaming screen of k_fast_score compares pixels at seven bits ((t + 1) >> 1 against p >> 1 - v >> 1, bit 7 of every byte as
    the carry stop): every parity of t, the thresholds where v + t or v - t leave the byte range, and frames that hold the extreme
    values next to each other (0 / 1 / 254 / 255 blobs on dark, bright and mid-grey ground, saturated noise) -- candidates of every level
    and the final features against the oracle, in both FAST forms."""
    rng = np.random.default_rng(1000 + th)
    H, W = 240, 320
    base = synth.make_frame(555, W, H, n_shapes=120)
    ext = np.choose(rng.integers(0, 3, (H, W)), [np.full((H, W), 3, np.uint8), np.full((H, W), 128, np.uint8), np.full((H, W), 252, np.uint8)])
    ext = (ext.astype(np.int16) + rng.integers(-3, 4, (H, W))).clip(0, 255).astype(np.uint8)
    for _ in range(600):
        y, x = int(rng.integers(20, H - 20)), int(rng.integers(20, W - 20))
        ext[y:y + int(rng.integers(1, 5)), x:x + int(rng.integers(1, 5))] = int(rng.choice([0, 1, 2, 127, 128, 129, 253, 254, 255]))
    sat = rng.choice(np.array([0, 255], np.uint8), (H, W), p=[0.7, 0.3])
    oe = oracle.extractor(500, 1.2, 4, th)
    for mode in (uvo.UVO_FAST_MODE_TWO_PASS, uvo.UVO_FAST_MODE_SINGLE_PASS):
        ex = uvo.ORBextractor(500, 1.2, 4, 0, th, max_width=W, max_height=H)
        ex.tune(uvo.UVO_TUNE_FAST_MODE, mode)
        for name, img in (("textured", base), ("extremes", ext), ("saturated", sat)):
            kp_g, de_g = ex(img)
            kp_o, de_o = oe(img)
            for l in range(4):
                c_g, c_o = ex.read_candidates(l), oe.level_candidates(l)
                assert sorted(map(tuple, c_g.tolist())) == sorted(zip(c_o["x"].astype(int).tolist(), c_o["y"].astype(int).tolist(), c_o["response"].astype(int).tolist())), \
                    "fastTh %d mode %d %s level %d" % (th, mode, name, l)
            _assert_same_features(kp_g, de_g, kp_o, de_o, "fastTh %d mode %d %s" % (th, mode, name))
        ex.close()


def test_blur_planes_at_the_saturation_edge(uvo, oracle):
    """k_gauss7's column pass runs on the fp32 pipe: exact while the result is not saturated, i.e. the images to try are the ones whose
    sums sit at and across 2^24 -- constant 253 / 254 / 255 (taps sum to 257 per pass, so 254 already blurs to 255), bright noise,
    0 / 255 salt, a bright ramp -- plus corners in them so that the oracle blurs every level (it skips levels without keypoints)."""
    rng = np.random.default_rng(77)
    H, W = 256, 320
    yy, xx = np.indices((H, W))
    marks = (rng.random((H, W)) < 0.02)
    imgs = {}
    for c in (253, 254, 255):
        im = np.full((H, W), c, np.uint8)
        im[marks] = 0
        imgs["const%d" % c] = im
    imgs["bright_noise"] = rng.integers(236, 256, (H, W)).astype(np.uint8)
    imgs["salt"] = np.where(rng.random((H, W)) < 0.5, 255, 0).astype(np.uint8)
    ramp = np.clip(200 + (xx + yy) // 6, 0, 255).astype(np.uint8)
    ramp[marks] = 40
    imgs["bright_ramp"] = ramp
    ex = uvo.ORBextractor(500, 1.2, 6, 0, 20, max_width=W, max_height=H)
    oe = oracle.extractor(500, 1.2, 6, 20)
    for name, img in imgs.items():
        kp_g, de_g = ex(img)
        kp_o, de_o = oe(img)
        compared = 0
        for l in range(6):
            if (kp_o["octave"] == l).any():
                bo, bg = oe.level_plane(l, blurred=True), ex.read_plane(l, blurred=True)
                np.testing.assert_array_equal(bg[14:-14, 14:-14], bo[14:-14, 14:-14], err_msg="%s: blurred level %d" % (name, l))
                compared += 1
        assert compared >= 1 or len(kp_o) == 0, name
        _assert_same_features(kp_g, de_g, kp_o, de_o, name)
    ex.close()


@pytest.mark.parametrize("shape", [(640, 512), (638, 510), (321, 243)])
def test_blur_rounding_contracts(uvo, oracle, synth, shape):
    """UVO_TUNE_BLUR_ROUNDING: the generic column filter rounds an exact .5 up on every column; an x86-64 OpenCV build's SSE2 column filter
    rounds it to even on the image columns 0 .. (w & ~3) - 1 and up on the last w % 4 (src/ORBextractor.cc:942).  The HIP
    path follows whichever contract is selected, byte for byte -- blurred planes and features -- on images that hold such ties (the two
    oracles differ on them), including widths with a scalar tail, saturating images, and ties planted in the tail columns."""
    w, h = shape
    rng = np.random.default_rng(w)
    imgs = [synth.make_frame(8100 + w, w, h, n_shapes=max(60, w * h // 1500)), rng.integers(0, 256, (h, w)).astype(np.uint8),
            rng.integers(230, 256, (h, w)).astype(np.uint8)]
    ex = uvo.ORBextractor(800, 1.2, 6, 0, 15, max_width=w, max_height=h)
    oe = oracle.extractor(800, 1.2, 6, 15)
    differing = 0
    for img in imgs:
        planes = {}
        for mode, tune in ((0, uvo.UVO_BLUR_ROUNDING_SCALAR), (1, uvo.UVO_BLUR_ROUNDING_SSE2)):
            oe.set_blur_rounding(mode)
            ex.tune(uvo.UVO_TUNE_BLUR_ROUNDING, tune)
            kp_g, de_g = ex(img)
            kp_o, de_o = oe(img)
            for l in range(6):
                if (kp_o["octave"] == l).any():
                    bo, bg = oe.level_plane(l, blurred=True), ex.read_plane(l, blurred=True)
                    np.testing.assert_array_equal(bg[14:-14, 14:-14], bo[14:-14, 14:-14], err_msg="contract %d: blurred level %d" % (mode, l))
                    planes[(mode, l)] = bo[16:-16, 16:-16]
            _assert_same_features(kp_g, de_g, kp_o, de_o, "contract %d" % mode)
        for l in range(6):
            if (0, l) in planes and (1, l) in planes:
                d = planes[(0, l)] != planes[(1, l)]
                differing += int(d.sum())
                lw = planes[(0, l)].shape[1]
                assert not d[:, lw & ~3:].any()          # the scalar tail rounds the same under both contracts
                assert (np.abs(planes[(0, l)].astype(int) - planes[(1, l)].astype(int)) <= 1).all()
    assert differing > 0, "no exact tie in these images: the test would prove nothing"
    ex.close()


def test_topup_mode(uvo, oracle, frames):
    """FullDetect=false: caller keypoints pass through level 0, occupancy grid filters and is mutated (src/ORBextractor.cc:872-909)."""
    img = frames[0]
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512, max_input_keypoints=600)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    rng = np.random.default_rng(11)
    min_px = 20
    rows, cols = 512 // min_px + 2, 640 // min_px + 2
    for n_in, need in ((0, 1000), (300, 700), (550, 450), (5, 40)):
        kin = np.zeros(n_in, uvo.KEYPOINT_DTYPE)
        kin["x"] = rng.uniform(20, 619, n_in).astype(np.float32)
        kin["y"] = rng.uniform(20, 491, n_in).astype(np.float32)
        kin["size"], kin["angle"], kin["response"], kin["octave"], kin["class_id"] = 31, -1, rng.uniform(0, 99, n_in), 0, np.arange(n_in)
        grid = np.zeros((rows, cols), np.int32, order="F")
        for k in kin:
            grid[int(k["y"] / min_px), int(k["x"] / min_px)] += 1
        g_gpu, g_orc = grid.copy(order="F"), grid.copy(order="F")
        kp_g, de_g = ex(img, kin.copy(), g_gpu, min_px, False, need)
        kp_o, de_o = oe(img, kin.copy(), g_orc, min_px, False, need)
        _assert_same_features(kp_g, de_g, kp_o, de_o, "topup n_in=%d need=%d" % (n_in, need))
        np.testing.assert_array_equal(g_gpu, g_orc)
    ex.close()


def test_hamming_knn2_and_matrix(uvo, oracle):
    rng = np.random.default_rng(3)
    m = uvo.ORBmatcher(0.8, max_query=2048, max_train=2048)
    for nq, nt in ((1000, 1000), (1, 1), (257, 3), (5, 0), (0, 7), (64, 2000)):
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        if nq and nt > 4:
            t[: min(nt, nq) // 2] = q[: min(nt, nq) // 2]       # exact matches
            t[1] = t[0]                                         # duplicated train rows: tie must keep the lower index
            q[-1] = 0
            t[-1] = 255                                         # distance 256
        idx0, d0, idx1, d1 = m.knn2(q, t)
        o = oracle.knn2(q, t)
        np.testing.assert_array_equal(idx0, o[0])
        np.testing.assert_array_equal(idx1, o[2])
        np.testing.assert_array_equal(d0.astype(np.int32), np.where(o[0] < 0, 0xFFFF, o[1]))
        if nq and nt:
            # low-entropy descriptors: almost every best / second-best is a tie that the lower train index must win
            ql = rng.choice(np.array([0, 1, 128, 255], np.uint8), (nq, 32), p=[0.7, 0.1, 0.1, 0.1])
            tl = rng.choice(np.array([0, 1, 128, 255], np.uint8), (nt, 32), p=[0.7, 0.1, 0.1, 0.1])
            gi0, gd0, gi1, gd1 = m.knn2(ql, tl)
            ol = oracle.knn2(ql, tl)
            np.testing.assert_array_equal(gi0, ol[0])
            np.testing.assert_array_equal(gi1, ol[2])
            np.testing.assert_array_equal(gd0.astype(np.int32), np.where(ol[0] < 0, 0xFFFF, ol[1]))
            np.testing.assert_array_equal(gd1.astype(np.int32), np.where(ol[2] < 0, 0xFFFF, ol[3]))
        np.testing.assert_array_equal(d1.astype(np.int32), np.where(o[2] < 0, 0xFFFF, o[3]))
        if nq and nt:
            dm = m.distance_matrix(q, t)
            ref = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
            np.testing.assert_array_equal(dm, ref)
    # masked
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    mask = (rng.random((300, 200)) < 0.3).astype(np.uint8)
    mask[7] = 0
    mask[8] = 0
    mask[8, 5] = 1
    idx0, d0, idx1, d1 = m.knn2(q, t, mask)
    o = oracle.knn2(q, t, mask)
    np.testing.assert_array_equal(idx0, o[0])
    np.testing.assert_array_equal(idx1, o[2])
    m.close()


def test_extract_then_match_consecutive_frames(uvo, oracle, synth):
    a = synth.make_frame(4242)
    b = synth.warp_frame(a, 1)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512, max_batch=2)
    (kp_a, de_a), (kp_b, de_b) = ex.extract_batch(np.stack([a, b]))
    m = uvo.ORBmatcher(0.8)
    got = m.ratio_matching(de_a, de_b, 0.8)
    o = oracle.knn2(de_a, de_b)
    ok = (o[2] >= 0) & (o[1].astype(np.float64) <= o[3].astype(np.float64) * 0.8)
    ref = np.stack([np.nonzero(ok)[0], o[0][ok], o[1][ok]], 1)
    np.testing.assert_array_equal(got, ref)
    assert len(got) > 100  # the warp is small: most features must find their partner
    ex.close()
    m.close()


def test_search_by_projection(uvo, oracle, synth):
    rng = np.random.default_rng(9)
    img = synth.make_frame(31337, 752, 480)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 7, max_width=752, max_height=480)
    kp, de = ex(img)
    n = len(kp)
    # synthetic local map: 5000 points, 40 % true correspondences with ~6 % bit flips
    M = 5000
    src = rng.integers(0, n, M)
    true = rng.random(M) < 0.4
    mp_desc = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    flips = (rng.random((M, 256)) < 0.06)
    noisy = np.packbits(np.unpackbits(de[src], axis=1) ^ flips, axis=1)
    mp_desc[true] = noisy[true]
    px = (kp["x"][src] + rng.normal(0, 1.5, M)).astype(np.float32)
    py = (kp["y"][src] + rng.normal(0, 1.5, M)).astype(np.float32)
    level = np.clip(kp["octave"][src] + rng.integers(-1, 2, M), 0, 7).astype(np.int32)
    vc = np.where(rng.random(M) < 0.5, 0.999, 0.9).astype(np.float32)
    inview = (rng.random(M) < 0.9).astype(np.uint8)
    m = uvo.ORBmatcher(0.8, max_query=4096, max_map_points=8192)
    for th in (1.0, 5.0):
        a_g = np.full(n, -1, np.int32)
        a_g[rng.integers(0, n, 30)] = 123456  # some keypoints already hold a map point
        a_o = a_g.copy()
        nm_g = m.SearchByProjection(kp, de, (0, 0, 752, 480), a_g, px, py, level, vc, inview, mp_desc, ex.mvScaleFactor, th)
        nm_o = oracle.search_by_projection(kp, de, (0, 0, 752, 480), a_o, px, py, level, vc, inview, mp_desc, ex.mvScaleFactor, th, 0.8)
        assert nm_g == nm_o
        np.testing.assert_array_equal(a_g, a_o)
        assert nm_g > 300
    ex.close()
    m.close()


@pytest.mark.parametrize("matcher_stream", ["own", "lane", "lane_once"])
def test_hbm_resident_pipeline_depth2(uvo, oracle, synth, matcher_stream):
    """uvo_extract_batch_device with two alternating scratch sets / streams feeding the batched HBM-resident matcher -- on its own stream
    behind events, or attached to the extracting lane's stream; a host-buffer matcher call while attached waits for that lane.
    lane_once: attached ONCE before the first batch -- the extractor moves its followers along when a batch goes to the other lane."""
    import torch
    B, W, H = 4, 640, 512
    batches = [synth.make_batch(B, W, H, seed0=3000 + 10 * k) for k in range(3)]
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=B)
    ex.set_pipeline(2)
    cap = ex.cap
    mt = uvo.ORBmatcher(0.8, max_query=cap, max_train=cap, max_batch=B)
    dev = torch.device("cuda", 0)
    outs = []
    for k in range(3):
        d_img = torch.from_numpy(batches[k]).to(dev)
        kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        de = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(B, dtype=torch.int32, device=dev)
        i0 = torch.zeros((B, cap), dtype=torch.int32, device=dev)
        i1 = torch.zeros((B, cap), dtype=torch.int32, device=dev)
        d0 = torch.zeros((B, cap), dtype=torch.int16, device=dev)
        d1 = torch.zeros((B, cap), dtype=torch.int16, device=dev)
        outs.append((d_img, kp, de, n, i0, i1, d0, d1))
    torch.cuda.synchronize()
    if matcher_stream == "lane_once":
        mt.attach(ex)
    for d_img, kp, de, n, i0, i1, d0, d1 in outs:   # three calls back to back, no host sync in between
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap)
        if matcher_stream == "own":
            mt.wait_extractor(ex)
        elif matcher_stream == "lane":
            mt.attach(ex)
        # pair p = (frame p, frame p+1) for p < B-1
        mt.knn2_batch_device(B - 1, de.data_ptr(), n.data_ptr(), cap, de.data_ptr() + cap * 32, n.data_ptr() + 4, cap, i0.data_ptr(),
                             d0.data_ptr(), i1.data_ptr(), d1.data_ptr())
        if matcher_stream == "own":
            mt.release_to_extractor(ex)
    if matcher_stream != "own":   # a host-buffer call in the attached state: enqueued in, and waiting for, the lane's stream
        rng = np.random.default_rng(5)
        a, b = rng.integers(0, 256, (70, 32), dtype=np.uint8), rng.integers(0, 256, (90, 32), dtype=np.uint8)
        got = mt.knn2(a, b)
        want = oracle.knn2(a, b)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1].astype(np.int32), want[1])
    ex.synchronize()
    mt.synchronize()
    mt.attach(None)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    for k, (d_img, kp, de, n, i0, i1, d0, d1) in enumerate(outs):
        n_h = n.cpu().numpy()
        ref = [oe(batches[k][b]) for b in range(B)]
        for b in range(B):
            kp_g = kp[b, :n_h[b]].cpu().numpy().view(uvo.KEYPOINT_DTYPE).reshape(-1)
            _assert_same_features(kp_g, de[b, :n_h[b]].cpu().numpy(), ref[b][0], ref[b][1], "call %d frame %d" % (k, b))
        for b in range(B - 1):
            o = oracle.knn2(ref[b][1], ref[b + 1][1])
            nq = len(ref[b][1])
            np.testing.assert_array_equal(i0[b, :nq].cpu().numpy(), o[0])
            np.testing.assert_array_equal(i1[b, :nq].cpu().numpy(), o[2])
            np.testing.assert_array_equal(d0[b, :nq].cpu().numpy().astype(np.uint16).astype(np.int32), o[1])
    ex.close()
    mt.close()


def test_config4_hd_1920x1080_2000_features(uvo, oracle, synth):
    """BASELINE.json configs[3] geometry: 1920x1080 @ 2000 features (two quad-tree roots per level, 6594 FAST cells)."""
    imgs = np.stack([synth.make_frame(7000 + i, 1920, 1080, n_shapes=2500) for i in range(2)])
    ex = uvo.ORBextractor(2000, 1.2, 8, 0, 20, max_width=1920, max_height=1080, max_batch=2)
    oe = oracle.extractor(2000, 1.2, 8, 20)
    assert ex.mnFeaturesPerLevel.tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    for i, (kp_g, de_g) in enumerate(ex.extract_batch(imgs)):
        kp_o, de_o = oe(imgs[i])
        _assert_same_features(kp_g, de_g, kp_o, de_o, "HD frame %d" % i)
        assert len(kp_g) >= 2000
    ex.close()


def test_config1_harbor_parameters(uvo, oracle, synth):
    """Data/Settings_VI_Aqualoc_harbor.yaml:67-79 as shipped: nFeatures 400, scaleFactor 1.2, nLevels 8, fastTh 20, Px_distance 20;
    top-up call as src/Tracking.cc:946 makes it in WORKING state (FullDetect = false)."""
    img = synth.make_frame(4711, 640, 512)
    ex = uvo.ORBextractor(400, 1.2, 8, 0, 20, max_width=640, max_height=512, max_input_keypoints=800)
    oe = oracle.extractor(400, 1.2, 8, 20)
    assert ex.mnFeaturesPerLevel.tolist() == [87, 72, 60, 50, 42, 35, 29, 25]
    rng = np.random.default_rng(1)
    n_in = 330                                   # tracked points; 70 missing (> 5 % of 400, src/Tracking.cc:931-935)
    kin = np.zeros(n_in, uvo.KEYPOINT_DTYPE)
    kin["x"], kin["y"] = rng.uniform(20, 619, n_in).astype(np.float32), rng.uniform(20, 491, n_in).astype(np.float32)
    kin["size"], kin["angle"], kin["octave"], kin["class_id"] = 31, -1, 0, -1
    rows, cols = 512 // 20 + 2, 640 // 20 + 2
    grid = np.zeros((rows, cols), np.int32, order="F")
    for k in kin:
        grid[int(k["y"] / 20), int(k["x"] / 20)] += 1
    g1, g2 = grid.copy(order="F"), grid.copy(order="F")
    kp_g, de_g = ex(img, kin.copy(), g1, 20, False, 400 - n_in)
    kp_o, de_o = oe(img, kin.copy(), g2, 20, False, 400 - n_in)
    _assert_same_features(kp_g, de_g, kp_o, de_o, "harbor top-up")
    np.testing.assert_array_equal(g1, g2)
    assert n_in < len(kp_g) <= 400
    ex.close()


@pytest.mark.parametrize("grid,nfeat,th,nms", [((8, 5), 200, 20, True), ((5, 3), 400, 10, True), ((4, 4), 100, 15, False), ((1, 1), 50, 25, True)])
def test_grider_fast_bucketing(uvo, oracle, synth, grid, nfeat, th, nms):
    """Grider_FAST::perform_griding (include/Grider_FAST.h:81-137) -- the alternative bucketing mode."""
    img = synth.make_frame(99, 320, 256, n_shapes=120)
    ex = uvo.ORBextractor(1000, 1.2, 4, 0, 20, max_width=320, max_height=256)   # output staging scales with nfeatures
    got = ex.grider_fast(img, nfeat, grid[0], grid[1], th, nms)
    ref = oracle.grider_fast(img, nfeat, grid[0], grid[1], th, nms)
    assert len(got) == len(ref) > 0
    assert got.tobytes() == ref.tobytes()
    ex.close()


@pytest.mark.parametrize("fast_th", [3, 5, 40, 0])
def test_threshold_extremes(uvo, oracle, synth, fast_th):
    """fastTh below the literal-7 fallback (t_min = fastTh, the fallback can then only shrink the set), far above it
    (most cells fall back to 7) and 0 (score-0 corners exist and must never survive NMS)."""
    img = synth.make_frame(424242, 320, 256, n_shapes=100)
    ex = uvo.ORBextractor(500, 1.2, 4, 0, fast_th, max_width=320, max_height=256)
    oe = oracle.extractor(500, 1.2, 4, fast_th)
    kp_g, de_g = ex(img)
    kp_o, de_o = oe(img)
    for l in range(4):
        c_g = sorted(map(tuple, ex.read_candidates(l).tolist()))
        c_o = oe.level_candidates(l)
        assert c_g == sorted(zip(c_o["x"].astype(int).tolist(), c_o["y"].astype(int).tolist(), c_o["response"].astype(int).tolist()))
    _assert_same_features(kp_g, de_g, kp_o, de_o, "fastTh %d" % fast_th)
    ex.close()


def test_distinctive_descriptors_batch(uvo, oracle):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:197-270): least-median-distance observation per map point."""
    rng = np.random.default_rng(17)
    lists = []
    for n in [1, 2, 3, 7, 20, 64, 257, 300, 0, 5]:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        d = np.repeat(base[None, :], n, 0)
        flips = rng.random((n, 256)) < rng.uniform(0.02, 0.3)
        d = np.packbits(np.unpackbits(d, axis=1) ^ flips, axis=1) if n else d
        if n >= 3:
            d[2] = d[1]  # duplicated observation: ties in the medians, first index must win
        lists.append(d)
    m = uvo.ORBmatcher(0.8)
    idx, med = m.distinctive_descriptors(lists)
    for p, d in enumerate(lists):
        ref = oracle.distinctive_descriptor(d) if len(d) else (-1, -1)
        assert (int(idx[p]), int(med[p])) == ref, "point %d (N=%d)" % (p, len(d))
    m.close()


# ---- the other ORBmatcher search loops (generic engine) ---------------------------------------------------------------
def _two_views(uvo, synth, seed, W=752, H=480, fast_th=7):
    """Two frames of a synthetic sequence (second = warped first) with their keypoints / descriptors."""
    a = synth.make_frame(seed, W, H)
    b = synth.warp_frame(a, seed + 1)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, fast_th, max_width=W, max_height=H)
    kp1, de1 = ex(a)
    kp2, de2 = ex(b)
    sf = ex.mvScaleFactor.copy()
    ex.close()
    return kp1, de1, kp2, de2, sf


def _noisy_copies(rng, de, src, flip_p=0.06, true_frac=0.5):
    M = len(src)
    out = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    true = rng.random(M) < true_frac
    noisy = np.packbits(np.unpackbits(de[src], axis=1) ^ (rng.random((M, 256)) < flip_p), axis=1)
    out[true] = noisy[true]
    return out


def _bow_groups(rng, de, n_nodes=60):
    """A stand-in vocabulary: node id = a hash of 6 descriptor bits (true matches mostly share a node), ids spread out."""
    bits = np.unpackbits(de, axis=1)[:, [3, 41, 77, 130, 201, 250]]
    node = (bits * (1 << np.arange(6))).sum(1) % n_nodes
    groups = {}
    for i in rng.permutation(len(de)):          # DBoW2 keeps features of a node in insertion order; any order must work
        groups.setdefault(int(node[i]) * 7 + 3, []).append(int(i))
    return groups


@pytest.mark.parametrize("check_ori", [False, True])
def test_search_by_projection_kf(uvo, oracle, synth, check_ori):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1622-1746)."""
    rng = np.random.default_rng(21)
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 4100)
    n, M = len(kp2), 1500
    src = rng.integers(0, n, M)
    mp_desc = _noisy_copies(rng, de2, src)
    u = (kp2["x"][src] + rng.normal(0, 2.0, M)).astype(np.float32)
    v = (kp2["y"][src] + rng.normal(0, 2.0, M)).astype(np.float32)
    level = np.clip(kp2["octave"][src] + rng.integers(-1, 2, M), 0, 7).astype(np.int32)
    valid = (rng.random(M) < 0.85).astype(np.uint8)
    kf_angle = np.where(rng.random(M) < 0.8, kp2["angle"][src] + rng.normal(0, 4, M), rng.uniform(0, 360, M)).astype(np.float32) % np.float32(360)
    m = uvo.ORBmatcher(0.9, check_ori, max_query=4096, max_map_points=8192)
    for th, orbdist in ((10.0, 100), (15.0, 64)):
        a_g = np.full(n, -1, np.int32)
        a_g[rng.integers(0, n, 25)] = 999999
        a_o = a_g.copy()
        nm_g = m.SearchByProjectionKF(kp2, de2, (0, 0, 752, 480), a_g, u, v, level, valid, mp_desc, kf_angle, sf, th, orbdist)
        nm_o = oracle.search_by_projection_kf(kp2, de2, (0, 0, 752, 480), a_o, u, v, level, valid, mp_desc, kf_angle, sf, th, orbdist, check_ori)
        np.testing.assert_array_equal(a_g, a_o)
        assert nm_g == nm_o and nm_g > 150
    m.close()


@pytest.mark.parametrize("kf_kf", [False, True])
def test_search_by_bow(uvo, oracle, synth, kf_kf):
    """SearchByBoW KF-Frame (:155-284) and KF-KF (:715-850) over a stand-in vocabulary."""
    rng = np.random.default_rng(22)
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 4200)
    g1, g2 = _bow_groups(rng, de1), _bow_groups(rng, de2)
    usable1 = (rng.random(len(kp1)) < 0.8).astype(np.uint8)
    usable2 = (rng.random(len(kp2)) < 0.85).astype(np.uint8) if kf_kf else None
    for ratio, ori in ((0.75, True), (0.9, False), (0.6, True)):
        m = uvo.ORBmatcher(ratio, ori)
        mg, ng = m.SearchByBoW(uvo.FeatureVector(g1), de1, kp1["angle"], usable1, uvo.FeatureVector(g2), de2, kp2["angle"], usable2, kf_kf=kf_kf)
        mo, no = oracle.search_by_bow(kf_kf, g1, de1, kp1["angle"], usable1, g2, de2, kp2["angle"], usable2, ratio, ori)
        np.testing.assert_array_equal(mg, mo)
        assert ng == no == int((mo >= 0).sum())
        m.close()
    assert no > 50


def test_search_for_triangulation(uvo, oracle, synth):
    """SearchForTriangulation (:852-1014) with the epipolar test of :136-153."""
    rng = np.random.default_rng(23)
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 4300)
    g1, g2 = _bow_groups(rng, de1, 40), _bow_groups(rng, de2, 40)
    has1 = (rng.random(len(kp1)) < 0.3).astype(np.uint8)
    has2 = (rng.random(len(kp2)) < 0.3).astype(np.uint8)
    sigma2 = (sf * sf).astype(np.float32)
    # a fundamental matrix of a near-pure translation along x (epipolar lines ~ horizontal), slightly perturbed
    F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 2e-4, (3, 3)).astype(np.float32))
    tot = 0
    for ori in (False, True):
        m = uvo.ORBmatcher(0.6, ori)
        for scale in (1.0, 400.0):                       # 400: wide acceptance band, many candidates pass
            s2 = (sigma2 * np.float32(scale)).astype(np.float32)
            mg, ng = m.SearchForTriangulation(uvo.FeatureVector(g1), kp1, de1, has1, uvo.FeatureVector(g2), kp2, de2, has2, F12, s2)
            mo, no = oracle.search_for_triangulation(g1, kp1, de1, has1, g2, kp2, de2, has2, F12, s2, ori)
            np.testing.assert_array_equal(mg, mo)
            assert ng == no
            tot += no
        m.close()
    assert tot > 40


def test_fuse_search_and_generic_windows(uvo, oracle, synth):
    """Search core of Fuse (:1077-1101) and the window engine used without exclusivity."""
    rng = np.random.default_rng(24)
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 4400)
    n, M = len(kp2), 3000
    src = rng.integers(0, n, M)
    mp_desc = _noisy_copies(rng, de2, src, 0.04)
    u = (kp2["x"][src] + rng.normal(0, 1.5, M)).astype(np.float32)
    v = (kp2["y"][src] + rng.normal(0, 1.5, M)).astype(np.float32)
    u[:20] = rng.uniform(-50, 800, 20).astype(np.float32)     # some windows partly or wholly outside the grid
    v[:20] = rng.uniform(-50, 530, 20).astype(np.float32)
    level = np.clip(kp2["octave"][src] + rng.integers(-1, 2, M), 0, 7).astype(np.int32)
    valid = (rng.random(M) < 0.9).astype(np.uint8)
    m = uvo.ORBmatcher(0.6, True)
    for th in (3.0, 12.0):
        bi, bd = m.FuseSearch(kp2, de2, (0, 0, 752, 480), u, v, level, valid, mp_desc, sf, th)
        oi, od = oracle.fuse_search(kp2, de2, (0, 0, 752, 480), u, v, level, valid, mp_desc, sf, th)
        np.testing.assert_array_equal(bi, oi)
        np.testing.assert_array_equal(bd, od)
    assert (oi >= 0).sum() > 500
    m.close()


def test_search_for_triangulation_batch_equals_twenty_single_calls(uvo, oracle, synth):
    """CreateNewMapPoints' loop (src/LocalMapping.cc:1058-1180): SearchForTriangulation(current KF, neighbour k) for 20 neighbours, and
    between two calls some of the matched features of the current key frame get a map point (the triangulation accepted them).  The
    batched form -- one launch for all pairs, then the acceptance loop of :886-984 replayed per pair on the host -- must give, pair by
    pair, exactly what the single calls made in the same order give (and what the oracle's sequential restatement gives), with the
    has_mp1 that evolves in between."""
    rng = np.random.default_rng(2301)
    W, H = 752, 480
    base = synth.make_frame(7100, W, H)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 7, max_width=W, max_height=H)
    kp1, de1 = ex(base)
    sigma2 = (ex.mvScaleFactor * ex.mvScaleFactor).astype(np.float32)
    neigh = []
    for k in range(20):
        kp2, de2 = ex(synth.warp_frame(base, 7200 + k))
        if k == 7:
            kp2, de2 = kp2[:0], de2[:0]                      # a neighbour without key points
        g2 = _bow_groups(rng, de2, 40) if len(de2) else {}
        has2 = (rng.random(len(kp2)) < 0.3).astype(np.uint8)
        F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 3e-4, (3, 3)).astype(np.float32))
        s2 = (sigma2 * np.float32(rng.choice([1.0, 40.0, 400.0]))).astype(np.float32)
        neigh.append((g2, kp2, de2, has2, F12, s2))
    ex.close()
    g1 = _bow_groups(rng, de1, 40)
    fv1 = uvo.FeatureVector(g1)
    has1_0 = (rng.random(len(kp1)) < 0.25).astype(np.uint8)
    for ori in (False, True):
        m = uvo.ORBmatcher(0.6, ori)
        mb = uvo.ORBmatcher(0.6, ori)
        fv2s = [uvo.FeatureVector(g) for g, *_ in neigh]
        mb.SearchForTriangulationBatch(fv1, kp1, de1, has1_0, [(fv2s[k],) + neigh[k][1:] for k in range(20)])
        has1 = has1_0.copy()
        accept = np.random.default_rng(99)
        total = 0
        for k, (g2, kp2, de2, has2, F12, s2) in enumerate(neigh):
            mo, no = oracle.search_for_triangulation(g1, kp1, de1, has1, g2, kp2, de2, has2, F12, s2, ori) if len(kp2) else (np.full(len(kp1), -1, np.int32), 0)
            ms, ns = m.SearchForTriangulation(fv1, kp1, de1, has1, fv2s[k], kp2, de2, has2, F12, s2)
            mbk, nbk = mb.SearchForTriangulationNext(k, has1)
            np.testing.assert_array_equal(ms, mo, err_msg="single call, pair %d" % k)
            np.testing.assert_array_equal(mbk, mo, err_msg="batched form, pair %d" % k)
            assert ns == no and nbk == no
            total += no
            # the triangulation accepts about two thirds of the pair's matches: those features now hold a map point
            won = np.nonzero(mo >= 0)[0]
            has1[won[accept.random(len(won)) < 0.66]] = 1
        assert total > 150
        # replaying an earlier pair with the has_mp1 of that time gives that time's result again; a feature that LOST its point is refused
        again, _ = mb.SearchForTriangulationNext(0, has1_0)
        ref0, _ = oracle.search_for_triangulation(g1, kp1, de1, has1_0, neigh[0][0], neigh[0][1], neigh[0][2], neigh[0][3], neigh[0][4], neigh[0][5], ori)
        np.testing.assert_array_equal(again, ref0)
        lost = has1_0.copy()
        lost[np.nonzero(has1_0)[0][0]] = 0
        with pytest.raises(uvo.UvoError):
            mb.SearchForTriangulationNext(0, lost)
        m.close()
        mb.close()


def test_fuse_batch_equals_the_single_calls(uvo, oracle, synth):
    """SearchInNeighbors' loop (src/LocalMapping.cc:1228-1236): Fuse(target k, the current key frame's map points) for 20 targets with
    their own poses -- uvo_fuse_batch (projection tests + search core for every (target, point) in one pass) against
    uvo_project_points + uvo_fuse per target and against the oracle."""
    rng = np.random.default_rng(2401)
    W, H = 752, 480
    base = synth.make_frame(7400, W, H)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 7, max_width=W, max_height=H)
    sf = ex.mvScaleFactor.copy()
    M = 2500
    xyz = (rng.normal(0, 1, (M, 3)) * [3, 2, 1.5] + [0, 0, 6]).astype(np.float32)
    targets, per_target = [], []
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    for t in range(20):
        kp, de = ex(synth.warp_frame(base, 7500 + t))
        if t == 5:
            kp, de = kp[:0], de[:0]
        R, tv, Ow = _random_pose(rng)
        cam = uvo.CameraPose.make(R, tv, Ow, fx, fy, cx, cy, (0.0, 0.0, float(W), float(H)))
        cam_o = np.concatenate([R.reshape(9), tv, Ow, np.float32([fx, fy, cx, cy]), np.float32([0, W, 0, H])]).astype(np.float32)
        targets.append((kp, de, cam, sf))
        per_target.append(cam_o)
    ex.close()
    # map points: most are key points of some "home" target back-projected to a depth (so they project onto that key point there, with a
    # descriptor close to its own, a normal facing the home camera and a distance range that predicts the key point's level); the rest random
    nrm = rng.normal(0, 1, (M, 3))
    mn = np.full(M, 1.0, np.float32)
    mp_desc = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    real = [t for t in range(20) if len(targets[t][0])]
    for i in range(M):
        if rng.random() < 0.15:
            continue
        t = real[rng.integers(len(real))]
        kp, de, cam, _ = targets[t]
        j = rng.integers(len(kp))
        R, tv, Ow = np.array(cam.rcw, np.float64).reshape(3, 3), np.array(cam.tcw, np.float64), np.array(cam.ow, np.float64)
        z = rng.uniform(3, 9)
        xc = np.array([(kp["x"][j] + rng.normal(0, 0.7) - cx) / fx * z, (kp["y"][j] + rng.normal(0, 0.7) - cy) / fy * z, z])
        xw = R.T @ (xc - tv)
        xyz[i] = xw
        nrm[i] = (xw - Ow) + rng.normal(0, 0.2, 3)
        dist = np.linalg.norm(xw - Ow)
        mn[i] = dist / (0.8 * sf[int(kp["octave"][j])]) * rng.uniform(0.9, 1.15)
        mp_desc[i] = np.packbits(np.unpackbits(de[j]) ^ (rng.random(256) < 0.04))
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    mx = (mn * rng.uniform(3.0, 8.0, M)).astype(np.float32)
    usable = (rng.random(M) < 0.9).astype(np.uint8)
    m = uvo.ORBmatcher(0.6, True, max_map_points=4096)
    for th in (3.0, 12.0):
        bi, bd = m.FuseBatch(targets, xyz, nrm, mn, mx, usable, mp_desc, th)
        hits = 0
        for t, (kp, de, cam, _) in enumerate(targets):
            valid, u, v, level, _ = m.project_points(uvo.PROJECT_FUSE, cam, xyz, nrm, mn, mx, usable, sf)
            ov, ou, ovv, ol, _ = oracle.project_points(uvo.PROJECT_FUSE, per_target[t], xyz, nrm, mn, mx, usable, sf, 1.2, 0.5)
            np.testing.assert_array_equal(valid, ov)
            if len(kp):
                si, sd = m.FuseSearch(kp, de, (0, 0, W, H), u, v, level, valid, mp_desc, sf, th)
                oi, od = oracle.fuse_search(kp, de, (0, 0, W, H), ou, ovv, ol, ov, mp_desc, sf, th)
            else:
                si = sd = oi = od = np.full(M, -1, np.int32)
            np.testing.assert_array_equal(si, oi, err_msg="single call, target %d" % t)
            np.testing.assert_array_equal(bi[t], oi, err_msg="batched form, target %d th %g" % (t, th))
            np.testing.assert_array_equal(bd[t], od, err_msg="batched form (distances), target %d" % t)
            hits += int((oi >= 0).sum())
        assert hits > 1500, hits
    m.close()


def test_window_search_on_a_frame_larger_than_the_lds_grid_build(uvo, oracle):
    """More key points than k_grid_build keeps in LDS (4096): the global-memory phases of the same kernel."""
    rng = np.random.default_rng(77)
    n, M = 6000, 3000
    kp = np.zeros(n, uvo.KEYPOINT_DTYPE)
    kp["x"], kp["y"], kp["octave"] = rng.uniform(-3, 755, n), rng.uniform(-3, 483, n), rng.integers(0, 8, n)
    de = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    src = rng.integers(0, n, M)
    mpd = _noisy_copies(rng, de, src, 0.05)
    u = (kp["x"][src] + rng.normal(0, 2, M)).astype(np.float32)
    v = (kp["y"][src] + rng.normal(0, 2, M)).astype(np.float32)
    lvl = np.clip(kp["octave"][src] + rng.integers(0, 2, M), 0, 7).astype(np.int32)
    valid = (rng.random(M) < 0.9).astype(np.uint8)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    m = uvo.ORBmatcher(0.8, max_query=8192, max_map_points=8192)
    bi, bd = m.FuseSearch(kp, de, (0, 0, 752, 480), u, v, lvl, valid, mpd, sf, 6.0)
    oi, od = oracle.fuse_search(kp, de, (0, 0, 752, 480), u, v, lvl, valid, mpd, sf, 6.0)
    np.testing.assert_array_equal(bi, oi)
    np.testing.assert_array_equal(bd, od)
    a, b = np.full(n, -1, np.int32), np.full(n, -1, np.int32)
    vc = np.full(M, 0.9, np.float32)
    na = m.SearchByProjection(kp, de, (0, 0, 752, 480), a, u, v, lvl, vc, valid, mpd, sf, 2.0)
    nb = oracle.search_by_projection(kp, de, (0, 0, 752, 480), b, u, v, lvl, vc, valid, mpd, sf, 2.0, 0.8)
    assert na == nb and na > 500
    np.testing.assert_array_equal(a, b)
    m.close()


def _random_pose(rng):
    a = rng.normal(0, 0.15, 3)
    th = np.linalg.norm(a)
    k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32)
    t = rng.normal(0, 0.5, 3).astype(np.float32)
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    return R, t, Ow


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_project_points(uvo, oracle, mode):
    """uvo_project_points against the restated isInFrustum / PredictScale, SearchByProjection(F, pKF) and Fuse prologues."""
    rng = np.random.default_rng(30 + mode)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    m = uvo.ORBmatcher(0.8)
    hits = 0
    for trial in range(4):
        R, t, Ow = _random_pose(rng)
        bounds = (0.0, 0.0, 752.0, 480.0)
        cam = uvo.CameraPose.make(R, t, Ow, 458.654, 457.296, 367.215, 248.375, bounds)
        cam_o = np.concatenate([R.reshape(9), t, Ow, np.float32([458.654, 457.296, 367.215, 248.375]), np.float32([0, 752, 0, 480])]).astype(np.float32)
        n = 20000
        xyz = (rng.normal(0, 1, (n, 3)) * [4, 3, 4] + [0, 0, 6]).astype(np.float32)
        nrm = rng.normal(0, 1, (n, 3))
        nrm[: n // 2] = (xyz[: n // 2] - Ow) + rng.normal(0, 1.0, (n // 2, 3))     # half the normals roughly face the camera
        nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
        d = np.linalg.norm(xyz - Ow, axis=1)
        mn = (d * rng.uniform(0.3, 1.4, n)).astype(np.float32)
        mx = (mn * rng.uniform(1.5, 6.0, n)).astype(np.float32)
        usable = (rng.random(n) < 0.9).astype(np.uint8)
        got = m.project_points(mode, cam, xyz, nrm, mn, mx, usable, sf, 1.2, 0.5)
        ref = oracle.project_points(mode, cam_o, xyz, nrm, mn, mx, usable, sf, 1.2, 0.5)
        for g, r, name in zip(got, ref, ("valid", "u", "v", "level", "view_cos")):
            assert g.dtype == r.dtype
            np.testing.assert_array_equal(g.view(np.uint32) if g.dtype == np.float32 else g, r.view(np.uint32) if r.dtype == np.float32 else r,
                                          err_msg="mode %d %s" % (mode, name))
        hits += int(ref[0].sum())
        assert len(set(ref[3][ref[0] > 0].tolist())) >= 5          # several levels exercised
    assert hits > 4000
    m.close()


def test_async_host_form_matches_the_synchronous_one(uvo, synth):
    """uvo_extract_batch_submit / _wait with two batches in flight (pipeline depth 2, page-locked buffers) against uvo_extract_batch."""
    W, H, B = 320, 256, 6
    frames = [np.stack([synth.make_frame(9100 + 10 * k + i, W, H) for i in range(B)]) for k in range(3)]
    ex = uvo.ORBextractor(500, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=B)
    ref = [ex.extract_batch(f) for f in frames]
    ex.set_pipeline(2)
    cap = ex.cap
    bufs = []
    for k in range(3):
        img = uvo.pinned_empty((B, H, W), np.uint8)
        img[:] = frames[k]
        bufs.append((img, uvo.pinned_empty((B, cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((B, cap, 32), np.uint8), uvo.pinned_empty((B,), np.int32)))
    t0 = ex.submit(*bufs[0])
    t1 = ex.submit(*bufs[1])
    assert t0 != t1
    with pytest.raises(uvo.UvoError):
        ex.submit(*bufs[2])                      # both lanes busy
    ex.wait(t0)
    t2 = ex.submit(*bufs[2])
    ex.wait(t1)
    ex.wait(t2)
    for k in range(3):
        _, kp, de, n = bufs[k]
        for b in range(B):
            rk, rd = ref[k][b]
            assert n[b] == len(rk)
            assert kp[b, :n[b]].tobytes() == rk.tobytes()
            np.testing.assert_array_equal(de[b, :n[b]], rd)
    ex.close()


def _bits(a):
    a = np.asarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_sim3_decompose_and_relative(uvo, oracle):
    """uvo_sim3_decompose / uvo_sim3_relative (host arithmetic behind the C ABI) against the restated cv::Mat expressions."""
    rng = np.random.default_rng(50)
    for trial in range(300):
        R, t, _ = _random_pose(rng)
        s = np.float32(rng.uniform(0.2, 5.0))
        Scw = np.eye(4, dtype=np.float32)
        Scw[:3, :3], Scw[:3, 3] = s * R, s * t
        cam = uvo.CameraPose.make(np.zeros(9), np.zeros(3), np.zeros(3), 1, 1, 0, 0, (0, 0, 1, 1))
        uvo.ORBmatcher.sim3_decompose(Scw, cam)
        r, tt, o = oracle.sim3_decompose(Scw)
        np.testing.assert_array_equal(_bits(np.float32(cam.rcw[:])), _bits(r))
        np.testing.assert_array_equal(_bits(np.float32(cam.tcw[:])), _bits(tt))
        np.testing.assert_array_equal(_bits(np.float32(cam.ow[:])), _bits(o))
        got = uvo.ORBmatcher.sim3_relative(s, R, t)
        ref = oracle.sim3_relative(s, R, t)
        for g, e in zip(got, ref):
            np.testing.assert_array_equal(_bits(g), _bits(e))


def test_project_sim3(uvo, oracle):
    """Per-point prologue of SearchBySim3 (:1323-1359): world -> own camera -> other camera -> pixel, distance window, level."""
    rng = np.random.default_rng(51)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    m = uvo.ORBmatcher(0.8)
    hits = 0
    for trial in range(4):
        R1, t1, _ = _random_pose(rng)
        R12, t12, _ = _random_pose(rng)
        s12 = np.float32(rng.uniform(0.5, 2.0))
        sR12, sR21, t21 = uvo.ORBmatcher.sim3_relative(s12, R12, t12)
        bounds = (0.0, 0.0, 752.0, 480.0)
        cam = uvo.CameraPose.make(np.eye(3), np.zeros(3), np.zeros(3), 458.654, 457.296, 367.215, 248.375, bounds)
        n = 20000
        xyz = (rng.normal(0, 1, (n, 3)) * [4, 3, 4] + [0, 0, 6]).astype(np.float32)
        d = np.linalg.norm(xyz, axis=1) / s12
        mn = (d * rng.uniform(0.3, 1.4, n)).astype(np.float32)
        mx = (mn * rng.uniform(1.5, 6.0, n)).astype(np.float32)
        usable = (rng.random(n) < 0.9).astype(np.uint8)
        for sr, tt in ((sR21, t21), (sR12, t12)):
            got = m.project_sim3(R1, t1, sr, tt, cam, xyz, mn, mx, usable, sf)
            ref = oracle.project_sim3(R1, t1, sr, tt, cam.as_array(), xyz, mn, mx, usable, sf)
            for g, r, name in zip(got, ref, ("valid", "u", "v", "level")):
                assert g.dtype == r.dtype
                np.testing.assert_array_equal(_bits(g), _bits(r), err_msg=name)
            hits += int(ref[0].sum())
            assert len(set(ref[3][ref[0] > 0].tolist())) >= 4
    assert hits > 4000
    m.close()


def test_search_by_projection_sim3_and_search_by_sim3(uvo, oracle, synth):
    """SearchByProjection(pKF, Scw, ...) search core (:357-398) and SearchBySim3 (:1361-1504) on two real views."""
    rng = np.random.default_rng(52)
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 4500)
    n1, n2 = len(kp1), len(kp2)
    bounds = (0, 0, 752, 480)
    m = uvo.ORBmatcher(0.75, True)
    # --- exclusive projection search into key frame 2 ---
    M = 3000
    src = rng.integers(0, n2, M)
    mp_desc = _noisy_copies(rng, de2, src, 0.04)
    u = (kp2["x"][src] + rng.normal(0, 1.5, M)).astype(np.float32)
    v = (kp2["y"][src] + rng.normal(0, 1.5, M)).astype(np.float32)
    level = np.clip(kp2["octave"][src] + rng.integers(0, 2, M), 0, 7).astype(np.int32)
    valid = (rng.random(M) < 0.9).astype(np.uint8)
    for th in (10, 3):
        mg = np.where(rng.random(n2) < 0.2, 5000 + np.arange(n2), -1).astype(np.int32)      # some key points already hold a point
        mo = mg.copy()
        ng = m.SearchByProjectionSim3(kp2, de2, bounds, mg, u, v, level, valid, mp_desc, sf, th)
        no = oracle.search_by_projection_sim3(kp2, de2, bounds, mo, u, v, level, valid, mp_desc, sf, th)
        assert ng == no
        np.testing.assert_array_equal(mg, mo)
    assert no > 200
    # --- SearchBySim3: every key point of either frame owns a map point; its projection into the other frame = the position of a
    #     true correspondent (nearest key point of the other frame after the known warp is not available here, so positions of
    #     descriptor-nearest neighbours stand in) plus noise ---
    i0, d0, _, _ = oracle.knn2(de1, de2)
    j0, e0, _, _ = oracle.knn2(de2, de1)

    def projection(kp_to, nn, n):
        uu = (kp_to["x"][nn] + rng.normal(0, 2.0, n)).astype(np.float32)
        vv = (kp_to["y"][nn] + rng.normal(0, 2.0, n)).astype(np.float32)
        lv = np.clip(kp_to["octave"][nn] + rng.integers(0, 2, n), 0, 7).astype(np.int32)
        va = (rng.random(n) < 0.85).astype(np.uint8)
        return va, uu, vv, lv
    p12, p21 = projection(kp2, i0, n1), projection(kp1, j0, n2)
    md1 = _noisy_copies(rng, de1, np.arange(n1), 0.03, 0.9)
    md2 = _noisy_copies(rng, de2, np.arange(n2), 0.03, 0.9)
    tot = 0
    for th in (7.5, 20.0):
        g12, gf = m.SearchBySim3(kp1, de1, bounds, kp2, de2, bounds, p12, md1, p21, md2, sf, sf, th)
        o12, of = oracle.search_by_sim3(kp1, de1, bounds, kp2, de2, bounds, p12, md1, p21, md2, sf, sf, th)
        assert gf == of
        np.testing.assert_array_equal(g12, o12)
        tot += of
    assert tot > 100
    m.close()


def test_frustum_then_search_by_projection_chain(uvo, oracle, synth):
    """Config-5 shape: isInFrustum on the device feeds SearchByProjection on the device; same chain through the oracle."""
    rng = np.random.default_rng(40)
    img = synth.make_frame(777, 752, 480)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 7, max_width=752, max_height=480)
    kp, de = ex(img)
    sf = ex.mvScaleFactor.copy()
    ex.close()
    n, M = len(kp), 5000
    R, t, Ow = _random_pose(rng)
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    # map points: back-project keypoints at random depths (+ noise), so that they land near their keypoint again
    src = rng.integers(0, n, M)
    z = rng.uniform(2, 12, M)
    pc = np.stack([(kp["x"][src] - cx) / fx * z, (kp["y"][src] - cy) / fy * z, z], 1) + rng.normal(0, 0.01, (M, 3))
    xyz = ((pc - t) @ R.astype(np.float64)).astype(np.float32)                      # Xw = R^T (Xc - t)
    nrm = xyz - Ow
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    d = np.linalg.norm(xyz - Ow, axis=1)
    lvl_true = kp["octave"][src]
    mxd = (d * sf[lvl_true] * rng.uniform(0.95, 1.05, M)).astype(np.float32)        # PredictScale then lands near the keypoint's octave
    mnd = (mxd / sf[7]).astype(np.float32)
    mp_desc = _noisy_copies(rng, de, src, 0.06, 0.6)
    bounds = (0, 0, 752, 480)
    cam = uvo.CameraPose.make(R, t, Ow, fx, fy, cx, cy, bounds)
    cam_o = np.concatenate([R.reshape(9), t, Ow, np.float32([fx, fy, cx, cy]), np.float32([0, 752, 0, 480])]).astype(np.float32)
    m = uvo.ORBmatcher(0.8, max_query=4096, max_map_points=8192)
    valid, u, v, level, vc = m.project_points(uvo.PROJECT_FRUSTUM, cam, xyz, nrm, mnd, mxd, None, sf, 1.2, 0.5)
    a_g = np.full(n, -1, np.int32)
    nm_g = m.SearchByProjection(kp, de, bounds, a_g, u, v, level, vc, valid, mp_desc, sf, 1.0)
    ov, ou, ovv, ol, ovc = oracle.project_points(0, cam_o, xyz, nrm, mnd, mxd, None, sf, 1.2, 0.5)
    a_o = np.full(n, -1, np.int32)
    nm_o = oracle.search_by_projection(kp, de, bounds, a_o, ou, ovv, ol, ovc, ov, mp_desc, sf, 1.0, 0.8)
    np.testing.assert_array_equal(a_g, a_o)
    assert nm_g == nm_o and nm_g > 300 and ov.sum() > 3000
    # the same as one call (uvo_search_points_in_frustum): projections stay on the device
    for th in (1.0, 5.0):
        usable = None if th == 1.0 else (rng.random(M) < 0.8).astype(np.uint8)
        a_f = np.where(rng.random(n) < 0.1, 9000, -1).astype(np.int32) if th == 5.0 else np.full(n, -1, np.int32)
        a_r = a_f.copy()
        nm_f, iv, fu, fv, fl, fc = m.SearchPointsInFrustum(kp, de, a_f, cam, xyz, nrm, mnd, mxd, usable, mp_desc, sf, 1.2, 0.5, th, want_projections=True)
        ov, ou, ovv, ol, ovc = oracle.project_points(0, cam_o, xyz, nrm, mnd, mxd, usable, sf, 1.2, 0.5)
        nm_r = oracle.search_by_projection(kp, de, bounds, a_r, ou, ovv, ol, ovc, ov, mp_desc, sf, th, 0.8)
        np.testing.assert_array_equal(iv, ov)
        for g, r in ((fu, ou), (fv, ovv), (fc, ovc)):
            np.testing.assert_array_equal(g.view(np.uint32), r.view(np.uint32))
        np.testing.assert_array_equal(fl, ol)
        np.testing.assert_array_equal(a_f, a_r)
        assert nm_f == nm_r
    # a matcher that has never sized its candidate buffer, dense windows (th = 40): the truncated first run must be repeated
    m2 = uvo.ORBmatcher(0.8, max_query=4096, max_map_points=8192)
    a_f = np.full(n, -1, np.int32)
    a_r = a_f.copy()
    nm_f, iv = m2.SearchPointsInFrustum(kp, de, a_f, cam, xyz, nrm, mnd, mxd, None, mp_desc, sf, 1.2, 0.5, 40.0)
    ov, ou, ovv, ol, ovc = oracle.project_points(0, cam_o, xyz, nrm, mnd, mxd, None, sf, 1.2, 0.5)
    nm_r = oracle.search_by_projection(kp, de, bounds, a_r, ou, ovv, ol, ovc, ov, mp_desc, sf, 40.0, 0.8)
    np.testing.assert_array_equal(a_f, a_r)
    assert nm_f == nm_r
    m2.close()
    m.close()


def _random_vocabulary(rng, k=10, L=4, weighting=0, normalize=1):
    """A k-ary tree of depth <= L whose node descriptors are noisy copies of their parent's (so the descent is meaningful);
    ~5 % of the inner candidates stop early as leaves, ~5 % of the words are stop words (weight 0)."""
    desc, children, level = [np.zeros(32, np.uint8)], [[]], [0]
    frontier = [0]
    for lvl in range(1, L + 1):
        nxt = []
        for p in frontier:
            base = rng.integers(0, 256, 32, dtype=np.uint8) if p == 0 else desc[p]
            for _ in range(k):
                flips = rng.random(256) < (0.25 if lvl == 1 else 0.08)
                d = np.packbits(np.unpackbits(base) ^ flips)
                if lvl > 1 and rng.random() < 0.02:
                    d = desc[children[p][-1]].copy() if children[p] else d        # duplicate sibling: a tie the first child must win
                desc.append(d), children.append([]), level.append(lvl)
                children[p].append(len(desc) - 1)
                if lvl < L and rng.random() > 0.05:
                    nxt.append(len(desc) - 1)
        frontier = nxt
    n = len(desc)
    child_start = np.zeros(n + 1, np.int32)
    flat = []
    for i in range(n):
        flat.extend(children[i])
        child_start[i + 1] = len(flat)
    word_id = np.full(n, -1, np.int32)
    leaves = [i for i in range(n) if not children[i]]
    word_id[leaves] = np.arange(len(leaves))
    weight = np.zeros(n)
    weight[leaves] = np.where(rng.random(len(leaves)) < 0.05, 0.0, rng.uniform(0.5, 9.0, len(leaves)))
    return dict(child_start=child_start, children=np.asarray(flat, np.int32), descriptor=np.stack(desc), word_id=word_id, weight=weight, L=L,
                weighting=weighting, normalize=normalize)


@pytest.mark.parametrize("weighting,normalize", [(0, 1), (0, 2), (1, 0), (2, 1), (3, 0)])
def test_bow_transform(uvo, oracle, synth, weighting, normalize):
    """DBoW2 TemplatedVocabulary::transform on the device against its restatement; then both feature vectors through SearchByBoW."""
    rng = np.random.default_rng(50 + weighting)
    voc = _random_vocabulary(rng, 10, 4, weighting, normalize)
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 4500)
    # descriptors that resemble vocabulary nodes (so that the descent is not a coin toss) mixed with the real ones
    leaf_desc = voc["descriptor"][voc["word_id"] >= 0]
    near = np.packbits(np.unpackbits(leaf_desc[rng.integers(0, len(leaf_desc), 500)], axis=1) ^ (rng.random((500, 256)) < 0.05), axis=1)
    V = uvo.ORBVocabulary(voc["child_start"], voc["children"], voc["descriptor"], voc["word_id"], voc["weight"], voc["L"], weighting, normalize)
    for feats in (np.concatenate([de1, near]), de2[:1], de2[:0]):
        for levelsup in (0, 2, 4, 6):
            g = V.transform(feats, levelsup)
            o = oracle.bow_transform(voc, feats, levelsup)
            np.testing.assert_array_equal(g[0], o[0])
            np.testing.assert_array_equal(g[2], o[2])
            np.testing.assert_array_equal(g[1].view(np.uint64), o[1].view(np.uint64))
            np.testing.assert_array_equal(g[3][0], o[3][0])
            np.testing.assert_array_equal(g[3][1].view(np.uint64), o[3][1].view(np.uint64))     # BowVector values bit for bit
            fvg = {int(g[4].node[j]): [int(x) for x in g[4].feat[g[4].start[j]:g[4].start[j + 1]]] for j in range(len(g[4].node))}
            assert fvg == o[4]
    # the chain the reference runs: ComputeBoW on both sides, then SearchByBoW
    f1, f2 = V.transform(de1, 2), V.transform(de2, 2)
    o1, o2 = oracle.bow_transform(voc, de1, 2), oracle.bow_transform(voc, de2, 2)
    m = uvo.ORBmatcher(0.8, True)
    usable1 = np.ones(len(de1), np.uint8)
    mg, ng = m.SearchByBoW(f1[4], de1, kp1["angle"], usable1, f2[4], de2, kp2["angle"])
    mo, no = oracle.search_by_bow(False, o1[4], de1, kp1["angle"], usable1, o2[4], de2, kp2["angle"], None, 0.8, True)
    np.testing.assert_array_equal(mg, mo)
    assert ng == no
    m.close()
    V.close()


def test_clahe(uvo, oracle, synth):
    """cv::CLAHE::apply (src/Tracking.cc:425-431) on the device: host entry, in-place HBM-resident batch, then extraction."""
    import torch
    rng = np.random.default_rng(60)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=768, max_height=512, max_batch=3)
    for (w, h), tiles, clip in (((640, 512), (12, 12), 4.0), ((752, 480), (8, 8), 4.0), ((637, 509), (12, 12), 2.0), ((320, 256), (4, 7), 0.0),
                                ((768, 512), (16, 16), 40.0)):
        img = synth.make_frame(900 + w, w, h)
        img = (img.astype(np.float32) * rng.uniform(0.2, 0.5) + 20).astype(np.uint8)          # dim, low contrast: the underwater case
        np.testing.assert_array_equal(ex.clahe(img, clip, tiles), oracle.clahe(img, clip, tiles), err_msg=str((w, h, tiles, clip)))
    flat = np.full((512, 640), 10, np.uint8)
    np.testing.assert_array_equal(ex.clahe(flat), oracle.clahe(flat))
    # enhance -> extract without leaving the device, in place, 3 frames
    B, W, H = 3, 640, 512
    frames = np.stack([(synth.make_frame(1200 + i, W, H).astype(np.float32) * 0.3 + 30).astype(np.uint8) for i in range(B)])
    d = torch.from_numpy(frames).cuda()
    ex.clahe_batch_device(d.data_ptr(), B, W, H, d.data_ptr())
    cap = ex.cap
    kp = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    de = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d.data_ptr(), B, W, H, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap)
    ex.synchronize()
    enhanced = d.cpu().numpy()
    oe = oracle.extractor(1000, 1.2, 8, 20)
    for b in range(B):
        ref = oracle.clahe(frames[b])
        np.testing.assert_array_equal(enhanced[b], ref)
        kp_o, de_o = oe(ref)
        nb = int(n[b])
        assert nb == len(kp_o)
        np.testing.assert_array_equal(de[b, :nb].cpu().numpy(), de_o)
    ex.close()


def test_hbm_resident_chain_clahe_pyramid_extract(uvo, synth):
    """uvo_clahe keeps its result in HBM: extraction with img = NULL and the KLT pyramid built from the extractor's handle must
    equal the same calls fed with the downloaded image."""
    W, H = 640, 512
    raw = (synth.make_frame(515, W, H).astype(np.float32) * 0.45 + 25).astype(np.uint8)
    ex = uvo.ORBextractor(800, 1.2, 8, 0, 20, max_width=W, max_height=H)
    klt = uvo.KLT(W, H, (21, 21), 5, max_points=64, slots=2)
    with pytest.raises(uvo.UvoError):
        klt.build_pyramid_from(0, ex)                     # no clahe() result yet
    enh = ex.clahe(raw, 4.0, (12, 12))
    kp_a, de_a = ex(enh)
    n_a = klt.build_pyramid(0, enh)
    assert ex.clahe(raw, 4.0, (12, 12), download=False) is None
    kp_b, de_b = ex(None)
    n_b = klt.build_pyramid_from(1, ex)
    assert kp_a.tobytes() == kp_b.tobytes() and np.array_equal(de_a, de_b) and n_a == n_b
    for lvl in range(n_a):
        ia, da = klt.read_level(0, lvl)
        ib, db = klt.read_level(1, lvl)
        np.testing.assert_array_equal(ia, ib)
        np.testing.assert_array_equal(da, db)
    ex.close()
    ex2 = uvo.ORBextractor(800, 1.2, 8, 0, 20, max_width=W, max_height=H)
    ex2._clahe_shape = (H, W)
    with pytest.raises(uvo.UvoError):
        ex2(None)                                         # img = NULL without a preceding clahe()
    ex2.close()
    klt.close()


def test_haloc_hash(uvo, oracle, synth):
    """haloc::Hash::getHash bit for bit (the accumulation order is part of the result)."""
    rng = np.random.default_rng(70)
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 4600)
    m = uvo.ORBmatcher(0.8)
    for de in (de1, de2[:1], de2[:0], rng.integers(0, 256, (3000, 32), dtype=np.uint8)):
        proj = rng.normal(0, 1, (3, 6000)).astype(np.float32)
        proj /= np.linalg.norm(proj, axis=1, keepdims=True).astype(np.float32)
        np.testing.assert_array_equal(m.haloc_hash(proj, de).view(np.uint32), oracle.haloc_hash(proj, de).view(np.uint32))
    m.close()


def test_klt_pyramid_and_tracking(uvo, oracle, synth):
    """cv::buildOpticalFlowPyramid + cv::calcOpticalFlowPyrLK: pyramid levels and derivatives bit-exact.  The tracker's window sums are
    float accumulations, so their value depends on the order of the additions: the kernel must equal -- bit for bit, positions, status
    and error -- the oracle run in the kernel's own order (sum_mode 1: lane-strided partial sums + xor butterfly); how far that is from
    the raster order of OpenCV's generic loop (sum_mode 0) is a property of the algorithm, bounded here and examined on the CPU in
    tests/test_oracle_kat.py::test_klt_association_order_only_moves_decisions_at_their_thresholds."""
    rng = np.random.default_rng(80)
    for (w, h), win, ml in (((640, 512), (21, 21), 5), ((752, 480), (15, 15), 3), ((331, 257), (21, 21), 5)):
        a = synth.make_frame(6000 + w, w, h)
        b = synth.warp_frame(a, 6001 + w)
        k = uvo.KLT(w, h, win, ml, max_points=4096, slots=2)
        na, nb = k.build_pyramid(0, a), k.build_pyramid(1, b)
        pa, pb = oracle.klt_pyramid(a, win, ml), oracle.klt_pyramid(b, win, ml)
        assert na == nb == pa.levels
        for l in range(na):
            gi, gd = k.read_level(0, l)
            oi, od = pa.level(l)
            np.testing.assert_array_equal(gi, oi, err_msg="image level %d" % l)
            np.testing.assert_array_equal(gd, od, err_msg="derivative level %d" % l)
        n = 2000
        pts = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)   # some outside / at the border
        init = (pts + rng.normal(0, 1.0, (n, 2))).astype(np.float32)                                   # OPTFLOW_USE_INITIAL_FLOW
        g_next, g_st, g_err = k.track(0, 1, pts, init)
        o_next, o_st, o_err, _ = oracle.klt_track_ex(pa, pb, pts, init, win, ml, sum_mode=1)
        np.testing.assert_array_equal(g_st, o_st)
        np.testing.assert_array_equal(g_next.view(np.uint32), o_next.view(np.uint32))
        np.testing.assert_array_equal(g_err.view(np.uint32), o_err.view(np.uint32))
        assert (g_st > 0).sum() > 0.6 * n
        # against the raster order: same status wherever no decision sits at its threshold, positions to float rounding
        r_next, r_st, r_err, r_mg = oracle.klt_track_ex(pa, pb, pts, init, win, ml, sum_mode=0)
        assert ((g_st == r_st) | (r_mg < 1e-3)).all()
        both = (g_st > 0) & (r_st > 0)
        d = np.abs(g_next[both] - r_next[both]).max(axis=1)
        assert np.median(d) < 1e-3 and (d[r_mg[both] > 0.1] < 0.01).all()
        # second use of a pyramid in the other role (frame t becomes "previous"), slots swapped; default initial flow
        g2, s2, e2 = k.track(1, 0, pts)
        o2, so2, eo2, _ = oracle.klt_track_ex(pb, pa, pts, None, win, ml, sum_mode=1)
        np.testing.assert_array_equal(s2, so2)
        np.testing.assert_array_equal(g2.view(np.uint32), o2.view(np.uint32))
        k.close()


def test_new_entry_points_reject_bad_arguments(uvo):
    """Error behaviour of the search / BoW / KLT / CLAHE entry points: bad input -> UvoError with a message, never a crash."""
    m = uvo.ORBmatcher(0.8)
    kp = np.zeros(4, uvo.KEYPOINT_DTYPE)
    kp["x"], kp["y"] = [10, 20, 30, 40], [10, 20, 30, 40]
    de = np.zeros((4, 32), np.uint8)
    one = np.ones(1, np.float32)
    with pytest.raises(uvo.UvoError):                       # unknown rule
        m.match_windows(kp, de, (0, 0, 64, 48), one, one, one, [0], [0], [1], de[:1], 9, 50)
    with pytest.raises(uvo.UvoError):                       # empty bounds
        m.match_windows(kp, de, (0, 0, 0, 48), one, one, one, [0], [0], [1], de[:1], uvo.RULE_BEST_ONLY, 50)
    with pytest.raises(uvo.UvoError):                       # feature index outside the keypoint range
        m.SearchByBoW(uvo.FeatureVector({1: [0, 7]}), de, np.zeros(4), np.ones(4), uvo.FeatureVector({1: [0]}), de, np.zeros(4))
    with pytest.raises(uvo.UvoError):                       # map point level outside the scale table
        m.FuseSearch(kp, de, (0, 0, 64, 48), one, one, [9], [1], de[:1], np.ones(8, np.float32))
    # degenerate but legal: no queries, no targets
    mt, dist, n = m.match_windows(kp[:0], de[:0], (0, 0, 64, 48), one, one, one, [0], [0], [1], de[:1], uvo.RULE_BEST_ONLY, 50)
    assert n == 0 and mt.tolist() == [-1]
    mm, n = m.SearchByBoW(uvo.FeatureVector({}), de, np.zeros(4), np.ones(4), uvo.FeatureVector({1: [0]}), de, np.zeros(4))
    assert n == 0 and (mm == -1).all()
    m.close()
    with pytest.raises(uvo.UvoError):                       # child id pointing at the root
        uvo.ORBVocabulary([0, 1, 1], [0], np.zeros((2, 32), np.uint8), [-1, 0], [0.0, 1.0], 1)
    with pytest.raises(uvo.UvoError):                       # window larger than 1024 pixels
        uvo.KLT(640, 480, (40, 40), 3)
    k = uvo.KLT(640, 480, (21, 21), 3)
    with pytest.raises(uvo.UvoError):                       # tracking before any pyramid was built
        k.track(0, 1, np.zeros((1, 2), np.float32))
    with pytest.raises(uvo.UvoError):                       # image larger than the handle
        k.build_pyramid(0, np.zeros((500, 700), np.uint8))
    k.close()
    ex = uvo.ORBextractor(500, 1.2, 4, 0, 20, max_width=320, max_height=240)
    with pytest.raises(uvo.UvoError):                       # more tiles than pixels
        ex.clahe(np.zeros((240, 320), np.uint8), 4.0, (400, 12))
    # asynchronous host form: output rows shorter than a frame can need, waiting for nothing
    img = np.zeros((1, 240, 320), np.uint8)
    with pytest.raises(uvo.UvoError):
        ex.submit(img, np.zeros((1, 8), uvo.KEYPOINT_DTYPE), np.zeros((1, 8, 32), np.uint8), np.zeros(1, np.int32))
    with pytest.raises(uvo.UvoError):
        ex.wait(0)
    with pytest.raises(uvo.UvoError):
        ex.wait(17)
    ex.close()
    # loop-closing forms and the fused frustum search
    m = uvo.ORBmatcher(0.8, max_query=64, max_map_points=64)
    cam = uvo.CameraPose.make(np.eye(3), np.zeros(3), np.zeros(3), 500, 500, 320, 240, (0, 0, 640, 480))
    with pytest.raises(uvo.UvoError):                       # Scw with a zero rotation block
        uvo.ORBmatcher.sim3_decompose(np.zeros((4, 4), np.float32), cam)
    with pytest.raises(uvo.UvoError):                       # non-positive scale
        uvo.ORBmatcher.sim3_relative(0.0, np.eye(3), np.zeros(3))
    with pytest.raises(uvo.UvoError):                       # level outside the scale table
        m.SearchByProjectionSim3(kp, de, (0, 0, 64, 48), np.full(4, -1, np.int32), one, one, [9], [1], de[:1], np.ones(8, np.float32), 3)
    with pytest.raises(uvo.UvoError):                       # more map points than the handle was sized for
        m.SearchPointsInFrustum(kp, de, np.full(4, -1, np.int32), cam, np.zeros((100, 3)), np.zeros((100, 3)), np.ones(100), np.ones(100), None,
                                np.zeros((100, 32), np.uint8), np.ones(8, np.float32))
    nm, iv = m.SearchPointsInFrustum(kp[:0], de[:0], np.zeros(0, np.int32), cam, [[0, 0, 5.0]], [[0, 0, 1.0]], [2.0], [10.0], None,
                                     np.zeros((1, 32), np.uint8), (np.float32(1.2) ** np.arange(8)).astype(np.float32))
    assert nm == 0 and iv.tolist() == [1]                   # no keypoints: the projection still reports the point in view
    m.close()


def test_wide_image_with_few_features(uvo, oracle, synth):
    """A wide image makes DistributeOctTree start from several root nodes; their first, unconditional split can return more
    keypoints than quota + 3 on a level (4 * nIni nodes).  Found by tools/soak_parity.py."""
    for (w, h), nfeat, nlev, th in (((572, 164), 58, 6, 12), ((900, 200), 40, 4, 20), ((640, 130), 300, 5, 7)):
        img = synth.make_frame(1234 + w, w, h, n_shapes=max(20, w * h // 900))
        ex = uvo.ORBextractor(nfeat, 1.2, nlev, 0, th, max_width=w, max_height=h)
        oe = oracle.extractor(nfeat, 1.2, nlev, th)
        kp_g, de_g = ex(img)
        kp_o, de_o = oe(img)
        assert len(kp_o) > nfeat                                   # the case in point: more than nfeatures come back
        _assert_same_features(kp_g, de_g, kp_o, de_o, "%dx%d nfeat %d" % (w, h, nfeat))
        assert len(kp_g) <= ex.cap
        ex.close()


def test_randomised_extraction_soak(uvo, oracle, synth):
    """A short run of tools/soak_parity.py (random sizes, level counts, scale factors, thresholds, content)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_parity.py"), "40", "7"], capture_output=True, text=True)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_randomised_matcher_soak(uvo, oracle):
    """A short run of tools/soak_matcher.py (every search entry point on random sizes, ratios, thresholds, empty sets)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_matcher.py"), "25", "11"], capture_output=True, text=True)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---- the four ORBmatcher members without a caller in the reference ----
@pytest.mark.parametrize("check_ori", [False, True])
def test_window_search_and_search_for_initialization(uvo, oracle, synth, check_ori):
    """WindowSearch (src/ORBmatcher.cc:409-516) and SearchForInitialization (:598-713) on two consecutive views."""
    rng = np.random.default_rng(77 + check_ori)
    kp1, de1, kp2, de2, _ = _two_views(uvo, synth, 6100, fast_th=20)
    bounds = (0, 0, 752, 480)
    m = uvo.ORBmatcher(0.9, check_ori, max_query=4096, max_map_points=8192)
    for window, lo, hi, frac in [(15, -1, 0x7fffffff, 0.8), (40, 1, 3, 0.5), (8, -1, 2, 1.0), (100, -1, 0x7fffffff, 0.9)]:
        has = (rng.random(len(kp1)) < frac).astype(np.uint8)
        m21_g, n_g = m.WindowSearch(kp1, de1, has, kp2, de2, bounds, window, lo, hi)
        m21_o, n_o = oracle.window_search(kp1, de1, has, kp2, de2, bounds, window, lo, hi, 0.9, check_ori)
        np.testing.assert_array_equal(m21_g, m21_o)
        assert n_g == n_o
    assert n_o > 200
    # SearchForInitialization: level-0 keypoints only, targets change hands when a later query is strictly closer
    for window, jitter in [(10, 2.0), (30, 6.0), (100, 1.0)]:
        prev = np.stack([kp1["x"], kp1["y"]], 1).astype(np.float32) + rng.normal(0, jitter, (len(kp1), 2)).astype(np.float32)
        p_g, p_o = prev.copy(), prev.copy()
        m12_g, n_g = m.SearchForInitialization(kp1, de1, kp2, de2, bounds, p_g, window)
        m12_o, n_o = oracle.search_for_initialization(kp1, de1, kp2, de2, bounds, p_o, window, 0.9, check_ori)
        np.testing.assert_array_equal(m12_g, m12_o)
        np.testing.assert_array_equal(p_g, p_o)
        assert n_g == n_o
    assert n_o > 50
    # a crowd competing for few targets: every level-0 query looks at the same spot with a huge window
    sel = np.nonzero(kp1["octave"] == 0)[0][:300]
    prev = np.tile(np.float32([[376, 240]]), (len(kp1), 1))
    p_g, p_o = prev.copy(), prev.copy()
    few = slice(0, 40)
    m12_g, n_g = m.SearchForInitialization(kp1, de1, kp2[few], de2[few], bounds, p_g, 400)
    m12_o, n_o = oracle.search_for_initialization(kp1, de1, kp2[few], de2[few], bounds, p_o, 400, 0.9, check_ori)
    np.testing.assert_array_equal(m12_g, m12_o)
    assert n_g == n_o and len(sel) > 100
    m.close()


@pytest.mark.parametrize("check_ori", [False, True])
def test_projection_searches_between_frames(uvo, oracle, synth, check_ori):
    """SearchByProjection(F1, F2, windowSize) (:519-594) and SearchByProjection(CurrentFrame, LastFrame, th) (:1507-1620)."""
    rng = np.random.default_rng(91 + check_ori)
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 6200, fast_th=20)
    n1, n2 = len(kp1), len(kp2)
    R, t, Ow = _random_pose(rng)
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    bounds = (0, 0, 752, 480)
    cam = uvo.CameraPose.make(R, t, Ow, fx, fy, cx, cy, bounds)
    cam_o = np.concatenate([R.reshape(9), t, Ow, np.float32([fx, fy, cx, cy]), np.float32([0, 752, 0, 480])]).astype(np.float32)
    # F1's map points: the back-projection of its own keypoints into frame 2's camera (+ noise), some behind the camera, some far outside
    z = rng.uniform(2, 12, n1) * np.where(rng.random(n1) < 0.05, -1, 1)
    pc = np.stack([(kp1["x"] - cx) / fx * z, (kp1["y"] - cy) / fy * z, z], 1) + rng.normal(0, 0.02, (n1, 3))
    pc[rng.random(n1) < 0.05] *= [30, 1, 1]
    xyz = ((pc - t) @ R.astype(np.float64)).astype(np.float32)
    m = uvo.ORBmatcher(0.9, check_ori, max_query=4096, max_map_points=8192)
    for window in (10, 40):
        usable = (rng.random(n1) < 0.8).astype(np.uint8)
        a_g = np.where(rng.random(n2) < 0.15, 5000, -1).astype(np.int32)
        a_o = a_g.copy()
        n_g = m.SearchByProjectionFrames(kp1, de1, usable, xyz, cam, kp2, de2, a_g, window)
        n_o = oracle.search_by_projection_frames(kp1, de1, usable, xyz, cam_o, kp2, de2, a_o, window, 0.9)
        np.testing.assert_array_equal(a_g, a_o)
        assert n_g == n_o
    assert n_o > 200
    for th in (7.0, 15.0):
        usable = (rng.random(n1) < 0.8).astype(np.uint8)
        a_g = np.where(rng.random(n2) < 0.15, 5000, -1).astype(np.int32)
        a_o = a_g.copy()
        n_g = m.SearchByProjectionLast(kp2, de2, a_g, cam, usable, xyz, kp1["octave"], kp1["angle"], de1, sf, th)
        n_o = oracle.search_by_projection_last(cam_o, kp2, de2, a_o, usable, xyz, kp1["octave"], kp1["angle"], de1, sf, th, check_ori)
        np.testing.assert_array_equal(a_g, a_o)
        assert n_g == n_o
    assert n_o > 200
    m.close()


def test_undistort_points_and_the_fused_track_call(uvo, oracle, synth):
    """Tracking::undistort_point (src/Tracking.cc:1265-1283), pin-hole and fisheye, on the device; and uvo_klt_track_undistorted = the
    tracker + the undistortion of both point sets (:1046-1053) as one call."""
    rng = np.random.default_rng(12)
    w, h = 640, 512
    k = uvo.KLT(w, h, (21, 21), 5, max_points=4096, slots=2)
    models = [(458.654, 457.296, 367.215, 248.375, [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], False),      # Data/Settings_VIORB.yaml
              (458.654, 457.296, 367.215, 248.375, [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.01], False),
              (413.32595366596017, 413.70198739483686, 305.9507483284928, 259.4439948946375,                             # Settings_VI_Aqualoc_harbor.yaml
               [-0.06125568297136998, -0.003796743395135256, 0.027326634771204592, -0.030296403142887066], True),
              (300.0, 300.0, 320.0, 256.0, [], False), (300.0, 300.0, 320.0, 256.0, [0.1], True)]
    pts = np.stack([rng.uniform(-20, w + 20, 4000), rng.uniform(-20, h + 20, 4000)], 1).astype(np.float32)
    pts[:3] = [[367.215, 248.375], [305.9507483284928, 259.4439948946375], [0, 0]]
    for fx, fy, cx, cy, dist, fisheye in models:
        cam = uvo.CameraModel.make(fx, fy, cx, cy, dist, fisheye)
        got = k.undistort(cam, pts)
        ref = oracle.undistort_points(pts, np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), dist, fisheye)
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32), err_msg="model %s" % ((fx, dist, fisheye),))
    # fused with the tracker
    a = synth.make_frame(6640, w, h)
    b = synth.warp_frame(a, 6641)
    k.build_pyramid(0, a), k.build_pyramid(1, b)
    p0 = np.stack([rng.uniform(20, w - 20, 1500), rng.uniform(20, h - 20, 1500)], 1).astype(np.float32)
    cam = uvo.CameraModel.make(*models[2][:4], models[2][4], True)
    nxt, st, er, pu, nu = k.track_undistorted(0, 1, p0, cam)
    n2, s2, e2 = k.track(0, 1, p0)
    np.testing.assert_array_equal(nxt.view(np.uint32), n2.view(np.uint32))
    np.testing.assert_array_equal(st, s2)
    np.testing.assert_array_equal(pu.view(np.uint32), oracle.undistort_points(p0, *[np.float32(v) for v in models[2][:4]], models[2][4], True).view(np.uint32))
    np.testing.assert_array_equal(nu.view(np.uint32), oracle.undistort_points(nxt, *[np.float32(v) for v in models[2][:4]], models[2][4], True).view(np.uint32))
    with pytest.raises(uvo.UvoError):
        k.undistort(uvo.CameraModel.make(0.0, 1.0, 0.0, 0.0, []), pts[:4])
    k.close()


def _orbvoc_shaped_vocabulary(rng, k=10, L=6):
    """The shape of the ORBvoc.txt the reference loads (k = 10, L = 6: 1 111 111 nodes, 10^6 words) with random content: a complete
    tree in level order, every child a noisy copy (p = 1/8 per bit) of its parent, tf-idf-like weights on the leaves."""
    sizes = [k ** l for l in range(L + 1)]
    first = np.concatenate([[0], np.cumsum(sizes)])
    n = int(first[-1])
    desc = np.zeros((n, 32), np.uint8)
    desc[0] = rng.integers(0, 256, 32, dtype=np.uint8)
    for l in range(1, L + 1):
        parents = np.repeat(desc[first[l - 1]:first[l]], k, axis=0)
        m = sizes[l]
        flips = rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8)
        desc[first[l]:first[l + 1]] = parents ^ flips
    child_start = np.zeros(n + 1, np.int32)
    inner = int(first[L])
    child_start[1:inner + 1] = np.arange(1, inner + 1, dtype=np.int64) * k
    child_start[inner + 1:] = inner * k
    children = np.arange(1, n, dtype=np.int32)                 # level order: the children of node i are 1 + i*k .. 1 + i*k + k - 1
    word_id = np.full(n, -1, np.int32)
    word_id[inner:] = np.arange(n - inner)
    weight = np.zeros(n)
    weight[inner:] = rng.uniform(0.5, 12.0, n - inner)
    return dict(child_start=child_start, children=children, descriptor=desc, word_id=word_id, weight=weight, L=L, weighting=0, normalize=1)


def test_bow_transform_and_search_on_a_vocabulary_of_orbvoc_shape(uvo, oracle, synth):
    """k_bow_descend and SearchByBoW at the memory footprint the reference runs with (Vocabulary/ORBvoc.txt: k = 10, L = 6, ~1.1 M nodes,
    35 MB of node descriptors), levelsup = 4 as at the call sites (src/FrameKTL.cc:439-446, src/KeyFrame.cc:203-210)."""
    rng = np.random.default_rng(2027)
    voc = _orbvoc_shaped_vocabulary(rng)
    assert len(voc["descriptor"]) == 1111111 and (voc["word_id"] >= 0).sum() == 10 ** 6
    kp1, de1, kp2, de2, sf = _two_views(uvo, synth, 4600)
    leaves = voc["descriptor"][-10 ** 6:]
    near = leaves[rng.integers(0, 10 ** 6, 600)] ^ (rng.integers(0, 256, (600, 32), dtype=np.uint8) & rng.integers(0, 256, (600, 32), dtype=np.uint8)
                                                    & rng.integers(0, 256, (600, 32), dtype=np.uint8) & rng.integers(0, 256, (600, 32), dtype=np.uint8))
    V = uvo.ORBVocabulary(voc["child_start"], voc["children"], voc["descriptor"], voc["word_id"], voc["weight"], voc["L"], 0, 1)
    for feats, levelsup in ((np.concatenate([de1, near]), 4), (de2, 4), (near, 2)):
        g = V.transform(feats, levelsup)
        o = oracle.bow_transform(voc, feats, levelsup)
        np.testing.assert_array_equal(g[0], o[0])                                        # word ids
        np.testing.assert_array_equal(g[1].view(np.uint64), o[1].view(np.uint64))        # word weights
        np.testing.assert_array_equal(g[2], o[2])                                        # node ids at levelsup
        np.testing.assert_array_equal(g[3][0], o[3][0])
        np.testing.assert_array_equal(g[3][1].view(np.uint64), o[3][1].view(np.uint64))  # BowVector values bit for bit
        fvg = {int(g[4].node[j]): [int(x) for x in g[4].feat[g[4].start[j]:g[4].start[j + 1]]] for j in range(len(g[4].node))}
        assert fvg == o[4]
        assert len(set(g[0].tolist())) > 0.5 * len(feats)                                # the descent spreads over the words
    f1, f2 = V.transform(de1, 4), V.transform(de2, 4)
    o1, o2 = oracle.bow_transform(voc, de1, 4), oracle.bow_transform(voc, de2, 4)
    m = uvo.ORBmatcher(0.8, True)
    usable1 = np.ones(len(de1), np.uint8)
    for kf_kf in (False, True):
        mg, ng = m.SearchByBoW(f1[4], de1, kp1["angle"], usable1, f2[4], de2, kp2["angle"], usable2=np.ones(len(de2), np.uint8) if kf_kf else None, kf_kf=kf_kf)
        mo, no = oracle.search_by_bow(kf_kf, o1[4], de1, kp1["angle"], usable1, o2[4], de2, kp2["angle"], np.ones(len(de2), np.uint8) if kf_kf else None,
                                      0.8, True)
        np.testing.assert_array_equal(mg, mo)
        assert ng == no
    m.close()
    V.close()


def test_adaptive_fast_state_is_the_same_with_and_without_the_short_single_frame_chain(uvo, synth):
    """The lane's adaptive FAST state (per level: the threshold the next batch streams at, the fall-back cells of the last batch) is decided
    in k_assemble for ordinary batches and in k_describe<DIRECT> for the frame or two of a latency call (UVO_TUNE_FEW_FRAMES): both go
    through one helper (describe.hip: decide_fast_pass) and must leave the same state after the same sequence of frames -- the keypoints
    cannot show a drift, both modes of a level give the same candidates (src/ORBextractor.cc:792-799)."""
    seq = [img for _, img in _fast_mode_images(synth)] * 2   # textured / low-contrast / flat ...: levels switch mode back and forth
    states = []
    for few in (1, 0):
        ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512)
        ex.tune(uvo.UVO_TUNE_FEW_FRAMES, few)
        st = []
        for img in seq:
            ex(img)
            st.append(tuple(tuple(a.tolist()) for a in ex.fast_state()))
        states.append(st)
        ex.close()
    assert states[0] == states[1]
    assert len(set(s[0] for s in states[0])) > 1, "the sequence must make at least one level change its mode"


@pytest.mark.parametrize("knobs", [{}, {"UVO_TUNE_FEW_FRAMES": 0}, {"UVO_TUNE_ZERO_COPY_OUT": 0}, {"UVO_TUNE_SPIN_WAIT": 0}, {"UVO_TUNE_ZERO_COPY_OUT": 0, "UVO_TUNE_SPIN_WAIT": 0},
                                   {"UVO_TUNE_OCT_WIDE_MAX": 256}, {"UVO_TUNE_PYR_FORM": 1},
                                   {"UVO_TUNE_FEW_FRAMES": 0, "UVO_TUNE_OCT_WIDE_MAX": 256, "UVO_TUNE_ZERO_COPY_OUT": 0, "UVO_TUNE_PYR_FORM": 1}])
def test_single_frame_launch_shapes_give_the_same_bytes(uvo, oracle, frames, knobs):
    """The short launch chain of one or two frames (k_pyr_tiles, the quad-tree sharing its launch with the blur -- or its 1024-thread form
    with the blur in a side stream --, FullDetect without k_assemble, results written into page-locked memory, small inputs read from it, the
    bounded busy wait) against the launches of a large batch, knob by knob: FullDetect of one frame and of two, the top-up call with caller
    keypoints and grid (src/ORBextractor.cc:861-913), and the tracked form (src/Tracking.cc:896-946) -- always the oracle's bytes."""
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512, max_batch=2, max_input_keypoints=600)
    for k, v in knobs.items():
        ex.tune(getattr(uvo, k), v)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    rng = np.random.default_rng(23)
    min_px = 20
    rows, cols = 512 // min_px + 2, 640 // min_px + 2
    for rep in range(2):   # the second round runs on the lane state the first left behind
        for img in frames[:2]:
            kp_o, de_o = oe(img)
            kp_g, de_g = ex(img)
            _assert_same_features(kp_g, de_g, kp_o, de_o, "FullDetect %s" % knobs)
        both = ex.extract_batch(np.stack(frames[:2]))
        for b in range(2):
            kp_o, de_o = oe(frames[b])
            _assert_same_features(both[b][0], both[b][1], kp_o, de_o, "batch of two, frame %d %s" % (b, knobs))
        for n_in, need in ((300, 700), (0, 1000), (7, 30)):
            kin = np.zeros(n_in, uvo.KEYPOINT_DTYPE)
            kin["x"] = rng.uniform(20, 619, n_in).astype(np.float32)
            kin["y"] = rng.uniform(20, 491, n_in).astype(np.float32)
            kin["size"], kin["angle"], kin["response"], kin["octave"], kin["class_id"] = 31, -1, rng.uniform(0, 99, n_in), 0, np.arange(n_in)
            grid = np.zeros((rows, cols), np.int32, order="F")
            for k in kin:
                grid[int(k["y"] / min_px), int(k["x"] / min_px)] += 1
            g_gpu, g_orc = grid.copy(order="F"), grid.copy(order="F")
            kp_g, de_g = ex(frames[2], kin.copy(), g_gpu, min_px, False, need)
            kp_o, de_o = oe(frames[2], kin.copy(), g_orc, min_px, False, need)
            _assert_same_features(kp_g, de_g, kp_o, de_o, "top-up n_in=%d need=%d %s" % (n_in, need, knobs))
            np.testing.assert_array_equal(g_gpu, g_orc)
            # the tracked form: the keypoints only fill the grid, the extractor is called with an empty keypoint vector
            kq, dq, gq = ex.extract_tracked(frames[2], kin, min_px, need, want_grid=True)
            g2 = grid.copy(order="F")
            kp_t, de_t = oe(frames[2], np.zeros(0, uvo.KEYPOINT_DTYPE), g2, min_px, False, need)
            _assert_same_features(kq, dq, kp_t, de_t, "tracked n_in=%d need=%d %s" % (n_in, need, knobs))
            np.testing.assert_array_equal(gq, g2)
    ex.close()
