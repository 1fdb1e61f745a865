"""Independent cross-checks of the oracle's [OCV-RECALL] primitives against third-party implementations that
happen to be in the image: torch (bilinear interpolate, conv2d, reflect pad) and scikit-image 0.18.3 under /opt/conda/bin/python3.9
(corner_fast, corner_orientations, its copy of the ORB sampling pattern), plus numpy's arctan2.

What this is NOT: a pin.  None of these is OpenCV; they agree with the oracle up to the tolerance each comparison states, which bounds
the damage a wrong recollection could do -- a wrong half-pixel convention, tap table, border mode, ring order or patch shape fails
here -- but the last grey level / tie is only settled by OpenCV 3.4.x itself (tools/pin/).  DESIGN.md section 5 says so."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY39 = "/opt/conda/bin/python3.9"


@pytest.fixture(scope="module")
def frames(synth):
    return [synth.make_frame(9100 + i, 320, 256, n_shapes=120) for i in range(3)]


def test_reflect101_border_equals_torch_reflect_pad(oracle, frames):
    """cv::copyMakeBorder(BORDER_REFLECT_101) (src/ORBextractor.cc:988,996) = torch's 'reflect' padding (the edge pixel is not repeated)."""
    import torch
    import torch.nn.functional as F
    for img in frames + [np.arange(40 * 33, dtype=np.uint8).reshape(40, 33)]:
        t = torch.from_numpy(img.astype(np.float32))[None, None]
        ref = F.pad(t, (16, 16, 16, 16), mode="reflect")[0, 0].numpy().astype(np.uint8)
        np.testing.assert_array_equal(oracle.border101(img, 16), ref)


@pytest.mark.parametrize("scale", [1.2, 1.1, 1.5, 2.0])
def test_resize_linear_is_within_one_grey_level_of_float_bilinear(oracle, frames, scale):
    """cv::resize(INTER_LINEAR) (src/ORBextractor.cc:982): 11-bit fixed-point weights, two truncating shifts -- against torch's float
    bilinear interpolation with the same half-pixel centres (align_corners = False, no antialiasing).  A wrong sampling convention
    (align_corners, pixel-corner vs pixel-centre, a swapped clamp) moves edges by whole grey levels; the fixed-point rounding does not."""
    import torch
    import torch.nn.functional as F
    for img in frames:
        h, w = img.shape
        dw, dh = int(round(w / scale)), int(round(h / scale))
        got = oracle.resize_linear(img, dw, dh).astype(np.int32)
        t = torch.from_numpy(img.astype(np.float64))[None, None]
        ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
        err = np.abs(got - ref)
        assert err.max() <= 1.0, err.max()
        assert (err > 0.75).mean() < 0.02        # and hardly ever near one: the fixed-point truncations cost at most 3/4 of a level
        assert abs((got - ref).mean()) < 0.2     # no brightness drift


def test_gaussian_blur_engine_and_taps_against_float_convolution(oracle, frames):
    """cv::GaussianBlur(7 x 7, sigma 2) on the ROI of the padded plane (src/ORBextractor.cc:942), two comparisons:
    (1) the ENGINE -- separable, border taps read the real pad pixels, one rounding at the end -- against a float conv2d with the same
        8-bit quantised taps (18, 34, 49, 55, 49, 34, 18) / 256 per pass: identical except where the float sum lands within rounding
        error of a tie;
    (2) the TAPS -- against the ideal float Gaussian exp(-x^2 / 8) normalised to 1: the quantised taps sum to 257 / 256 per pass, so the
        integer blur is 0.78 % brighter; what remains after that gain is below one grey level."""
    import torch
    import torch.nn.functional as F
    taps = np.array(oracle.gauss_taps(), np.float64)
    assert taps.tolist() == [18, 34, 49, 55, 49, 34, 18]
    g = np.exp(-(np.arange(7) - 3.0) ** 2 / 8.0)
    g /= g.sum()
    np.testing.assert_array_equal(np.rint(g.astype(np.float32) * 256.0), taps)   # cvRound(getGaussianKernel(7, 2) * 256)
    for img in frames:
        plane = oracle.border101(img, 16)
        got = oracle.gauss7_padded(plane, 16)[16:-16, 16:-16].astype(np.float64)
        t = torch.from_numpy(plane.astype(np.float64))[None, None]

        def sep(k):
            kk = torch.from_numpy(k)
            x = F.conv2d(t, kk.view(1, 1, 1, 7))
            x = F.conv2d(x, kk.view(1, 1, 7, 1))
            return x[0, 0].numpy()[13:-13, 13:-13]            # 'valid' output aligned with the ROI (pad 16 - radius 3)

        q = sep(taps / 256.0)
        ref_q = np.clip(np.floor(q + 0.5), 0, 255)
        assert (got != ref_q).mean() < 1e-4                    # same engine, same taps: equal but for float ties
        assert np.abs(got - ref_q).max() <= 1
        ideal = sep(g)
        resid = got - np.clip(ideal * (257.0 / 256.0) ** 2, 0, 255)
        assert np.abs(resid).max() <= 1.0 and abs(resid.mean()) < 0.1
        assert np.abs(got - ideal).max() <= 3.0                # and in absolute terms the integer blur stays within 3 levels of the ideal one


def test_fast_atan2_is_within_its_documented_error_of_arctan2(oracle):
    """cv::fastAtan2 (src/ORBextractor.cc:151): degrees in [0, 360), documented accuracy about 0.3 degrees."""
    rng = np.random.default_rng(5)
    y = np.concatenate([rng.normal(0, 1000, 4000), [0, 0, 1, -1, 5, -5, 0]]).astype(np.float32)
    x = np.concatenate([rng.normal(0, 1000, 4000), [1, -1, 0, 0, 5, 5, 0]]).astype(np.float32)
    got = np.array([oracle.fast_atan2(float(a), float(b)) for a, b in zip(y, x)])
    ref = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
    d = np.abs(got - ref)
    d = np.minimum(d, 360.0 - d)
    assert d[:-1].max() < 0.3
    assert got[-1] == 0.0 and (got >= 0).all() and (got < 360.0 + 1e-3).all()


def _skimage_probe(img, thresholds, corners, **extra):
    if not os.path.exists(PY39):
        pytest.skip("no /opt/conda/bin/python3.9 (scikit-image) in this image")
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(a, img=img, thresholds=np.array(thresholds), corners=np.asarray(corners, np.int64), **extra)
        r = subprocess.run([PY39, "-W", "ignore", os.path.join(ROOT, "tests", "crosscheck", "skimage_probe.py"), a, b], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("scikit-image probe failed: " + r.stderr[-300:])
        return dict(np.load(b))


def test_fast_corner_set_equals_scikit_image(oracle, frames):
    """cv::FAST type 9_16 (src/ORBextractor.cc:792,797): the corner PREDICATE -- a 16-pixel ring of radius 3, nine contiguous
    pixels all brighter than centre + t or all darker than centre - t -- against scikit-image's corner_fast(n = 9), at the two thresholds
    the extractor uses.  Identical corner sets on every frame (the score and the suppression are OpenCV's own and are not compared)."""
    for i, img in enumerate(frames):
        out = _skimage_probe(img, [7, 20], np.zeros((1, 2)))
        for t in (7, 20):
            kp = oracle.fast(img, t, nms=False)
            got = np.zeros(img.shape, bool)
            got[kp["y"].astype(int), kp["x"].astype(int)] = True
            ref = out["fast_%d" % t].copy()
            ref[:3] = ref[-3:] = False
            ref[:, :3] = ref[:, -3:] = False
            assert got.sum() > 200
            np.testing.assert_array_equal(got, ref, err_msg="frame %d threshold %d" % (i, t))


def test_corner_score_equals_the_largest_threshold_scikit_image_still_calls_a_corner(oracle, frames):
    """cornerScore<16> (behind cv::FAST's `response`, which drives the 3x3 suppression and every selection after it): "the largest threshold
    that keeps the pixel a corner" -- evaluated literally with scikit-image's segment test at every threshold 0 .. 200."""
    img = frames[1]
    out = _skimage_probe(img, [20], np.zeros((1, 2)), score_upto=np.array(200))
    ref = out["score"]
    ref[:3] = ref[-3:] = -1
    ref[:, :3] = ref[:, -3:] = -1
    assert ref.max() < 200                                         # the sweep saw every corner die
    for th in (7, 20):
        kp = oracle.fast(img, th, nms=False)
        y, x = kp["y"].astype(int), kp["x"].astype(int)
        assert len(kp) > 200
        np.testing.assert_array_equal(kp["response"].astype(int), ref[y, x], err_msg="threshold %d" % th)
        # and nothing with a score of at least the threshold is missing
        assert len(kp) == int((ref >= th).sum()), (len(kp), int((ref >= th).sum()))


def test_orientation_patch_and_pattern_equal_scikit_images(oracle, frames):
    """IC_Angle (src/ORBextractor.cc:125-152): the circular patch (umax table of :494-511) and the centroid angle, against scikit-image's
    OFAST mask and corner_orientations (arctan2 of the same moments: within fastAtan2's 0.3 degrees); and the 256 rBRIEF point pairs
    (bit_pattern_31_, :198-456) against the copy of the published pattern that scikit-image ships."""
    img = frames[0]
    rng = np.random.default_rng(9)
    pts = np.stack([rng.integers(20, img.shape[0] - 20, 300), rng.integers(20, img.shape[1] - 20, 300)], 1)   # (row, col)
    out = _skimage_probe(img, [20], pts)
    oe = oracle.extractor(500, 1.2, 8, 20)
    # the patch: rows v = -15 .. 15, |u| <= umax[|v|]
    mask = np.zeros((31, 31), np.uint8)
    for v in range(-15, 16):
        um = int(oe.umax[abs(v)])
        mask[v + 15, 15 - um:15 + um + 1] = 1
    np.testing.assert_array_equal(mask, out["ofast_mask"])
    plane = oracle.border101(img, 16)
    got = np.array([oe.ic_angle(plane, float(c), float(r)) for r, c in pts])
    ref = np.degrees(out["orientations"]) % 360.0
    d = np.abs(got - ref)
    d = np.minimum(d, 360.0 - d)
    assert d.max() < 0.3, d.max()
    # the sampling pattern: the same 256 x (x0, y0, x1, y1) numbers in the same order
    pat = oe.pattern().reshape(256, 4)
    np.testing.assert_array_equal(pat[:, :2], out["pos0"])
    np.testing.assert_array_equal(pat[:, 2:], out["pos1"])


def test_steered_brief_bits_equal_scikit_images_descriptor_loop(oracle, frames):
    """computeOrbDescriptor (src/ORBextractor.cc:156-195): which pixel pairs are compared, how they turn with the keypoint's angle, which
    way the comparison goes and where the bit lands in the 32 bytes -- against scikit-image's ORB descriptor loop, fed with the oracle's own
    blurred level 0, keypoints and angles.  The two round the rotated sample points differently (cvRound on float32 products: half to
    even; C round() on float64 products: half away from zero), so a sample point that falls within float32 rounding of a half-integer may
    be read one pixel apart: a handful of bits per thousand keypoints.  A wrong pair order, steering sign, comparison direction or bit
    order would flip half of them."""
    total = bits = 0
    for img in frames:
        oe = oracle.extractor(1000, 1.2, 8, 20)
        kp, desc = oe(img)
        lv0 = kp["octave"] == 0
        assert lv0.sum() > 100
        blurred = oe.level_plane(0, blurred=True)                      # padded by 16
        rc = np.stack([np.rint(kp["y"][lv0]).astype(np.int64) + 16, np.rint(kp["x"][lv0]).astype(np.int64) + 16], 1)
        out = _skimage_probe(img, [20], np.zeros((1, 2)), desc_img=blurred, desc_kp=rc, desc_angle=np.radians(kp["angle"][lv0].astype(np.float64)))
        ref = out["descriptors"].astype(bool)                          # (M, 256): bit j of the descriptor
        got = np.unpackbits(desc[lv0], axis=1, bitorder="little").astype(bool)
        diff = got != ref
        assert diff.sum(1).max() <= 3, diff.sum(1).max()
        total += int(diff.sum())
        bits += diff.size
    assert total <= bits // 2000, (total, bits)                        # measured: 0 of 56 064 bits on the first frame


@pytest.mark.parametrize("w", [320, 318, 157])
def test_sse2_blur_contract_is_round_half_even_on_the_vector_columns_only(oracle, w):
    """The selectable x86-64 contract of the blur's column pass (oracle rounding = 1 / UVO_BLUR_ROUNDING_SSE2): against a numpy evaluation
    of the exact column sums, the result is round-half-to-even on the image columns 0 .. (w & ~3) - 1 and round-half-up on the last w % 4,
    i.e. it differs from the default contract exactly where a sum is an exact .5 whose upper neighbour is odd."""
    rng = np.random.default_rng(w)
    H = 1500
    img = rng.integers(0, 256, (H, w)).astype(np.uint8)
    plane = oracle.border101(img, 16)
    k = np.array(oracle.gauss_taps(), np.int64)
    p = plane.astype(np.int64)
    rows = sum(k[i] * p[:, 13 + i:13 + i + w] for i in range(7))                       # row pass over every padded row
    s = sum(k[j] * rows[13 + j:13 + j + H] for j in range(7))                         # exact column sums of the ROI
    half_up = np.clip((s + (1 << 15)) >> 16, 0, 255)
    tie = (s & 0xffff) == 0x8000
    half_even = np.where(tie, half_up & ~1, half_up)
    vec = np.arange(w)[None, :] < (w & ~3)
    np.testing.assert_array_equal(oracle.gauss7_padded_ex(plane, 16, 0)[16:-16, 16:-16], half_up)
    np.testing.assert_array_equal(oracle.gauss7_padded_ex(plane, 16, 1)[16:-16, 16:-16], np.where(vec, half_even, half_up))
    assert (tie & vec & ((half_up & 1) == 1)).sum() >= 1                                   # the image holds such ties
