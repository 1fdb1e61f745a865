"""GPU parity of the split pyramid mode (UVO_PYR_MODE_SPLIT: k_pyr_stream + k_pyramid, csrc/pyramid.hip + csrc/pyr_schedule.hpp) against
the oracle's ComputePyramid (src/ORBextractor.cc:963-1004): every padded plane byte for byte, for every split point, band count,
workgroup width, run length and step size the schedules can be compiled for -- the keypoints never depend on those knobs.  (The default
mode, one launch per level, is what every other GPU test runs.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planes_equal(ex, oe, nlev, msg, frame=0):
    for l in range(nlev):
        assert ex.level_dims(l) == oe.level_dims(l)
        np.testing.assert_array_equal(ex.read_plane(l, frame=frame), oe.level_plane(l), err_msg="%s level %d" % (msg, l))


@pytest.mark.parametrize("tail", [0, 1, 2, 3, 4, 6, 7, 8])
def test_every_split_between_streaming_and_fused_launches(uvo, oracle, synth, tail):
    """UVO_TUNE_PYR_TAIL: the levels below it stream (k_pyr_stream: level 1 reads the image in place, the border copy rides along),
    the rest shares the fused launch (k_pyramid) -- 0 = everything fused, 8 = everything streams; runs of 1 .. 9 blocks per wavefront."""
    w, h = 640, 512
    img = synth.make_frame(5150, w, h)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    kp_o, de_o = oe(img)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=w, max_height=h)
    ex.tune(uvo.UVO_TUNE_PYR_MODE, uvo.UVO_PYR_MODE_SPLIT)
    ex.tune(uvo.UVO_TUNE_PYR_TAIL, tail)
    for run in (0, 1, 4, 9, 64):
        ex.tune(uvo.UVO_TUNE_PYR_RUN, run)
        for bands in (1, 4):
            ex.tune(uvo.UVO_TUNE_PYR_BANDS, bands)
            kp, de = ex(img)
            _planes_equal(ex, oe, 8, "tail %d run %d bands %d" % (tail, run, bands))
            assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all()
    ex.close()


@pytest.mark.parametrize("waves", [0, 8, 16])
def test_every_band_count_and_step_size_gives_the_oracle_pyramid(uvo, oracle, synth, waves):
    w, h = 640, 512
    img = synth.make_frame(5100, w, h)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    kp_o, de_o = oe(img)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=w, max_height=h)
    ex.tune(uvo.UVO_TUNE_PYR_MODE, uvo.UVO_PYR_MODE_SPLIT)
    ex.tune(uvo.UVO_TUNE_PYR_TAIL, 0)   # the fused launch builds every level
    ex.tune(uvo.UVO_TUNE_PYR_WAVES, waves)
    for rows in (7, 4, 1):
        ex.tune(uvo.UVO_TUNE_PYR_ROWS, rows)
        for bands in (1, 2, 4, 8, 16):
            ex.tune(uvo.UVO_TUNE_PYR_BANDS, bands)
            kp, de = ex(img)
            _planes_equal(ex, oe, 8, "waves %d rows %d bands %d" % (waves, rows, bands))
            assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all()
    ex.close()


@pytest.mark.parametrize("shape,scale,nlev", [((321, 243), 1.2, 8), ((752, 480), 1.2, 8), ((1241, 376), 1.2, 8), ((200, 180), 1.1, 6), ((400, 300), 1.5, 4),
                                              ((512, 384), 2.0, 3), ((333, 222), 1.33, 5), ((1920, 1080), 1.2, 8), ((97, 131), 1.2, 3)])
def test_other_shapes_and_scale_factors(uvo, oracle, synth, shape, scale, nlev):
    """Level sizes that leave partial column chunks, scale factors on both sides of the 12-byte tap window (byte-gather path above
    ~1.33), widths above one 1024-byte copy chunk."""
    w, h = shape
    img = synth.make_frame(5200 + w, w, h, n_shapes=max(40, w * h // 3000))
    oe = oracle.extractor(500, scale, nlev, 20)
    oe(img)
    ex = uvo.ORBextractor(500, scale, nlev, 0, 20, max_width=w, max_height=h)
    ex.tune(uvo.UVO_TUNE_PYR_MODE, uvo.UVO_PYR_MODE_SPLIT)
    for tail in (3, 0, 1, nlev):
        ex.tune(uvo.UVO_TUNE_PYR_TAIL, tail)
        for bands in (0, 1, 4):
            ex.tune(uvo.UVO_TUNE_PYR_BANDS, bands)
            ex(img)
            _planes_equal(ex, oe, nlev, "%dx%d scale %.2f tail %d bands %d" % (w, h, scale, tail, bands))
    ex.close()


def test_batches_take_their_band_count_from_the_batch_size(uvo, oracle, synth):
    """A batch's frames are independent workgroups; the default band count changes with the batch size (16 bands for one frame ...
    one band from 256 frames on) -- every frame of every batch size gives the oracle's planes."""
    w, h = 320, 256
    imgs = synth.make_batch(40, w, h, seed0=5300)
    oe = oracle.extractor(400, 1.2, 6, 20)
    ex = uvo.ORBextractor(400, 1.2, 6, 0, 20, max_width=w, max_height=h, max_batch=40)
    ex.tune(uvo.UVO_TUNE_PYR_MODE, uvo.UVO_PYR_MODE_SPLIT)
    for n in (1, 3, 9, 33, 40):
        ex.extract_batch(imgs[:n])
        for f in sorted({0, n // 2, n - 1}):
            oe(imgs[f])
            _planes_equal(ex, oe, 6, "batch %d frame %d" % (n, f), frame=f)
    ex.close()


def test_unaligned_rows_and_strided_input(uvo, oracle, synth):
    """Image widths that are no multiple of 16 (groups that straddle an image edge gather bytes) or of 4 (level 1 cannot read the image
    in place: a copy-only launch runs first), and the two-lane pipeline (each lane has its own planes)."""
    for (w, h) in ((637, 509), (333, 301), (636, 500)):
        img = synth.make_frame(5400 + w, w, h)
        oe = oracle.extractor(600, 1.2, 7, 20)
        kp_o, de_o = oe(img)
        ex = uvo.ORBextractor(600, 1.2, 7, 0, 20, max_width=w, max_height=h)
        ex.tune(uvo.UVO_TUNE_PYR_MODE, uvo.UVO_PYR_MODE_SPLIT)
        ex.set_pipeline(2)
        for _ in range(3):
            kp, de = ex(img)
            _planes_equal(ex, oe, 7, "%dx%d" % (w, h))
            assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all()
        ex.close()


def test_chain_and_split_modes_agree(uvo, oracle, synth):
    img = synth.make_frame(5500, 640, 512)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=640, max_height=512)
    ex(img)
    chain = [ex.read_plane(l) for l in range(8)]
    ex.tune(uvo.UVO_TUNE_PYR_MODE, uvo.UVO_PYR_MODE_SPLIT)
    ex(img)
    for l in range(8):
        np.testing.assert_array_equal(ex.read_plane(l), chain[l])
    ex.close()
