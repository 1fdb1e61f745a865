"""GPU parity of the pyramid stage (csrc/pyramid.hip) against the oracle's ComputePyramid (src/ORBextractor.cc:963-1004): every padded plane
byte for byte, for every launch form, on level sizes / scale factors / strides / alignments that take each of its code paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planes_equal(ex, oe, nlev, msg, frame=0):
    for l in range(nlev):
        assert ex.level_dims(l) == oe.level_dims(l)
        np.testing.assert_array_equal(ex.read_plane(l, frame=frame), oe.level_plane(l), err_msg="%s level %d" % (msg, l))


def _group(first, tx, ty, wide=False):
    """wide: True = 1024-thread workgroups, "r" = 1024 threads and one output row per work item"""
    return (1 << 25 if wide == "r" else (1 << 24 if wide else 0)) | first << 16 | tx << 8 | ty


@pytest.mark.parametrize("form", ["levels", "tiles", "auto"])
def test_every_pyramid_form_gives_the_oracle_pyramid(uvo, oracle, synth, form):
    """UVO_TUNE_PYR_FORM: one launch per level (k_resize_level), or one launch for all of them (k_pyr_tiles: what a single frame takes by
    default) -- planes, keypoints and descriptors never depend on it."""
    w, h = 640, 512
    img = synth.make_frame(5150, w, h)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    kp_o, de_o = oe(img)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=w, max_height=h)
    ex.tune(uvo.UVO_TUNE_PYR_FORM, {"levels": uvo.UVO_PYR_FORM_LEVELS, "tiles": uvo.UVO_PYR_FORM_TILES, "auto": uvo.UVO_PYR_FORM_AUTO}[form])
    for _ in range(2):
        kp, de = ex(img)
        _planes_equal(ex, oe, 8, form)
        assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all()
    if form != "levels":
        assert "k_pyr_tiles" in _kernels_of_one_call(ex, img) and "k_resize_level" not in _kernels_of_one_call(ex, img)
    ex.close()


def _kernels_of_one_call(ex, img):
    ex.profile(True)
    ex(img)
    names = set(ex.kernel_times().keys())
    ex.profile(False)
    return names


@pytest.mark.parametrize("groups", [[(1, 1, 1)], [(1, 2, 2)], [(1, 4, 4)], [(1, 8, 8)], [(1, 16, 16)], [(1, 3, 7)], [(1, 12, 10), (4, 4, 4)], [(1, 4, 4), (3, 2, 2), (5, 1, 1, True)],
                                    [(l, 3, 2) for l in range(1, 8)], [(1, 8, 8, True), (2, 1, 1, True)], [(1, 5, 5), (7, 2, 1)], [(1, 16, 16, "r")],
                                    [(1, 12, 10, "r"), (4, 4, 4, "r")], [(1, 4, 4, "r"), (3, 2, 2), (5, 1, 1, "r")]])
def test_forced_level_groups_and_tile_grids(uvo, oracle, synth, groups):
    """UVO_TUNE_PYR_TILE_GROUP: any cut of the levels into groups, any tile grid per group, 256- or 1024-thread workgroups (a grid whose tiles
    do not fit the LDS -- 1 x 1 at 640 x 512 -- silently takes the per-level launches)."""
    w, h = 640, 512
    img = synth.make_frame(5160, w, h)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    kp_o, de_o = oe(img)
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=w, max_height=h)
    for g in groups:
        ex.tune(uvo.UVO_TUNE_PYR_TILE_GROUP, _group(*g))
    kp, de = ex(img)
    _planes_equal(ex, oe, 8, str(groups))
    assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all()
    names = _kernels_of_one_call(ex, img)
    fits = groups not in ([(1, 1, 1)], [(1, 8, 8, True), (2, 1, 1, True)])     # (whole levels 1 or 2 of this size do not fit the LDS)
    assert ("k_pyr_tiles" in names) == fits, names
    ex.tune(uvo.UVO_TUNE_PYR_TILE_GROUP, 0)     # back to the defaults
    kp, de = ex(img)
    _planes_equal(ex, oe, 8, "defaults after " + str(groups))
    ex.close()


@pytest.mark.parametrize("shape,scale,nlev", [((321, 243), 1.2, 8), ((752, 480), 1.2, 8), ((1241, 376), 1.2, 8), ((200, 180), 1.1, 6), ((400, 300), 1.5, 4),
                                              ((512, 384), 2.0, 3), ((333, 222), 1.33, 5), ((1920, 1080), 1.2, 8), ((97, 131), 1.2, 3), ((640, 512), 1.2, 2),
                                              ((640, 512), 1.2, 1)])
def test_other_shapes_and_scale_factors(uvo, oracle, synth, shape, scale, nlev):
    """Level sizes that leave partial dword columns and row groups, scale factors on both sides of the 12-byte tap window (above ~1.33 the
    levels gather bytes: per-level launches whatever the form), pyramids of one and two levels."""
    w, h = shape
    img = synth.make_frame(5200 + w, w, h, n_shapes=max(40, w * h // 3000))
    oe = oracle.extractor(500, scale, nlev, 20)
    kp_o, de_o = oe(img)
    ex = uvo.ORBextractor(500, scale, nlev, 0, 20, max_width=w, max_height=h)
    for form in (uvo.UVO_PYR_FORM_AUTO, uvo.UVO_PYR_FORM_TILES, uvo.UVO_PYR_FORM_LEVELS):
        ex.tune(uvo.UVO_TUNE_PYR_FORM, form)
        kp, de = ex(img)
        _planes_equal(ex, oe, nlev, "%dx%d scale %.2f form %d" % (w, h, scale, form))
        assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all()
    ex.close()


def test_batches_of_every_size_in_every_form(uvo, oracle, synth):
    """A batch's frames are independent workgroups; the default form changes with the batch size (tiles up to 8 frames) -- every frame of
    every batch size gives the oracle's planes in every form."""
    w, h = 320, 256
    imgs = synth.make_batch(40, w, h, seed0=5300)
    oe = oracle.extractor(400, 1.2, 6, 20)
    ex = uvo.ORBextractor(400, 1.2, 6, 0, 20, max_width=w, max_height=h, max_batch=40)
    for form in (uvo.UVO_PYR_FORM_AUTO, uvo.UVO_PYR_FORM_TILES):
        ex.tune(uvo.UVO_TUNE_PYR_FORM, form)
        for n in (1, 3, 9, 33, 40):
            ex.extract_batch(imgs[:n])
            for f in sorted({0, n // 2, n - 1}):
                oe(imgs[f])
                _planes_equal(ex, oe, 6, "form %d batch %d frame %d" % (form, n, f), frame=f)
    ex.close()


def test_unaligned_rows_and_the_two_lane_pipeline(uvo, oracle, synth):
    """Image widths that are no multiple of 4 (level 1 cannot read the image in place: the padded copy of level 0 is the source), and the
    two-lane pipeline (each lane has its own planes)."""
    for (w, h) in ((637, 509), (333, 301), (636, 500)):
        img = synth.make_frame(5400 + w, w, h)
        oe = oracle.extractor(600, 1.2, 7, 20)
        kp_o, de_o = oe(img)
        ex = uvo.ORBextractor(600, 1.2, 7, 0, 20, max_width=w, max_height=h)
        ex.set_pipeline(2)
        for form in (uvo.UVO_PYR_FORM_TILES, uvo.UVO_PYR_FORM_LEVELS):
            ex.tune(uvo.UVO_TUNE_PYR_FORM, form)
            for _ in range(3):
                kp, de = ex(img)
                _planes_equal(ex, oe, 7, "%dx%d" % (w, h))
                assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all()
        ex.close()


def _device_extract(uvo, ex, torch, buf, off, B, W, H, stride, fstride):
    cap = ex.cap
    kp = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    de = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(buf.data_ptr() + off, B, W, H, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap, stride=stride, frame_stride=fstride)
    ex.synchronize()
    n = n.cpu().numpy()
    return [(kp[b, :n[b]].cpu().numpy(), de[b, :n[b]].cpu().numpy()) for b in range(B)]


@pytest.mark.parametrize("W,H,stride_extra,off", [(640, 512, 0, 0), (640, 512, 64, 4 * 37), (636, 500, 4, 8), (640, 512, 3, 0), (640, 512, 64, 1),
                                                  (638, 510, 2, 0), (320, 240, 0, 0), (128, 96, 12, 4)])
def test_level0_is_read_in_place_from_the_callers_rows(uvo, oracle, synth, W, H, stride_extra, off):
    """UVO_TUNE_LEVEL0_INPLACE (default on): with dword-aligned rows of a width that is a multiple of 4 no padded copy of the image is made
    -- FAST, the orientation patch and the resize to level 1 read the caller's rows, the blur reflects its border on the fly.  The images
    sit inside a poisoned buffer (every byte around and between the rows is noise): a read outside the image would show.  Rows or widths
    that are not dword multiples, and an odd base address, take the padded copy; the keypoints never depend on which."""
    torch = pytest.importorskip("torch")
    B, nlev = 3, (6 if W >= 600 else 4 if W >= 320 else 2)
    stride = W + stride_extra
    fstride = stride * H + 4 * 11 * (stride % 4 == 0) + (0 if stride % 4 == 0 else 7)
    rng = np.random.default_rng(W * 7 + H + off)
    host = rng.integers(0, 256, off + B * fstride + 4096, dtype=np.uint8)
    frames = [synth.make_frame(6100 + W + b, W, H) for b in range(B)]
    for b in range(B):
        rows = host[off + b * fstride: off + b * fstride + stride * H].reshape(H, stride)
        rows[:, :W] = frames[b]
    buf = torch.from_numpy(host).cuda()
    oe = oracle.extractor(500, 1.2, nlev, 20)
    ex = uvo.ORBextractor(500, 1.2, nlev, 0, 20, max_width=W, max_height=H, max_batch=B)
    got_on = _device_extract(uvo, ex, torch, buf, off, B, W, H, stride, fstride)
    planes_on = [[ex.read_plane(l, frame=b) for l in range(nlev)] for b in range(B)]      # level 0: made on demand from the caller's rows
    blur_on = [[ex.read_plane(l, blurred=True, frame=b) for l in range(nlev)] for b in range(B)]
    ex.tune(uvo.UVO_TUNE_LEVEL0_INPLACE, 0)
    got_off = _device_extract(uvo, ex, torch, buf, off, B, W, H, stride, fstride)
    for b in range(B):
        kp_o, de_o = oe(frames[b])
        for got in (got_on, got_off):
            kp, de = got[b]
            assert len(kp) == len(kp_o) and (de == de_o).all(), "frame %d" % b
            assert np.array_equal(kp[:, 0], kp_o["x"]) and np.array_equal(kp[:, 1], kp_o["y"]) and np.array_equal(kp[:, 3], kp_o["angle"])
        for l in range(nlev):
            np.testing.assert_array_equal(planes_on[b][l], oe.level_plane(l), err_msg="frame %d level %d" % (b, l))
            np.testing.assert_array_equal(ex.read_plane(l, frame=b), oe.level_plane(l), err_msg="frame %d level %d (copy)" % (b, l))
            # the blurred planes of the two forms: the whole region the blur writes (interior + the 2-px ring a descriptor can reach)
            np.testing.assert_array_equal(blur_on[b][l][14:-14, 14:-14], ex.read_plane(l, blurred=True, frame=b)[14:-14, 14:-14])
            if (kp_o["octave"] == l).any():
                np.testing.assert_array_equal(blur_on[b][l][14:-14, 14:-14], oe.level_plane(l, blurred=True)[14:-14, 14:-14])
    assert (buf.cpu().numpy() == host).all()                 # the caller's buffer is read only
    ex.close()


@pytest.mark.parametrize("ring", [0, 4, 8, 12])
def test_the_resize_launches_may_stop_four_pixels_outside_the_image(uvo, oracle, synth, ring):
    """UVO_TUNE_PYR_RING: the chain's resize launches write a level's image and `ring` pixels of its border (default 4: nothing reads further
    out -- the blur reaches 3 and copies 4 into the blurred plane's ring); 0 = the whole 16-pixel border.  Keypoints, descriptors and the
    blurred planes never depend on it; uvo_extractor_read_plane completes the border of an un-blurred level on demand, for these tests."""
    for (w, h, nlev) in ((640, 512, 8), (637, 509, 7), (333, 301, 5)):
        img = synth.make_frame(6300 + w, w, h)
        oe = oracle.extractor(700, 1.2, nlev, 20)
        kp_o, de_o = oe(img)
        ex = uvo.ORBextractor(700, 1.2, nlev, 0, 20, max_width=w, max_height=h)
        ex.tune(uvo.UVO_TUNE_PYR_RING, ring)
        for _ in range(2):
            kp, de = ex(img)
            assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all()
            for l in range(nlev):
                if (kp_o["octave"] == l).any():
                    np.testing.assert_array_equal(ex.read_plane(l, blurred=True)[12:-12, 12:-12], oe.level_plane(l, blurred=True)[12:-12, 12:-12],
                                                  err_msg="blurred level %d" % l)
            _planes_equal(ex, oe, nlev, "%dx%d ring %d" % (w, h, ring))
        ex.close()
    with pytest.raises(uvo.UvoError):
        uvo.ORBextractor(100, 1.2, 4, 0, 20, max_width=320, max_height=240).tune(uvo.UVO_TUNE_PYR_RING, 5)


def test_kernel_times_report_the_spread_of_the_launches(uvo, synth):
    """uvo_extractor_kernel_times: behind the kernels' rows the spread rows of every kernel with two or more launches -- shortest / median /
    longest launch, start-to-start period of consecutive launches and half the period of launches two apart (bench.py's `step_spread`)."""
    img = synth.make_frame(5400, 320, 256)
    ex = uvo.ORBextractor(300, 1.2, 4, 0, 20, max_width=320, max_height=256)
    ex.profile(True)
    for _ in range(6):
        ex(img)
    kt = ex.kernel_times()
    sp = ex.last_spread
    ex.profile(False)
    assert all(":" not in k for k in kt) and "k_fast_score" in kt and kt["k_fast_score"][1] == 6
    s = sp["k_fast_score"]
    assert set(s) == {"min", "p50", "max", "period_min", "period_p50", "period_max", "period2_min", "period2_p50", "period2_max"}
    assert 0 < s["min"] <= s["p50"] <= s["max"] and abs(kt["k_fast_score"][0] - 6 * s["p50"]) < 6 * (s["max"] - s["min"]) + 1e-3
    assert s["period_min"] >= s["min"] * 0.5 and s["period_min"] <= s["period_p50"] <= s["period_max"]   # a call's launches are a chain: the next FAST pass starts a whole call later
    assert s["period2_min"] <= s["period2_p50"] <= s["period2_max"]
    assert ex.kernel_times() == {} and ex.last_spread == {}    # reported and cleared
    ex.close()
