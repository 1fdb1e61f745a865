"""GPU parity of the pyramid stage (csrc/pyramid.hip) against the oracle's ComputePyramid (src/ORBextractor.cc:963-1004): every padded plane
byte for byte, for every launch form, on level sizes / scale factors / strides / alignments that take each of its code paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planes_equal(ex, oe, nlev, msg, frame=0):
    for l in range(nlev):
        assert ex.level_dims(l) == oe.level_dims(l)
        np.testing.assert_array_equal(ex.read_plane(l, frame=frame), oe.level_plane(l), err_msg="%s level %d" % (msg, l))


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
