"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact bar: pyramid / blurred planes, FAST candidates (as sets), keypoints (x, y, size, angle, response, octave,
class_id as raw bytes) and 256-bit descriptors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def test_hbm_resident_pipeline_depth2(uvo, oracle, synth):
    """uvo_extract_batch_device with two alternating scratch sets / streams feeding the batched HBM-resident matcher."""
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
    for d_img, kp, de, n, i0, i1, d0, d1 in outs:   # three calls back to back, no host sync in between
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap)
        mt.wait_extractor(ex)
        # pair p = (frame p, frame p+1) for p < B-1
        mt.knn2_batch_device(B - 1, de.data_ptr(), n.data_ptr(), cap, de.data_ptr() + cap * 32, n.data_ptr() + 4, cap, i0.data_ptr(),
                             d0.data_ptr(), i1.data_ptr(), d1.data_ptr())
        mt.release_to_extractor(ex)
    ex.synchronize()
    mt.synchronize()
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
