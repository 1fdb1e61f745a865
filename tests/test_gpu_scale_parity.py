"""The code path bench.py times, under the oracle at the benchmark's own sizes.

BASELINE.json configs[2] (batch 256 @ 640x512, 1000 features: `k_octree<256>`, the XCD-contiguous dealing of 256-frame grids,
`k_assemble<false>` at scale) and configs[3]'s per-GPU share (batch 128 @ 1920x1080, 2000 features) are run exactly as the bench
runs them -- HBM-resident, pipeline depth 2, all-pairs knn-2 of consecutive frames on the matcher's stream -- and EVERY frame is
compared with the CPU oracle byte for byte (keypoints as raw 28-byte records, descriptors, knn-2 rows).  A second group forces each
launch shape of the quad-tree kernel (DistributeOctTree, src/ORBextractor.cc:1006-1230) on the same input and holds both to the
oracle, and a third covers the host-buffer entry points after uvo_extractor_set_pipeline(2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _kp_bytes(kp):
    return np.ascontiguousarray(kp).view(np.uint8).reshape(len(kp), -1)


def _same(kp_g, de_g, kp_o, de_o, what):
    assert len(kp_g) == len(kp_o), "%s: %d keypoints on the GPU, %d in the oracle" % (what, len(kp_g), len(kp_o))
    bad = np.nonzero((_kp_bytes(kp_g) != _kp_bytes(kp_o)).any(1))[0]
    assert len(bad) == 0, "%s: %d keypoints differ, first %d: gpu=%s oracle=%s" % (what, len(bad), bad[0], kp_g[bad[0]], kp_o[bad[0]])
    assert (de_g == de_o).all(), "%s: descriptors differ" % what


def _run_bench_path(uvo, frames, nfeat, fast_th, passes=3, tune=None, matcher_stream="lane"):
    """bench.py's step: extract_batch_device on alternating lanes + knn-2 of (frame i, frame i+1), `passes` steps back to back
    without a host synchronisation; returns the outputs of the LAST step on each lane (so both lanes' scratch sets are checked)."""
    import torch
    B, H, W = frames.shape
    dev = torch.device("cuda", 0)
    ex = uvo.ORBextractor(nfeat, 1.2, 8, 0, fast_th, max_width=W, max_height=H, max_batch=B)
    if tune is not None:
        ex.tune(uvo.UVO_TUNE_OCT_WIDE_MAX, tune)
    ex.set_pipeline(2)
    cap = ex.cap
    mt = uvo.ORBmatcher(0.8, max_query=cap, max_train=cap, max_batch=B)
    d_img = torch.from_numpy(frames).to(dev)

    class Out:
        def __init__(self):
            self.kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
            self.de = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
            self.n = torch.zeros(B, dtype=torch.int32, device=dev)
            self.i0 = torch.full((B, cap), -7, dtype=torch.int32, device=dev)
            self.i1 = torch.full((B, cap), -7, dtype=torch.int32, device=dev)
            self.d0 = torch.zeros((B, cap), dtype=torch.int16, device=dev)
            self.d1 = torch.zeros((B, cap), dtype=torch.int16, device=dev)

    outs = [Out(), Out()]
    torch.cuda.synchronize()
    for k in range(passes):
        o = outs[k % 2]
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, o.kp.data_ptr(), o.de.data_ptr(), o.n.data_ptr(), cap)
        if matcher_stream == "lane":   # bench.py's default: the matching queues up behind the batch in the lane's own stream
            mt.attach(ex)
        else:                          # the matcher's own stream, ordered by events
            mt.wait_extractor(ex)
        mt.knn2_batch_device(B - 1, o.de.data_ptr(), o.n.data_ptr(), cap, o.de.data_ptr() + cap * 32, o.n.data_ptr() + 4, cap,
                             o.i0.data_ptr(), o.d0.data_ptr(), o.i1.data_ptr(), o.d1.data_ptr())
        if matcher_stream != "lane":
            mt.release_to_extractor(ex)
    ex.synchronize()
    mt.synchronize()
    mt.attach(None)
    res = []
    for o in outs:
        res.append(dict(n=o.n.cpu().numpy(), kp=o.kp.cpu().numpy(), de=o.de.cpu().numpy(), i0=o.i0.cpu().numpy(), i1=o.i1.cpu().numpy(),
                        d0=o.d0.cpu().numpy().astype(np.uint16), d1=o.d1.cpu().numpy().astype(np.uint16)))
    ex.close()
    mt.close()
    return res


def _check_against_oracle(uvo, oracle, frames, nfeat, fast_th, res, what, min_kp):
    oe = oracle.extractor(nfeat, 1.2, 8, fast_th)
    B = len(frames)
    ref = [oe(frames[b]) for b in range(B)]
    assert min(len(r[0]) for r in ref) >= min_kp
    for li, r in enumerate(res):
        for b in range(B):
            n = int(r["n"][b])
            kp_g = np.ascontiguousarray(r["kp"][b, :n]).view(uvo.KEYPOINT_DTYPE).reshape(-1)
            _same(kp_g, r["de"][b, :n], ref[b][0], ref[b][1], "%s lane %d frame %d" % (what, li, b))
        for b in range(B - 1):
            o = oracle.knn2(ref[b][1], ref[b + 1][1])
            nq = len(ref[b][1])
            np.testing.assert_array_equal(r["i0"][b, :nq], o[0], err_msg="%s lane %d pair %d idx0" % (what, li, b))
            np.testing.assert_array_equal(r["d0"][b, :nq].astype(np.int32), o[1], err_msg="%s lane %d pair %d d0" % (what, li, b))
            np.testing.assert_array_equal(r["i1"][b, :nq], o[2], err_msg="%s lane %d pair %d idx1" % (what, li, b))
            np.testing.assert_array_equal(r["d1"][b, :nq].astype(np.int32), o[3], err_msg="%s lane %d pair %d d1" % (what, li, b))
    return ref


@pytest.mark.parametrize("batch,matcher_stream", [(64, "lane"), (64, "own"), (256, "lane")])
def test_configs2_throughput_path_every_frame_vs_oracle(uvo, oracle, synth, batch, matcher_stream):
    """BASELINE.json configs[2] as bench.py runs it (batch 256: 2048 quad-tree problems -> k_octree<256>, four per CU); the matching in
    the extracting lane's stream (uvo_matcher_attach_extractor, bench.py's default) and in the matcher's own stream behind events."""
    frames = synth.make_sequence(0, batch, 640, 512)
    res = _run_bench_path(uvo, frames, 1000, 20, matcher_stream=matcher_stream)
    _check_against_oracle(uvo, oracle, frames, 1000, 20, res, "configs[2] batch %d" % batch, 1000)


def test_corner_rich_frames_at_scale_every_frame_vs_oracle(uvo, oracle, synth):
    """The round-2 generator's frames (noise accumulating along a chain: up to 18 % of the pixels pass as FAST corners, 1100 per
    248 x 24 region): nearly every region of k_fast_score overflows its LDS corner list and finishes its non-max suppression out of
    memory, the quad-tree gathers 50 000 candidates per frame -- the paths the sensor-noise frames of the test above leave cold."""
    frames = synth.make_sequence(0, 96, 640, 512, noise="cumulative")
    res = _run_bench_path(uvo, frames, 1000, 20)
    _check_against_oracle(uvo, oracle, frames, 1000, 20, res, "corner-rich batch 96", 1000)


def _run_bench_path_mode(uvo, frames, nfeat, fast_th, mode):
    import torch
    B, H, W = frames.shape
    dev = torch.device("cuda", 0)
    ex = uvo.ORBextractor(nfeat, 1.2, 8, 0, fast_th, max_width=W, max_height=H, max_batch=B)
    ex.tune(uvo.UVO_TUNE_FAST_MODE, mode)
    ex.set_pipeline(2)
    cap = ex.cap
    d_img = torch.from_numpy(frames).to(dev)
    outs = [(torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
             torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(2)]
    torch.cuda.synchronize()
    states = []
    for k in range(4):
        o = outs[k % 2]
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), cap)
        if k >= 2:
            states.append(ex.fast_state())
    ex.synchronize()
    res = [dict(n=o[2].cpu().numpy(), kp=o[0].cpu().numpy(), de=o[1].cpu().numpy()) for o in outs]
    ex.close()
    return res, states


@pytest.mark.parametrize("mode", ["adaptive", "two_pass"])
def test_low_contrast_batch_256_every_frame_vs_oracle(uvo, oracle, synth, mode):
    """Batch 256 of frames whose contrast is cut to a fraction (every cell of the upper levels and most of the lower ones falls back to
    the literal 7, src/ORBextractor.cc:797): the sparse per-cell pass at scale when forced, and the adaptive mode, which must have
    moved the levels that fall back to the single pass by its third batch on each lane."""
    frames = synth.make_sequence(0, 256, 640, 512)
    frames = (frames.astype(np.float32) * 0.12 + 100).astype(np.uint8)
    res, states = _run_bench_path_mode(uvo, frames, 1000, 20, uvo.UVO_FAST_MODE_ADAPTIVE if mode == "adaptive" else uvo.UVO_FAST_MODE_TWO_PASS)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    ref = [oe(frames[b]) for b in range(len(frames))]
    for li, r in enumerate(res):
        for b in range(len(frames)):
            n = int(r["n"][b])
            kp_g = np.ascontiguousarray(r["kp"][b, :n]).view(uvo.KEYPOINT_DTYPE).reshape(-1)
            _same(kp_g, r["de"][b, :n], ref[b][0], ref[b][1], "low contrast %s lane %d frame %d" % (mode, li, b))
    for t, fb, cells in states:
        assert fb.sum() * 2 > cells.sum() * 256, "these frames should leave most cells empty at fastTh: %s of %s x 256" % (fb, cells)
        for l in range(8):
            if mode == "two_pass":
                assert t[l] == 20
            elif fb[l] * 100 > cells[l] * 256 * 22:
                assert t[l] == 7, (l, fb[l], cells[l], t[l])


def test_configs3_hd_share_of_one_gpu_every_frame_vs_oracle(uvo, oracle, synth):
    """BASELINE.json configs[3]: 1920x1080 @ 2000 features, 128 frames per GPU (two quad-tree roots per level, 6594 FAST cells)."""
    frames = synth.make_sequence(0, 128, 1920, 1080, n_shapes=2500)
    res = _run_bench_path(uvo, frames, 2000, 20, passes=2)
    _check_against_oracle(uvo, oracle, frames, 2000, 20, res, "configs[3] batch 128", 2000)


@pytest.mark.parametrize("shape,nfeat,batch", [((512, 640), 1000, 40), ((1080, 1920), 2000, 6), ((480, 752), 1000, 3)])
def test_both_quad_tree_launch_shapes_equal_the_oracle(uvo, oracle, synth, shape, nfeat, batch):
    """The same batch through k_octree<1024> (UVO_TUNE_OCT_WIDE_MAX = huge) and k_octree<256> (= 0): both must be the oracle's
    DistributeOctTree.  (By default batch 40 x 8 levels = 320 problems takes the 256-thread form, batch <= 32 the 1024-thread one.)"""
    H, W = shape
    frames = synth.make_sequence(64, batch, W, H, n_shapes=400 if W < 1000 else 2500)
    ref = None
    for tune in (0, 1 << 30):
        res = _run_bench_path(uvo, frames, nfeat, 20, passes=2, tune=tune)
        ref = _check_against_oracle(uvo, oracle, frames, nfeat, 20, res, "wide_max=%d" % tune, nfeat)
    assert ref is not None


def test_host_entry_points_after_set_pipeline_2(uvo, oracle, synth):
    """uvo_extract / uvo_extract_batch / uvo_clahe -> uvo_extract(img = NULL) on a handle with two lanes: uploads, kernels and
    downloads of a call must share one lane's stream (the synchronous forms alternate lanes like the asynchronous ones)."""
    W, H, B = 640, 512, 5
    frames = synth.make_sequence(200, B, W, H)
    oe = oracle.extractor(1000, 1.2, 8, 20)
    ref = [oe(f) for f in frames]
    ex = uvo.ORBextractor(1000, 1.2, 8, 0, 20, max_width=W, max_height=H, max_batch=B, max_input_keypoints=64)
    ex.set_pipeline(2)
    for rep in range(3):                       # consecutive calls land on alternating lanes
        for b in range(B):
            kp, de = ex(frames[b])
            _same(kp, de, ref[b][0], ref[b][1], "uvo_extract rep %d frame %d" % (rep, b))
        for b, (kp, de) in enumerate(ex.extract_batch(frames)):
            _same(kp, de, ref[b][0], ref[b][1], "uvo_extract_batch rep %d frame %d" % (rep, b))
    # top-up mode (uploads of grid / counts / caller keypoints ride the same lane)
    rng = np.random.default_rng(5)
    kin = np.zeros(40, uvo.KEYPOINT_DTYPE)
    kin["x"], kin["y"] = rng.uniform(20, W - 21, 40).astype(np.float32), rng.uniform(20, H - 21, 40).astype(np.float32)
    kin["size"], kin["angle"], kin["octave"], kin["class_id"] = 31, -1, 0, -1
    rows, cols = H // 20 + 2, W // 20 + 2
    for rep in range(3):
        g1, g2 = np.zeros((rows, cols), np.int32, order="F"), np.zeros((rows, cols), np.int32, order="F")
        kp_g, de_g = ex(frames[rep], kin.copy(), g1, 20, False, 300)
        kp_o, de_o = oe(frames[rep], kin.copy(), g2, 20, False, 300)
        _same(kp_g, de_g, kp_o, de_o, "top-up rep %d" % rep)
        np.testing.assert_array_equal(g1, g2)
    # CLAHE result kept in HBM, then extraction from it
    for rep in range(3):
        enh = ex.clahe(frames[rep])
        np.testing.assert_array_equal(enh, oracle.clahe(frames[rep], 4.0, (12, 12)))
        kp_g, de_g = ex(None)
        kp_o, de_o = oe(enh)
        _same(kp_g, de_g, kp_o, de_o, "clahe -> extract(NULL) rep %d" % rep)
    ex.close()


def test_topup_grid_smaller_than_the_image_is_rejected(uvo, synth):
    """src/ORBextractor.cc:884-891 indexes grid_2d(int(y / d), int(x / d)); the call site sizes it rows / d + 2 (src/Tracking.cc:930-934).
    A smaller grid is a caller error here (Eigen would assert), never an out-of-bounds write."""
    W, H = 320, 256
    img = synth.make_frame(3, W, H, n_shapes=100)
    ex = uvo.ORBextractor(300, 1.2, 4, 0, 20, max_width=W, max_height=H, max_input_keypoints=8)
    ok = np.zeros((H // 20 + 2, W // 20 + 2), np.int32, order="F")
    ex(img, None, ok, 20, False, 100)
    for shape in [((H - 1) // 20, W // 20 + 2), (H // 20 + 2, (W - 1) // 20), (1, 1)]:
        with pytest.raises(uvo.UvoError) as ei:
            ex(img, None, np.zeros(shape, np.int32, order="F"), 20, False, 100)
        assert ei.value.code == uvo.UVO_E_BADARG
    exact = np.zeros(((H - 1) // 20 + 1, (W - 1) // 20 + 1), np.int32, order="F")   # the smallest grid every pixel position fits
    ex(img, None, exact, 20, False, 100)
    ex.close()


def test_extract_tracked_builds_the_callers_occupancy_grid_on_the_device(uvo, oracle, synth):
    """src/Tracking.cc:896-946 as one call: grid_2d((int)(pt.y / d), (int)(pt.x / d))++ for the tracked keypoints, then the top-up
    extraction -- against the oracle fed with the host-built grid, grid mutation included; also after uvo_clahe (image kept in HBM)."""
    W, H, d = 640, 512, 20
    img = synth.make_frame(4711, W, H)
    oe = oracle.extractor(400, 1.2, 8, 20)
    ex = uvo.ORBextractor(400, 1.2, 8, 0, 20, max_width=W, max_height=H, max_input_keypoints=800)
    rng = np.random.default_rng(9)
    for n_in in (330, 0, 800):
        kin = np.zeros(n_in, uvo.KEYPOINT_DTYPE)
        kin["x"], kin["y"] = rng.uniform(20, W - 21, n_in).astype(np.float32), rng.uniform(20, H - 21, n_in).astype(np.float32)
        kin["x"][: n_in // 4] = np.floor(kin["x"][: n_in // 4] / d) * d          # some exactly on cell boundaries
        kin["size"], kin["angle"], kin["octave"], kin["class_id"] = 31, -1, 0, -1
        grid = np.zeros((H // d + 2, W // d + 2), np.int32, order="F")
        for k in kin:
            grid[int(np.float32(k["y"]) / np.float32(d)), int(np.float32(k["x"]) / np.float32(d))] += 1
        need = 400 - n_in // 2
        # the reference calls the extractor with an EMPTY keypoint vector (pts0_ext, src/Tracking.cc:943-946) and the grid the tracked
        # points filled: only the new points come back; the oracle's grid ends as the reference's would (mutated by the accepted points)
        grid_o = grid.copy(order="F")
        kp_o, de_o = oe(img, None, grid_o, d, False, need)
        kp_g, de_g, grid_g = ex.extract_tracked(img, kin, d, need, want_grid=True)
        _same(kp_g, de_g, kp_o, de_o, "extract_tracked n_in=%d" % n_in)
        np.testing.assert_array_equal(grid_g, grid_o)
        assert n_in == 0 or not (set(zip(kp_g["x"].tolist(), kp_g["y"].tolist())) & set(zip(kin["x"].tolist(), kin["y"].tolist()))), \
            "tracked points must not come back as new ones"
    # KLT-tracked points a fraction of a cell outside the image: the reference's grid has two cells of slack and (int)(pt / d) truncates
    # towards zero, so they mark a cell like any other (src/Tracking.cc:901-907); only what the reference would index outside its grid
    # is refused
    kin = np.zeros(6, uvo.KEYPOINT_DTYPE)
    kin["x"] = np.float32([-0.4, -19.0, W + 3.5, 100.0, 200.0, W + 2 * d - 0.5])
    kin["y"] = np.float32([50.0, 60.0, 70.0, -7.25, H + 11.0, (H // d + 2) * d - 0.5])
    kin["size"], kin["angle"], kin["octave"], kin["class_id"] = 31, -1, 0, -1
    grid = np.zeros((H // d + 2, W // d + 2), np.int32, order="F")
    for k in kin:
        grid[int(np.float32(k["y"]) / np.float32(d)), int(np.float32(k["x"]) / np.float32(d))] += 1
    grid_o = grid.copy(order="F")
    kp_o, de_o = oe(img, None, grid_o, d, False, 300)
    kp_g, de_g, grid_g = ex.extract_tracked(img, kin, d, 300, want_grid=True)
    _same(kp_g, de_g, kp_o, de_o, "extract_tracked with points outside the image")
    np.testing.assert_array_equal(grid_g, grid_o)
    for bad in ((-20.5, 50.0), (W + 2 * d + 0.5, 50.0), (50.0, -d - 1.0), (50.0, float("nan"))):
        kin["x"][0], kin["y"][0] = bad
        with pytest.raises(uvo.UvoError):
            ex.extract_tracked(img, kin, d, 300)
    enh = ex.clahe(img, download=False)
    rng = np.random.default_rng(10)
    kin = np.zeros(300, uvo.KEYPOINT_DTYPE)
    kin["x"], kin["y"] = rng.uniform(20, W - 21, 300).astype(np.float32), rng.uniform(20, H - 21, 300).astype(np.float32)
    kin["size"], kin["angle"], kin["octave"], kin["class_id"] = 31, -1, 0, -1
    grid = np.zeros((H // d + 2, W // d + 2), np.int32, order="F")
    for k in kin:
        grid[int(k["y"] / d), int(k["x"] / d)] += 1
    kp_o, de_o = oe(oracle.clahe(img, 4.0, (12, 12)), None, grid, d, False, 150)
    kp_g, de_g = ex.extract_tracked(None, kin, d, 150)
    _same(kp_g, de_g, kp_o, de_o, "clahe -> extract_tracked(NULL)")
    ex.close()


def test_attached_handles_may_die_in_either_order(uvo, oracle):
    """uvo_matcher_attach_extractor: a destroyed extractor hands its matchers back to their own streams (they used to keep a dead stream),
    a destroyed matcher leaves the extractor's list (the extractor used to move a dead handle along)."""
    rng = np.random.default_rng(3)
    a, b = rng.integers(0, 256, (50, 32), dtype=np.uint8), rng.integers(0, 256, (60, 32), dtype=np.uint8)
    want = oracle.knn2(a, b)
    img = rng.integers(0, 256, (256, 320)).astype(np.uint8)
    ex = uvo.ORBextractor(300, 1.2, 4, 0, 20, max_width=320, max_height=256)
    ex.set_pipeline(2)
    m1, m2 = uvo.ORBmatcher(0.8), uvo.ORBmatcher(0.8)
    m1.attach(ex), m2.attach(ex)
    ex(img)
    np.testing.assert_array_equal(m1.knn2(a, b)[0], want[0])
    m2.close()                      # leaves the list ...
    ex(img), ex(img)                # ... so the lane changes move only m1
    np.testing.assert_array_equal(m1.knn2(a, b)[0], want[0])
    ex.close()                      # m1 is back on its own stream
    np.testing.assert_array_equal(m1.knn2(a, b)[0], want[0])
    m1.synchronize()
    m1.close()


def test_adaptive_fast_mode_follows_a_stream_that_changes_contrast(uvo, oracle, synth):
    """The adaptive FAST form of a level (`FAST(cell, fastTh)` then `FAST(cell, 7)` for empty cells, src/ORBextractor.cc:792-799) is chosen on
    the device from the PREVIOUS batch of the same pipeline lane: > 22 % fall-back cells -> one pass at 7, < 14 % -> two passes.  A stream
    that alternates textured and low-contrast batches therefore runs every batch in the form its predecessor asked for -- one batch of lag
    per lane -- and the keypoints must not notice.  One lane: the form flips after every batch (the recorded thresholds alternate 7 / fastTh).
    Two lanes: lane 0 only ever sees textured batches, lane 1 low-contrast ones, so each settles in its own form after its first batch."""
    w, h, B, TH = 320, 256, 24, 20
    tex = synth.make_batch(B, w, h, seed0=8800)
    low = [(f.astype(np.float32) * 0.12 + 110 * 0.88).astype(np.uint8) for f in synth.make_batch(B, w, h, seed0=8900)]
    oe = oracle.extractor(400, 1.2, 5, TH)
    ref = {"tex": [oe(f) for f in (tex[0], tex[B // 2], tex[-1])], "low": [oe(f) for f in (low[0], low[B // 2], low[-1])]}

    def check(res, kind):
        for (kp_o, de_o), f in zip(ref[kind], (0, B // 2, B - 1)):
            kp, de = res[f]
            assert kp.tobytes() == kp_o.tobytes() and (de == de_o).all(), (kind, f)

    for depth in (1, 2):
        ex = uvo.ORBextractor(400, 1.2, 5, 0, TH, max_width=w, max_height=h, max_batch=B)
        ex.set_pipeline(depth)
        seen = []
        for i in range(8):
            kind = "tex" if i % 2 == 0 else "low"
            res = ex.extract_batch(tex if kind == "tex" else low)
            check(res, kind)
            t, fb, cells = ex.fast_state()          # thresholds the NEXT batch of this lane streams at; fall-back cells of this batch
            share = fb.sum() / float(cells.sum() * B)
            assert (share < 0.10) if kind == "tex" else (share > 0.5), (kind, share)
            seen.append(int(t[0]))
        if depth == 1:
            # every batch decides for the next one: after a textured batch two passes (fastTh), after a low-contrast one a single pass (7)
            assert seen == [TH, 7] * 4, seen
        else:
            # lane 0 = batches 0, 2, 4, 6 (textured), lane 1 = batches 1, 3, 5, 7 (low contrast): no flip-flop inside a lane
            assert seen[0::2] == [TH] * 4 and seen[1::2] == [7] * 4, seen
        ex.close()
