"""uvo_sharder: N handles, one host thread per shard, results gathered by the device-to-host copies at precomputed offsets of one
page-locked region, pairs across block edges matched through a 1-frame halo from the neighbouring block (the call
site this batches is src/Tracking.cc:946).  A one-GPU box runs the shards as several handles on device 0 from several threads."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, NFEAT = 320, 256, 400


def _reference(uvo, oracle, frames):
    """Single-handle results of every frame + the oracle's knn-2 rows of every consecutive pair."""
    ex = uvo.ORBextractor(NFEAT, 1.2, 6, 0, 20, max_width=W, max_height=H, max_batch=len(frames))
    ref = ex.extract_batch(frames)
    ex.close()
    oe = oracle.extractor(NFEAT, 1.2, 6, 20)
    for b in (0, len(frames) // 2, len(frames) - 1):
        kp_o, de_o = oe(frames[b])
        assert ref[b][0].tobytes() == kp_o.tobytes() and (ref[b][1] == de_o).all()
    rows = [oracle.knn2(ref[p][1], ref[p + 1][1]) for p in range(len(frames) - 1)]
    return ref, rows


def _alloc(uvo, total, cap):
    return (uvo.pinned_empty((total, cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((total, cap, 32), np.uint8), uvo.pinned_empty((total,), np.int32),
            uvo.pinned_empty((total - 1, cap), np.int32), uvo.pinned_empty((total - 1, cap), np.uint16), uvo.pinned_empty((total - 1, cap), np.int32),
            uvo.pinned_empty((total - 1, cap), np.uint16))


def _check(ref, rows, kp, de, n, i0, d0, i1, d1, what):
    total = len(ref)
    for f in range(total):
        assert n[f] == len(ref[f][0]), "%s frame %d: %d keypoints, expected %d" % (what, f, n[f], len(ref[f][0]))
        assert kp[f, :n[f]].tobytes() == ref[f][0].tobytes(), "%s frame %d keypoints" % (what, f)
        assert (de[f, :n[f]] == ref[f][1]).all(), "%s frame %d descriptors" % (what, f)
    for p in range(total - 1):
        nq = n[p]
        np.testing.assert_array_equal(i0[p, :nq], rows[p][0], err_msg="%s pair %d" % (what, p))
        np.testing.assert_array_equal(d0[p, :nq].astype(np.int32), rows[p][1], err_msg="%s pair %d" % (what, p))
        np.testing.assert_array_equal(i1[p, :nq], rows[p][2], err_msg="%s pair %d" % (what, p))
        np.testing.assert_array_equal(d1[p, :nq].astype(np.int32), rows[p][3], err_msg="%s pair %d" % (what, p))


@pytest.fixture(scope="module")
def job(uvo, oracle, synth):
    frames = synth.make_sequence(0, 53, W, H, chain=16, n_shapes=120)
    pinned = uvo.pinned_empty(frames.shape, np.uint8)
    pinned[:] = frames
    return pinned, _reference(uvo, oracle, frames)


@pytest.mark.parametrize("n_shards,chunk", [(1, 64), (2, 8), (3, 5), (4, 53)])
def test_shards_on_one_device_gather_the_single_handle_result(uvo, job, n_shards, chunk):
    """n handles on device 0 driven from n host threads; several chunks per shard (two in flight); every frame and every pair --
    the ones across chunk and shard edges included -- byte for byte what one handle / the oracle produce."""
    frames, (ref, rows) = job
    total = len(frames)
    sh = uvo.Sharder(NFEAT, 1.2, 6, 20, max_width=W, max_height=H, devices=[0] * n_shards, chunk_frames=chunk, match=True)
    out = _alloc(uvo, total, sh.cap)
    for a in out:
        a.view(np.uint8)[...] = 0xEE
    for rep in range(2):                     # a second run reuses the lanes' staging
        sh.run(frames, 0, total, *out)
        _check(ref, rows, *out, "shards=%d chunk=%d rep %d" % (n_shards, chunk, rep))
    sh.close()


def test_jobs_stream_through_the_lanes_without_draining(uvo, job):
    """uvo_sharder_submit / _wait: three jobs queued back to back (different output arrays, different frame ranges of the same
    sequence), waited for in order; a ticket cannot be waited for twice."""
    frames, (ref, rows) = job
    total = len(frames)
    sh = uvo.Sharder(NFEAT, 1.2, 6, 20, max_width=W, max_height=H, devices=[0, 0], chunk_frames=6, match=True)
    spans = [(0, total), (0, 31), (0, 17)]          # a job is always frames [0, n) of what `imgs` holds
    outs = [_alloc(uvo, n, sh.cap) for _, n in spans]
    for o in outs:
        for a in o:
            a.view(np.uint8)[...] = 0xEE
    tickets = [sh.submit(frames, 0, n, *o) for (_, n), o in zip(spans, outs)]
    assert len(set(tickets)) == 3
    for (_, n), o, t in zip(spans, outs, tickets):
        sh.wait(t)
        _check(ref[:n], rows[:n - 1], *o, "streamed job of %d frames" % n)
    with pytest.raises(uvo.UvoError):
        sh.wait(tickets[0])
    # a stack that ends before a local shard's last frame (or its halo frame) is refused instead of read past its end
    with pytest.raises(uvo.UvoError):
        sh.submit(frames[:20], 0, 31, *outs[1])
    with pytest.raises(uvo.UvoError):
        sh.submit(frames[:total - 1], 0, total, *outs[0])
    sh.run(frames, 0, 17, *outs[2])               # and the handle still works afterwards
    _check(ref[:17], rows[:16], *outs[2], "after the refused jobs")
    sh.close()


def test_one_process_per_shard_form_fills_the_same_region(uvo, job):
    """The multi-process form: every sharder owns one shard (the others are UVO_SHARD_REMOTE), all write into the same arrays, and a
    process holds only its own frames + its block's halo frame."""
    frames, (ref, rows) = job
    total, n_shards = len(frames), 3
    cap = None
    out = None
    for s in range(n_shards):
        devices = [uvo.UVO_SHARD_REMOTE] * n_shards
        devices[s] = 0
        sh = uvo.Sharder(NFEAT, 1.2, 6, 20, max_width=W, max_height=H, devices=devices, chunk_frames=7, match=True)
        if out is None:
            cap = sh.cap
            out = _alloc(uvo, total, cap)
            for a in out:
                a.view(np.uint8)[...] = 0xEE
        p = sh.plan(total, s)
        last = p.first_frame + p.n_frames + (1 if p.halo_frame >= 0 else 0)
        mine = uvo.pinned_empty((last - p.first_frame, H, W), np.uint8)
        mine[:] = frames[p.first_frame:last]
        sh.run(mine, p.first_frame, total, *out)
        sh.close()
    _check(ref, rows, *out, "one sharder per shard")


def test_extraction_only_and_bad_arguments(uvo, job):
    frames, (ref, rows) = job
    total = len(frames)
    sh = uvo.Sharder(NFEAT, 1.2, 6, 20, max_width=W, max_height=H, devices=[0, 0], chunk_frames=16, match=False)
    kp, de, n = uvo.pinned_empty((total, sh.cap), uvo.KEYPOINT_DTYPE), uvo.pinned_empty((total, sh.cap, 32), np.uint8), uvo.pinned_empty((total,), np.int32)
    sh.run(frames, 0, total, kp, de, n)
    for f in range(total):
        assert kp[f, :n[f]].tobytes() == ref[f][0].tobytes() and (de[f, :n[f]] == ref[f][1]).all()
    i0, d0 = np.zeros((total - 1, sh.cap), np.int32), np.zeros((total - 1, sh.cap), np.uint16)
    with pytest.raises(uvo.UvoError) as ei:          # match outputs on a sharder created without matching
        sh.run(frames, 0, total, kp, de, n, i0, d0, i0.copy(), d0.copy())
    assert ei.value.code == uvo.UVO_E_BADARG
    small = np.zeros((total, sh.cap - 1), uvo.KEYPOINT_DTYPE)
    with pytest.raises(uvo.UvoError) as ei:
        sh.run(frames, 0, total, small, np.zeros((total, sh.cap - 1, 32), np.uint8), n)
    assert ei.value.code == uvo.UVO_E_CAPACITY
    with pytest.raises(uvo.UvoError) as ei:          # imgs starts after the first local frame
        sh.run(frames, 5, total, kp, de, n)
    assert ei.value.code == uvo.UVO_E_BADARG
    sh.close()
    with pytest.raises(uvo.UvoError) as ei:
        uvo.Sharder(NFEAT, 1.2, 6, 20, max_width=W, max_height=H, devices=[uvo.UVO_SHARD_REMOTE], chunk_frames=4)
    assert ei.value.code == uvo.UVO_E_BADARG


def test_thread_binds_next_to_the_gpu(uvo):
    """uvo_host_bind_near_device on a real device: the calling thread ends up on CPUs of the device's local_cpulist (where sysfs
    has one and the process may use them) and the NUMA node reported is the device's; a no-op otherwise.  In a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import importlib, os, sys, glob
sys.path.insert(0, %r)
uvo = importlib.import_module("u-vip-slam_amd")
before = set(os.sched_getaffinity(0))
bound, node = uvo.host_bind_near_device(0)
after = set(os.sched_getaffinity(0))
print(bound, node, len(before), len(after))
assert after <= before and len(after) > 0
if not bound:
    assert after == before
else:
    lists = [open(p).read().strip() for p in glob.glob("/sys/bus/pci/devices/*/local_cpulist")]
    def cpus(txt):
        out = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            out |= set(range(int(a), int(b or a) + 1))
        return out
    assert any(after <= cpus(t) for t in lists if t), "the mask is no subset of any device's local_cpulist"
os.environ["UVO_NUMA_BIND"] = "0"
""" % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    print("bind_near_device:", out.stdout.strip().splitlines()[-1])
