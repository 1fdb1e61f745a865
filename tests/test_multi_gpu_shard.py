"""The N>1 path on CPU: the shard plan (pure arithmetic behind the C ABI), the halo arithmetic of bench.py, and -- with two gloo
ranks -- the gather itself: every rank writes its block at the plan's offsets of ONE host region shared between the processes (the
same SharedHostRegion bench.py registers with HIP on a GPU box) and rank 0 finds the whole job there; the timed region reports the
MAX over ranks."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_plan_tiles_frames_and_pairs(uvo):
    """Blocks are contiguous, disjoint and cover the job; the pairs (p, p + 1) are owned exactly once; a block's halo is the
    neighbouring block's first frame."""
    for total in [1, 2, 7, 128, 1000, 1024, 1025]:
        for n in [1, 2, 3, 4, 8, 16]:
            frames, pairs, plans = [], [], [uvo.shard_plan(total, n, s, 128) for s in range(n)]
            for s, p in enumerate(plans):
                frames += list(range(p.first_frame, p.first_frame + p.n_frames))
                pairs += list(range(p.first_pair, p.first_pair + p.n_pairs))
                assert p.n_chunks == (p.n_frames + 127) // 128
                nxt = [q for q in plans[s + 1:] if q.n_frames > 0]
                if p.n_frames > 0 and nxt:
                    assert p.halo_frame == nxt[0].first_frame == p.first_frame + p.n_frames
                else:
                    assert p.halo_frame == -1
                assert abs(p.n_frames - total / n) < 1
            assert frames == list(range(total))
            assert pairs == list(range(total - 1))
    # BASELINE.json configs[3]: 1024 frames over 8 GPUs, 128 each, one chunk per GPU
    for s in range(8):
        p = uvo.shard_plan(1024, 8, s, 128)
        assert (p.first_frame, p.n_frames, p.n_chunks) == (128 * s, 128, 1)
        assert p.halo_frame == (128 * (s + 1) if s < 7 else -1) and p.n_pairs == (128 if s < 7 else 127)
    with __import__("pytest").raises(uvo.UvoError):
        uvo.shard_plan(10, 0, 0, 1)
    with __import__("pytest").raises(uvo.UvoError):
        uvo.shard_plan(10, 2, 2, 1)


def test_bench_halo_is_the_neighbours_first_frame(synth):
    sys.path.insert(0, ROOT)
    import bench
    for world, B in [(1, 256), (2, 256), (8, 128), (8, 256)]:
        firsts = [bench.shard_frames(r, world, B)[0] for r in range(world)]
        for r in range(world):
            first, halo = bench.shard_frames(r, world, B)
            assert first == r * B and halo == firsts[(r + 1) % world]
    # a shard of the global sequence generated on its own equals the same frames generated as part of the whole sequence, and the
    # halo frame a rank generates is bit-identical to its owner's frame
    whole = synth.make_sequence(0, 40, 96, 64, chain=8, n_shapes=20)
    part = synth.make_sequence(21, 12, 96, 64, chain=8, n_shapes=20)
    np.testing.assert_array_equal(whole[21:33], part)
    np.testing.assert_array_equal(synth.make_sequence(32, 1, 96, 64, chain=8, n_shapes=20)[0], whole[32])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    import bench
    uvo = importlib.import_module("u-vip-slam_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))   # rank 1 is the slow one

    dt = bench.timed_steps(step, lambda: None, 5, dist, None)
    # the gather: one region, every rank writes its plan's block (a function of the GLOBAL frame index stands in for the device copies)
    total, cap = 37, 5
    region = bench.SharedHostRegion(uvo, total * cap * 4 + total * 4 + 1024, rank, world, dist)
    rows, off = region.carve(0, (total, cap), np.int32)
    counts, off = region.carve(off, (total,), np.int32)
    p = uvo.shard_plan(total, world, rank, 8)
    for f in range(p.first_frame, p.first_frame + p.n_frames):
        rows[f] = f * 100 + np.arange(cap)
        counts[f] = f + 1
    dist.barrier()
    ok = True
    if rank == 0:
        ok = (counts == np.arange(total) + 1).all() and (rows == np.arange(total)[:, None] * 100 + np.arange(cap)[None, :]).all()
    region.close(rank, dist)
    q.put((rank, len(calls), dt, bool(ok), p.first_frame, p.n_frames, region.path))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_into_one_region_and_max_time():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, t0, ok0, f0, n0, path0), (r1, c1, t1, ok1, f1, n1, path1) = res
    assert (r0, r1) == (0, 1)
    assert c0 == c1 == 5                                                            # exactly K steps each
    assert abs(t0 - t1) < 1e-9 and t0 >= 5 * 0.02 * 0.95                            # both report the slow rank's time
    assert (f0, n0, f1, n1) == (0, 19, 19, 18)                                      # contiguous, disjoint blocks
    assert ok0 and path0 == path1 and not os.path.exists(path0)                     # rank 0 saw both blocks; the mapping is gone
