"""N>1 path of bench.py on CPU: two gloo ranks shard the frame sequence disjointly (no data-path collective) and the
timed region reports the MAX over ranks."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seed0 = bench.shard_seed0(rank, 8)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))   # rank 1 is the slow one

    dt = bench.timed_steps(step, lambda: None, 5, dist, None)
    q.put((rank, seed0, len(calls), dt))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_max_time():
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
    (r0, s0, c0, t0), (r1, s1, c1, t1) = res
    assert (r0, r1) == (0, 1)
    assert set(range(s0, s0 + 8)).isdisjoint(range(s1, s1 + 8)) and s1 == s0 + 8   # contiguous, disjoint shards
    assert c0 == c1 == 5                                                            # exactly K steps each
    assert abs(t0 - t1) < 1e-9 and t0 >= 5 * 0.02 * 0.95                            # both report the slow rank's time
