"""The N-rank code path of bench.py on one GPU: what the driver's SCALE run executes (one process per GPU under
torch.distributed.run, rank r owning frames [rB, (r+1)B) of one global sequence + the neighbour's first frame as halo, every rank's
results gathered into ONE /dev/shm region that each process page-locks, barrier + max-over-ranks timing) -- here with both ranks on
device 0 and gloo carrying the barrier (UVO_BENCH_DRYRUN_ONE_GPU=1: RCCL refuses two ranks on one device).  The launcher is a fresh
child process, started before anything in it touches the GPU.  Call site served: src/Tracking.cc:946, sharded per SURVEY.md 8(e)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_gather_into_one_shared_region():
    env = dict(os.environ, UVO_BENCH_DRYRUN_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29583",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "24", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-subrecords", "--h2h"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "bench.py --gpus 2 failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0: %r" % r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["batch_per_gpu"] == 24
    assert d["value"] > 0 and abs(d["value"] - 2 * 24 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-3 * d["value"]   # whole-job frames / max-over-ranks time
    assert d["verified_frames"] > 0, "rank 0 re-checks its timed buffers against the oracle"
    h = d["host_to_host"]
    assert h is not None and h["gathered_equals_hbm_resident"] is True and h["frames_per_job"] == 48
    assert "/dev/shm" in h["gather"], h["gather"]
    assert "cpu_baseline" not in d


def test_two_ranks_run_the_configs3_record_with_its_host_gather():
    """configs[3] (1920x1080 @ 2000 feats, frames sharded per GPU, host gather) as bench.py --gpus N runs it behind the headline record:
    every rank extracts + matches its share HBM-resident, then the sharder gathers world x batch frames into the one shared region
    (here 2 ranks x 6 frames on device 0)."""
    env = dict(os.environ, UVO_BENCH_DRYRUN_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", UVO_BENCH_C3_BATCH="6")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29585",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-subrecords", "--c3"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "bench.py --gpus 2 --c3 failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c3 = d["sub_records"]["configs[3] over 2 GPUs"]
    assert c3["n_gpus"] == 2 and c3["value"] > 0 and c3["verified_frames"] > 0
    assert "1920x1080" in c3["workload"]
    h = c3["host_to_host"]
    assert h["gathered_equals_hbm_resident"] is True and h["frames_per_job"] == 12 and "/dev/shm" in h["gather"]
    assert len(h["numa"]) == 2 and all("link_GBps" in r_ and r_["h2h_frac"] > 0 for r_ in h["numa"])
    for sp in (d["step_spread"], c3["step_spread"]):   # (None when the dominant kernel of so small a batch is one that is launched per level)
        assert sp is None or 0 < sp["ms_min"] <= sp["ms_median"] <= sp["ms_max"]
    assert d["step_spread"] is not None or d["roofline"]["kernel"] == "k_resize_level"
