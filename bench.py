#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one rank's batch of synthetic frames, all inputs already resident in HBM:
  uvo_extract_batch_device       (pyramid -> per-cell FAST -> quad-tree -> IC angle -> blur -> rBRIEF; B frames + 1 halo frame)
  uvo_hamming_knn2_batch_device  (frame i vs frame i+1, all-pairs 256-bit Hamming knn-2, B pairs)
Workloads (--config): 2 = BASELINE.json configs[2], the configuration the metric is quoted on (batch 256 synthetic 640x512 mono
frames per GPU, 1000 features, 8 levels, fastTh 20; default); 3 = configs[3] (1920x1080 @ 2000 features, 128 frames per GPU).
With N GPUs (one process per GPU) rank r owns frames [r*B, (r+1)*B) of ONE global synthetic sequence and its matching pair B-1 uses
the NEIGHBOURING rank's first frame, recomputed locally as a 1-frame halo: weak scaling, no data-path collective
(torch.distributed / RCCL carries the timing barrier and the max-over-ranks only).

`value` is the HBM-resident rate; consecutive steps read four DISTINCT input batches in turn (the sequence and its three mirror
images: same statistics, 4 x 84 MB, more than the 256 MB Infinity Cache keeps).  The same line carries: `roofline` = the HBM
roofline of the dominant kernel (algorithmic bytes per launch / its live launch duration, HIP events on the library's stream inside
the timed region, against 8 TB/s; `frac` = `frac_live`, with `frac_alone` = the same launch at pipeline depth 1 and both recomputed from
the committed rocprofv3 files `roofline.rocprof_csv` names) with `roofline.per_kernel` for every kernel of the step; `step_spread` (the
step's period inside the timed region, min / median / max, from the events of the dominant kernel's launches); top-level `whole_path_frac` (the step's
algorithmic bytes x frames/s against 8 TB/s) and `valu_frac` (the dominant kernel's VALU issue rate against the chip's issue peak --
what actually bounds it); the host-to-host rate through uvo_sharder (uploads of page-locked frames + the gather of all ranks'
results into one page-locked host region, `host_to_host`, with every rank's measured link rates and `h2h_frac` = what the leg does
against what the link can do); sub-records for configs[1] (batch-1 latency), configs[3] (1920x1080 @ 2000 features: one GPU's share,
verified; with N ranks its real shape, N x 128 frames with the host gather), extract-only and configs[4] (fused frustum search); a measured device-copy bandwidth next to the 8 TB/s spec; `verified_frames` (outputs of the TIMED
buffers compared with the CPU oracle after the timed region; a mismatch fails the run; the oracle is this repo's restatement of the
reference -- parity unpinned) and `cpu_baseline` (the same oracle timed on this host's cores on bounded
samples).  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import mmap
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NLEVELS, FAST_TH, SCALE = 8, 20, 1.2
CONFIGS = {
    2: dict(name="BASELINE.json configs[2]", W=640, H=512, nfeat=1000, batch=256, n_shapes=400, steps=100),
    3: dict(name="BASELINE.json configs[3]", W=1920, H=1080, nfeat=2000, batch=128, n_shapes=2500, steps=30),
}
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the measured device-copy rate is reported beside it
PROFILE_TAG = "r06_final"      # the round checkpoint under profiles/ that holds this build's rocprofv3 summaries (tools/round_checkpoint.sh)
VALU_PEAK_GINSTR = 1228.8        # wave-instructions/s the chip can issue: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 op on a SIMD32


def level_sizes(w, h, nlevels=NLEVELS, sf=SCALE):
    inv = np.float32(1.0)
    step = np.float32(np.float32(1.0) / np.float64(np.float32(sf)))
    out = []
    for _ in range(nlevels):
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        inv = np.float32(inv * step)
    return out


def algorithmic_bytes_per_frame(w, h, k):
    """per-kernel algorithmic HBM bytes of one frame (each level materialised once, read once per
    consuming stage; per-keypoint stages read their footprint once)."""
    s = [a * b for a, b in level_sizes(w, h)]
    tot = sum(s)
    return {
        "k_pad_level0": 2 * s[0],                       # not in the survey's model (a design that reads the input in place needs none)
        "k_resize_level": sum(s[:-1]) + sum(s[1:]),     # reads S0..S6, writes S1..S7 (all 7 launches)
        "k_pyr_tiles": sum(s[:-1]) + sum(s[1:]),        # the same model for the level-group launches together (they re-read fewer levels than it counts)
        "k_fast_score": tot,                            # reads every level once
        "k_gauss7": 2 * tot,
        "k_octree_gauss": 2 * tot,                      # the quad-tree and the blur as one launch: the blur's bytes
        "k_fast_blur": 3 * tot,                         # the fused form: one read serves FAST and the blur, one write
        "k_fast_score_gauss": 3 * tot,                  # FAST and the blur as one launch (small batches): each reads the levels, the blur writes them
        "k_octree": 0,
        "k_assemble": 0,
        "k_describe": 749 * k + (512 + 32) * k + 20 * k,
        "k_knn2": 64 * k + 12 * k,
    }


def shard_frames(rank, world, batch):
    """Frames shard embarrassingly: rank r owns frames [r*batch, (r+1)*batch) of the global sequence; its halo is the neighbouring
    rank's first frame (the last rank wraps to frame 0 so that every rank does the same work).  -> (first_frame, halo_frame)"""
    return rank * batch, ((rank + 1) * batch) % (world * batch)


def timed_steps(step, sync, steps, dist=None, device=None):
    """Time exactly `steps` steps bracketed by barrier + full device sync on both sides; returns the MAX over ranks (s)."""
    import torch
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# ------------------------------------------------------------------------------------------------ CPU baseline (the checker, timed)
def _cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, len(phys) or (os.cpu_count() or 1), os.cpu_count() or 1


def _oracle(native=False):
    import oracle_lib
    if not native:
        return oracle_lib.Oracle()
    # -march=native must be built on the host it runs on
    subprocess.check_call(["make", "-s", "-C", oracle_lib.ORACLE_DIR, "liborb_oracle_native.so"])
    return oracle_lib.Oracle(os.path.join(oracle_lib.ORACLE_DIR, "liborb_oracle_native.so"))


def _time_extract_match(o, frames, nfeat, budget_s, match=True):
    oe = o.extractor(nfeat, SCALE, NLEVELS, FAST_TH)
    t0 = time.perf_counter()
    prev, n = None, 0
    while time.perf_counter() - t0 < budget_s:
        kp, de = oe(frames[n % len(frames)])
        if match and prev is not None and len(prev) and len(de):
            o.knn2(prev, de)
        prev = de
        n += 1
    return n, time.perf_counter() - t0


def _time_threads(o, frames, nfeat, budget_s, threads, match=True):
    """One frame per thread (ctypes releases the GIL inside the oracle; every thread owns its extractor)."""
    from concurrent.futures import ThreadPoolExecutor
    counts = [0] * threads
    t0 = time.perf_counter()

    def work(i):
        oe = o.extractor(nfeat, SCALE, NLEVELS, FAST_TH)
        prev, n = None, 0
        while time.perf_counter() - t0 < budget_s:
            kp, de = oe(frames[(i + n * threads) % len(frames)])
            if match and prev is not None and len(prev) and len(de):
                o.knn2(prev, de)
            prev = de
            n += 1
        counts[i] = n

    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(work, range(threads)))
    return sum(counts), time.perf_counter() - t0


def cpu_baseline(frames, cfg, c4=None):
    """The oracle (kind 'port') on this host.  Headline row = BASELINE.md CPU-1 (1 core, -O3 without -march=native, mirroring the
    reference's CMakeLists.txt:19-22); `rows` holds the other rows of BASELINE.md section 2, each on a bounded sample."""
    model, phys, logical = _cpu_info()
    nfeat = cfg["nfeat"]
    what = "%dx%d, %d feats, %d levels, fastTh %d" % (cfg["W"], cfg["H"], nfeat, NLEVELS, FAST_TH)
    o = _oracle()
    n1, dt1 = _time_extract_match(o, frames, nfeat, 8.0)
    rows = {"CPU-1": {"frames_per_s": round(n1 / dt1, 3), "cores": 1, "flags": "-O3", "what": "extract + knn-2 of consecutive frames, " + what, "frames": n1}}
    try:
        on = _oracle(native=True)
        n, dt = _time_extract_match(on, frames, nfeat, 4.0)
        rows["CPU-1n"] = {"frames_per_s": round(n / dt, 3), "cores": 1, "flags": "-O3 -march=native", "what": "same", "frames": n}
        n, dt = _time_threads(on, frames, nfeat, 5.0, phys)
        rows["CPU-N"] = {"frames_per_s": round(n / dt, 3), "cores": phys, "flags": "-O3 -march=native", "frames": n,
                         "what": "same, one frame per thread on all physical cores"}
        # CPU-HD: extraction of 1920x1080 frames at 2000 features, one frame per thread on all physical cores
        synth = importlib.import_module("u-vip-slam_amd.synth")
        hd = synth.make_sequence(0, 8, 1920, 1080, n_shapes=2500)
        n, dt = _time_threads(on, hd, 2000, 4.0, phys, match=False)
        rows["CPU-HD"] = {"frames_per_s": round(n / dt, 3), "cores": phys, "flags": "-O3 -march=native", "frames": n,
                          "what": "extract only, 1920x1080, 2000 feats, %d levels, fastTh %d, one frame per thread" % (NLEVELS, FAST_TH)}
    except (OSError, subprocess.CalledProcessError) as e:  # no compiler on the box: the rows are simply absent
        rows["CPU-1n"] = {"error": str(e)[:200]}
    # CPU-400 (footnote row): the harbor YAML as shipped -- 400 features, CLAHE pre-processing on (Data/Settings_VI_Aqualoc_harbor.yaml:67-79,
    # src/Tracking.cc:425-431) -- extract only, 1 core
    oe400 = o.extractor(400, SCALE, NLEVELS, FAST_TH)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0:
        oe400(o.clahe(frames[n % len(frames)], 4.0, (12, 12)))
        n += 1
    dt = time.perf_counter() - t0
    rows["CPU-400"] = {"frames_per_s": round(n / dt, 3), "cores": 1, "flags": "-O3", "frames": n,
                       "what": "CLAHE (clip 4, 12x12 tiles) + extract, %dx%d, 400 feats, %d levels, fastTh %d" % (cfg["W"], cfg["H"], NLEVELS, FAST_TH)}
    # CPU-M: the matcher alone (all-pairs knn-2 + ratio test of include/utils.h:81-111 on two consecutive frames' descriptors)
    oe = o.extractor(nfeat, SCALE, NLEVELS, FAST_TH)
    d0, d1 = oe(frames[0])[1], oe(frames[1])[1]
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0:
        i0, dd0, i1, dd1 = o.knn2(d0, d1)
        (dd0 <= 0.8 * dd1).sum()
        n += 1
    dt = time.perf_counter() - t0
    rows["CPU-M"] = {"pairs_per_s": round(n / dt, 2), "cores": 1, "flags": "-O3", "what": "all-pairs knn-2 + ratio 0.8, %d x %d descriptors" % (len(d0), len(d1))}
    if c4 is not None:  # CPU-P: isInFrustum + SearchByProjection(Frame, MapPoints, th) of configs[4]
        kp, de, sf, mp, cam_o = c4
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            ov, ou, ovv, ol, ovc = o.project_points(0, cam_o, mp["xyz"], mp["normal"], mp["min_distance"], mp["max_distance"], None, sf, 1.2, 0.5)
            a = np.full(len(kp), -1, np.int32)
            nm = o.search_by_projection(kp, de, (0, 0, 752, 480), a, ou, ovv, ol, ovc, ov, mp["mp_desc"], sf, 1.0, 0.8)
            n += 1
        dt = time.perf_counter() - t0
        rows["CPU-P"] = {"calls_per_s": round(n / dt, 2), "ms_per_call": round(dt / n * 1e3, 3), "cores": 1, "flags": "-O3", "matches": int(nm),
                         "what": "isInFrustum + SearchByProjection, 752x480 frame (%d kp) vs 5000 map points" % len(kp)}
    return {"value": rows["CPU-1"]["frames_per_s"], "unit": "frames/s", "cores": 1, "kind": "port",
            "note": "a scalar line-by-line port of the reference path (OpenCV's own cv::FAST / GaussianBlur / resize are SIMD and several times "
                    "faster): a reported baseline, not a speed-up claim over the reference",
            "sample": "%d frames drawn in order from this run's batch: oracle extract (%s) + knn-2 match of consecutive frames, 1 thread, g++ -O3 "
                      "without -march=native (BASELINE.md row CPU-1)" % (n1, what),
            "host": {"cpu_model": model, "physical_cores": phys, "logical_cpus": logical}, "rows": rows}


# ------------------------------------------------------------------------------------------------ helpers
class SharedHostRegion:
    """One host region every rank gathers into.  One rank: page-locked memory from uvo_host_alloc.  Several ranks (processes): one
    /dev/shm mapping, page-locked in every process with uvo_host_register, so each rank's device-to-host copies land at their final
    offsets of the SAME physical pages -- the gather of north_star, with no collective."""

    def __init__(self, uvo, nbytes, rank, world, dist):
        self.uvo, self.nbytes, self.registered, self.mm, self.path = uvo, nbytes, True, None, None
        if world == 1:
            self.buf = uvo.pinned_empty((nbytes,), np.uint8)
            return
        name = [None]
        if rank == 0:
            name[0] = "/dev/shm/uvo_bench_%d_%d" % (os.getpid(), int(time.time() * 1e3) % 100000)
            with open(name[0], "wb") as fh:
                fh.truncate(nbytes)
        dist.broadcast_object_list(name, src=0)
        self.path = name[0]
        fd = os.open(self.path, os.O_RDWR)
        self.mm = mmap.mmap(fd, nbytes)
        os.close(fd)
        self.buf = np.frombuffer(self.mm, dtype=np.uint8)
        self.registered = False   # register() follows the ranks' first touch of their own slices

    def register(self, own_slices, dist):
        """Several ranks: every rank writes its OWN slices of the mapping first (from its thread, which is bound to the CPUs next to its GPU:
        the pages land on that NUMA node), all ranks meet, and only then is the region page-locked -- registration faults in every page
        that is still untouched, on the node of whoever registers first."""
        if self.mm is None:
            return
        for a in own_slices:
            a[...] = 0
        if dist is not None:
            dist.barrier()
        try:
            self.uvo.host_register(self.buf)
            self.registered = True
        except self.uvo.UvoError:
            self.registered = False   # still correct: the copies just stop being asynchronous

    def carve(self, offset, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.buf[offset:offset + n].view(dtype).reshape(shape), (offset + n + 255) // 256 * 256

    def close(self, rank, dist):
        if self.mm is not None:
            if self.registered:
                self.uvo.host_unregister(self.buf)
            if dist is not None:
                dist.barrier()
            if rank == 0:
                os.unlink(self.path)


def device_copy_gbps(torch, dev):
    """Measured device-to-device copy bandwidth (read + write bytes / time) of one 1 GiB buffer: what a pure streaming kernel reaches."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.fill_(1)
    b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.copy_(a)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del a, b
    return 2.0 * n * 10 / dt / 1e9


def rocprof_avg_ms(csv_path, kernel):
    """AverageNs x calls-per-step of a kernel's row(s) in a committed `rocprofv3 --kernel-trace --stats` summary (the device kernels behind a
    profiler name: k_knn2 -> k_knn2_mfma, k_fast_cells -> + _list); -> (ms per launch averaged over the file's launches, launches) or None"""
    import csv
    try:
        rows = list(csv.DictReader(open(os.path.join(ROOT, csv_path))))
    except OSError:
        return None
    tot, calls = 0.0, 0
    for r in rows:
        nm = r["Name"].replace("void ", "").replace("uvo::", "").split("(")[0].split("<")[0]
        if nm == kernel or nm.startswith(kernel + "_"):
            tot += float(r["TotalDurationNs"])
            calls = max(calls, int(r["Calls"]))
    return (tot / calls / 1e6, calls) if calls else None


def load_pmc(config):
    """Per-kernel counters from the committed rocprofv3 PMC passes of this same command (separate --pmc runs; the newest
    profiles/r*_pmc.json recorded for this config).  HBM bytes per launch = fetch_factor * FETCH_SIZE + WRITE_SIZE (KB as counted).
    fetch_factor is the file's `_fetch_factor` (per kernel: calibrated with tools/ubench/stream_read.hip at the kernel's load width,
    profiles/r*_fetch_calibration.json) -- older files without it get the guide's blanket 2.0."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            with open(path) as fh:
                j = json.load(fh)
        except (OSError, ValueError):
            continue
        if int(j.get("_config", 2)) != config or not j.get("kernels"):
            continue
        ff = j.get("_fetch_factor", {})
        out = {}
        for name, e in j["kernels"].items():
            f = float(ff.get(name, ff.get("_default", 2.0)))
            traffic = int((f * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024) if "FETCH_SIZE" in e and "WRITE_SIZE" in e else None
            out[name] = {"traffic": traffic, "valu": e.get("SQ_INSTS_VALU"), "salu": e.get("SQ_INSTS_SALU"), "fetch_factor": f}
        return out, j.get("_frames_per_launch"), "profiles/" + os.path.basename(path)
    return {}, None, None


class Ctx:
    """what every leg of the run needs: the package, torch, the process group and this rank's place in it"""


def make_frames(ctx, cfg, B, first, halo, distinct=None):
    """This rank's B frames of the global synthetic sequence + the halo frame (slot B), in page-locked host memory.  distinct = n: only the
    first n frames of the block are generated (a 1920x1080 frame takes the generator 0.3 s); the block is filled with them and their three
    mirror images in turn (left-right, up-down, both: the same corner statistics, different bytes) -- every slot still holds a different image."""
    uvo, synth = ctx.uvo, ctx.synth
    W, H = cfg["W"], cfg["H"]
    frames = uvo.pinned_empty((B + 1, H, W), np.uint8)
    if distinct is None or distinct >= B:
        frames[:B] = synth.make_sequence(first, B, W, H, n_shapes=cfg["n_shapes"])
        frames[B] = synth.make_sequence(halo, 1, W, H, n_shapes=cfg["n_shapes"])[0]
        return frames
    base = synth.make_sequence(first, distinct, W, H, n_shapes=cfg["n_shapes"])
    for i in range(B + 1):
        a, k = base[i % distinct], (i // distinct) % 4
        if k & 1:
            a = a[:, ::-1]
        if k & 2:
            a = a[::-1, :]
        frames[i] = a
    return frames


def link_probe(ctx, nbytes=256 << 20, reps=4):
    """What this rank's host link can do: page-locked host memory <-> HBM copy rates of one 256 MB buffer (up, down, both directions at
    once), measured on ALL ranks at the same time behind one barrier -- the rates the ranks get while their neighbours pull too."""
    torch, dist = ctx.torch, ctx.dist
    h = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
    d = [torch.empty(nbytes, dtype=torch.uint8, device=ctx.dev) for _ in range(2)]
    st = [torch.cuda.Stream(device=ctx.dev) for _ in range(2)]

    def run(up, down):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            if up:
                with torch.cuda.stream(st[0]):
                    d[0].copy_(h[0], non_blocking=True)
            if down:
                with torch.cuda.stream(st[1]):
                    h[1].copy_(d[1], non_blocking=True)
        torch.cuda.synchronize()
        return reps * nbytes / (time.perf_counter() - t0) / 1e9

    run(True, True)   # first touch of both buffers
    best = lambda up, down: max(run(up, down) for _ in range(3))   # (a single pass of 1 GB is at the mercy of whatever else the host does)
    out = {"h2d_GBps": round(best(True, False), 2), "d2h_GBps": round(best(False, True), 2), "both_directions_each_GBps": round(best(True, True), 2)}
    del h, d
    return out


class HbmWorkload:
    """One rank's share of a configuration, HBM-resident: extractor + matcher handles, the ring of input batches, double-buffered
    outputs, and the step the benchmark times (uvo_extract_batch_device + uvo_hamming_knn2_batch_device)."""
    NRING = 4

    def __init__(self, ctx, cfg, B, frames, depth, fast_mode="adaptive", own_stream=False, env_knobs=False):
        torch, uvo = ctx.torch, ctx.uvo
        self.ctx, self.cfg, self.B, self.frames, self.DEPTH, self.own_stream = ctx, cfg, B, frames, depth, own_stream
        W, H, NFEAT = cfg["W"], cfg["H"], cfg["nfeat"]
        self.W, self.H, self.NFEAT = W, H, NFEAT
        dev = ctx.dev
        # four distinct input batches, read in turn by consecutive steps: the sequence and its mirror images (left-right, up-down, both) --
        # the same corner statistics, different bytes, so that no step finds its input in the Infinity Cache of the step before
        self.d_ring = [torch.from_numpy(frames).to(dev)]
        for k in range(1, self.NRING):
            self.d_ring.append(torch.flip(self.d_ring[0], dims=[d for d, on in ((2, k & 1), (1, k & 2)) if on]).contiguous())
        self.ex = ex = uvo.ORBextractor(NFEAT, SCALE, NLEVELS, 0, FAST_TH, max_width=W, max_height=H, max_batch=B + 1, device=ctx.local_rank)
        self.cap = cap = ex.cap
        self.mt = uvo.ORBmatcher(0.8, max_query=cap, max_train=cap, max_batch=B, device=ctx.local_rank)

        # outputs stay in HBM, double buffered: with pipeline depth 2 the extractor alternates between two scratch sets / streams, so
        # batch i+1's streaming stages overlap batch i's latency-bound stages and its matching.
        class Out:
            def __init__(self):
                self.kp = torch.zeros((B + 1, cap, 7), dtype=torch.float32, device=dev)
                self.desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
                self.n = torch.zeros(B + 1, dtype=torch.int32, device=dev)
                self.idx0 = torch.zeros((B, cap), dtype=torch.int32, device=dev)
                self.idx1 = torch.zeros((B, cap), dtype=torch.int32, device=dev)
                self.d0 = torch.zeros((B, cap), dtype=torch.int16, device=dev)
                self.d1 = torch.zeros((B, cap), dtype=torch.int16, device=dev)

        self.outs = [Out() for _ in range(depth)]
        ex.set_pipeline(depth)
        ex.tune(uvo.UVO_TUNE_FAST_MODE, {"adaptive": uvo.UVO_FAST_MODE_ADAPTIVE, "two_pass": uvo.UVO_FAST_MODE_TWO_PASS,
                                         "single_pass": uvo.UVO_FAST_MODE_SINGLE_PASS}[fast_mode])
        if env_knobs:
            self._env_knobs()
        torch.cuda.synchronize()
        self.counter = [0]
        self.lane_variant = [0] * depth   # which input batch a lane's output buffers hold

    def _env_knobs(self):
        ex, uvo = self.ex, self.ctx.uvo
        if os.environ.get("UVO_BENCH_PYR_FORM"):   # experiment knob: 0 auto / 1 per-level launches / 2 k_pyr_tiles
            ex.tune(uvo.UVO_TUNE_PYR_FORM, int(os.environ["UVO_BENCH_PYR_FORM"]))
        if os.environ.get("UVO_BENCH_PYR_GROUPS"):   # experiment knob: forced level groups of k_pyr_tiles, "first:txXty[w],..."
            for g_ in os.environ["UVO_BENCH_PYR_GROUPS"].split(","):
                first_, grid_ = g_.split(":")
                tx_, ty_ = grid_.rstrip("wr").split("x")
                ex.tune(uvo.UVO_TUNE_PYR_TILE_GROUP, (1 << 24 if grid_.endswith("w") else 0) | (1 << 25 if grid_.endswith("r") else 0) | int(first_) << 16 | int(tx_) << 8 | int(ty_))
        if os.environ.get("UVO_BENCH_BLUR_ROUNDING"):   # experiment knob: 0 = half up on every column, 1 = the x86-64 contract (default)
            ex.tune(uvo.UVO_TUNE_BLUR_ROUNDING, int(os.environ["UVO_BENCH_BLUR_ROUNDING"]))
        if os.environ.get("UVO_BENCH_FUSE"):   # experiment knob: quad-tree + blur as one launch (1, default) or two (0)
            ex.tune(uvo.UVO_TUNE_FUSE_BLUR_TREE, int(os.environ["UVO_BENCH_FUSE"]))
        if os.environ.get("UVO_BENCH_L0"):   # experiment knob: level 0 read in place (1, default) or copied into a padded plane first (0)
            ex.tune(uvo.UVO_TUNE_LEVEL0_INPLACE, int(os.environ["UVO_BENCH_L0"]))
        if os.environ.get("UVO_BENCH_RING"):   # experiment knob: border pixels the resize launches write around a level (4 default, 0 = all 16)
            ex.tune(uvo.UVO_TUNE_PYR_RING, int(os.environ["UVO_BENCH_RING"]))
        if os.environ.get("UVO_BENCH_OCT_WIDE_MAX"):   # experiment knob: quad-tree launch shape (uvo_extractor_tune)
            ex.tune(uvo.UVO_TUNE_OCT_WIDE_MAX, int(os.environ["UVO_BENCH_OCT_WIDE_MAX"]))

    def host_variant(self, k):
        a = self.frames
        if k & 1:
            a = a[:, :, ::-1]
        if k & 2:
            a = a[:, ::-1, :]
        return np.ascontiguousarray(a)

    def extract_only(self):
        li, k = self.counter[0] % self.DEPTH, self.counter[0] % self.NRING
        o = self.outs[li]
        self.lane_variant[li] = k
        self.counter[0] += 1
        self.ex.extract_batch_device(self.d_ring[k].data_ptr(), self.B + 1, self.W, self.H, o.kp.data_ptr(), o.desc.data_ptr(), o.n.data_ptr(), self.cap)
        return o

    # The matching of a batch runs in the stream of the pipeline lane that extracted it (uvo_matcher_attach_extractor); with
    # --matcher-stream own it runs in the matcher's own stream behind an event, and the lane waits for another event before its next batch
    # (two hand-offs between queues per batch: the lane idles 0.3 ms around a 0.13 ms kernel, tools/step_trace_summary.py).
    def step(self):
        o, mt, ex, B, cap = self.extract_only(), self.mt, self.ex, self.B, self.cap
        if self.own_stream:
            mt.wait_extractor(ex)
        else:
            mt.attach(ex)
        # pair p = (frame p, frame p + 1), p < B; pair B-1's partner is the halo frame in slot B
        mt.knn2_batch_device(B, o.desc.data_ptr(), o.n.data_ptr(), cap, o.desc.data_ptr() + cap * 32, o.n.data_ptr() + 4, cap,
                             o.idx0.data_ptr(), o.d0.data_ptr(), o.idx1.data_ptr(), o.d1.data_ptr())
        if self.own_stream:
            mt.release_to_extractor(ex)

    def sync_all(self):
        self.ex.synchronize()
        self.mt.synchronize()
        self.ctx.torch.cuda.synchronize()

    def kernel_times(self):
        """{kernel: (ms, launches)}, {kernel: spread rows} of both handles since profiling was switched on"""
        kt = dict(self.ex.kernel_times())
        sp = dict(self.ex.last_spread)
        kt.update(self.mt.kernel_times())
        sp.update(self.mt.last_spread)
        return kt, sp

    def profile(self, on, only=None):
        if not on:
            self.ex.profile(False)
            self.mt.profile(False)
        elif only is None:
            self.ex.profile(True)
            self.mt.profile(True)
        elif only == "k_knn2":
            self.mt.profile(True)
        else:
            self.ex.profile(True, only=only)

    def measure(self, steps, warmup, n_serial=3):
        """warm-up; the per-kernel durations at pipeline depth 1 (they name the dominant kernel); then the TIMED region: exactly `steps`
        steps between barrier + device sync on both sides, only the dominant kernel's launches carrying events.  -> dict"""
        ctx = self.ctx
        for _ in range(warmup + self.DEPTH):
            self.step()
        self.sync_all()
        serial = {}
        dom = "k_fast_score"
        if os.environ.get("UVO_BENCH_SKIP_SERIAL") != "1":   # (profile collection: a run that holds launches of ONE pipeline depth only)
            # Per-kernel durations first, without cross-batch overlap (pipeline depth 1, every launch bracketed by HIP events on the
            # library's stream; untimed steps): they name the dominant kernel.
            self.ex.set_pipeline(1)
            self.profile(True)
            for _ in range(n_serial):
                self.step()
            self.sync_all()
            serial, _ = self.kernel_times()
            self.profile(False)
            self.ex.set_pipeline(self.DEPTH)
            for _ in range(self.DEPTH):
                self.step()
            self.sync_all()
            dom = max(serial.items(), key=lambda kv: kv[1][0])[0]
        # Timed region: only the dominant kernel's launches carry events (two event records around each of the ~14 launches of a
        # step cost 3 % of the throughput; the roofline needs the live duration of this one kernel only).
        self.profile(True, only=dom)
        dt = timed_steps(self.step, self.sync_all, steps, ctx.dist, ctx.red_dev)
        ktimes, spread = self.kernel_times()
        self.profile(False)
        return {"dt": dt, "dom": dom, "serial": serial, "n_serial": n_serial, "ktimes": ktimes, "spread": spread.get(dom, {}), "timed_variants": list(self.lane_variant)}

    def live_pass(self, n_live):
        """the same pipelined step with every launch bracketed by events (outside the timed region: the event records cost 3 % of the
        throughput) -- what each kernel takes with the other lane's kernels beside it"""
        self.profile(True)
        for _ in range(n_live):
            self.step()
        self.sync_all()
        live, _ = self.kernel_times()
        self.profile(False)
        return live

    def download(self, o):
        B, cap = self.B, self.cap
        return dict(n=o.n.cpu().numpy(), kp=o.kp.cpu().numpy().view(np.uint8).reshape(B + 1, cap, 28), de=o.desc.cpu().numpy(), i0=o.idx0.cpu().numpy(),
                    i1=o.idx1.cpu().numpy(), d0=o.d0.cpu().numpy().astype(np.uint16), d1=o.d1.cpu().numpy().astype(np.uint16))

    def verify(self, host, timed_variants, pairs):
        """outputs of the timed buffers (both lanes' last results, already on the host) against the CPU oracle; a mismatch fails the run"""
        import oracle_lib
        orc = oracle_lib.Oracle()
        oe = orc.extractor(self.NFEAT, SCALE, NLEVELS, FAST_TH)
        need = sorted(set(pairs) | set(p + 1 for p in pairs))
        for li, hb in enumerate(host):
            hv = self.host_variant(timed_variants[li])
            ref = {f: oe(hv[f]) for f in need}
            for f in need:
                kp_o, de_o = ref[f]
                n = int(hb["n"][f])
                if n != len(kp_o) or hb["kp"][f, :n].tobytes() != kp_o.tobytes() or not (hb["de"][f, :n] == de_o).all():
                    raise SystemExit("bench.py: VERIFICATION FAILED -- %s lane %d frame %d of the timed buffers differs from the oracle" % (self.cfg["name"], li, f))
            for p in pairs:
                r = orc.knn2(ref[p][1], ref[p + 1][1])
                nq = len(ref[p][1])
                got = (hb["i0"][p, :nq], hb["d0"][p, :nq].astype(np.int32), hb["i1"][p, :nq], hb["d1"][p, :nq].astype(np.int32))
                if not all((g == e).all() for g, e in zip(got, r)):
                    raise SystemExit("bench.py: VERIFICATION FAILED -- %s lane %d knn-2 rows of pair %d differ from the oracle" % (self.cfg["name"], li, p))
        return {"frames": len(need) * len(host), "distinct_frames": need, "knn2_pairs": list(pairs), "lanes": len(host), "input_batches_of_the_lanes": timed_variants,
                "against": "CPU oracle (this repo's restatement of the reference; parity unpinned), byte for byte"}

    def close(self):
        self.ex.close()
        self.mt.close()
        self.d_ring, self.outs = None, None


def step_spread(m, steps, depth=2):
    """min / median / max of the step's period inside the timed region, from the HIP events around the dominant kernel's launches (one per
    step): with two pipeline lanes taking the steps in turn, half the start-to-start time of launches two apart (one lane's period per
    step); `lane_offset_ms` = start-to-start of consecutive launches (the phase between the lanes); and that kernel's own launch duration"""
    sp = m["spread"]
    key = "period2_" if depth >= 2 else "period_"
    launches = m["ktimes"].get(m["dom"], (0, 0))[1]
    if key + "p50" not in sp or launches != steps:   # (needs three or more steps, and a dominant kernel that is launched once per step: not k_resize_level)
        return None
    out = {"ms_min": round(sp[key + "min"], 4), "ms_median": round(sp[key + "p50"], 4), "ms_max": round(sp[key + "max"], 4), "samples": steps - (2 if depth >= 2 else 1),
           "what": "%s of %s launches inside the timed region (one per step, HIP events on the lanes' streams, enqueue order); `ms_per_step` is the mean "
                   "over the whole region on the host clock" % ("half the start-to-start time of launches two apart (= one pipeline lane's period per step)" if depth >= 2
                                                                 else "start-to-start time of consecutive", m["dom"]),
           "dominant_kernel_launch_ms": {"min": round(sp["min"], 4), "median": round(sp["p50"], 4), "max": round(sp["max"], 4)}}
    if depth >= 2 and "period_p50" in sp:
        out["lane_offset_ms"] = {"min": round(sp["period_min"], 4), "median": round(sp["period_p50"], 4), "max": round(sp["period_max"], 4),
                                 "what": "start-to-start of consecutive launches (alternating lanes): how far apart the two lanes run the same stage"}
    return out


def host_to_host_leg(ctx, wl, first, steps, link):
    """The sharder (uploads from page-locked frames, results gathered into ONE host region shared by all ranks): north_star's "extract +
    match with a final pinned hipMemcpyAsync gather".  The gathered region is compared with the HBM-resident outputs of the same frames."""
    uvo, torch, dist, rank, world = ctx.uvo, ctx.torch, ctx.dist, ctx.rank, ctx.world
    B, W, H, NFEAT, frames, cap = wl.B, wl.W, wl.H, wl.NFEAT, wl.frames, wl.cap
    total = world * B
    chunk = max(B // int(os.environ.get("UVO_BENCH_CHUNKS", "2")), 1)   # chunks per rank and job, two in flight: the upload of one under the kernels of the other
    devices = [uvo.UVO_SHARD_REMOTE] * world
    devices[rank] = ctx.local_rank
    sh = uvo.Sharder(NFEAT, SCALE, NLEVELS, FAST_TH, max_width=W, max_height=H, devices=devices, chunk_frames=chunk, match=True)
    scap = sh.cap
    assert scap == cap
    sizes = [((total, scap), uvo.KEYPOINT_DTYPE), ((total, scap, 32), np.uint8), ((total,), np.int32), ((total, scap), np.int32),
             ((total, scap), np.uint16), ((total, scap), np.int32), ((total, scap), np.uint16)]
    nbytes = sum(int(np.prod(s)) * np.dtype(t).itemsize + 256 for s, t in sizes)
    # two gather regions: a stream of jobs keeps two in flight (job k+1 is submitted before job k is waited for), each gathers into its own
    region = SharedHostRegion(uvo, 2 * nbytes, rank, world, dist)
    sets, off = [], 0
    for _ in range(2):
        arrs = []
        for s, t in sizes:
            a, off = region.carve(off, s, t)
            arrs.append(a)
        sets.append(arrs)
    region.register([a[first:first + B] for arrs in sets for a in arrs], dist)
    g_kp, g_de, g_n, g_i0, g_d0, g_i1, g_d1 = sets[0]
    jobs = [0]
    pending = [None]

    def h2h_run():
        # frames of this rank start at global index `first`; its halo frame sits right behind them in `frames` (for the last
        # rank the job simply ends there: the wrap-around pair exists only in the HBM-resident leg).  One call = one job submitted;
        # the job before it is waited for afterwards, so two are in flight and the lanes never drain.
        t = sh.submit(frames, first, total, *sets[jobs[0] % 2])
        jobs[0] += 1
        if pending[0] is not None:
            sh.wait(pending[0])
        pending[0] = t

    def h2h_drain():
        if pending[0] is not None:
            sh.wait(pending[0])
            pending[0] = None

    for _ in range(3):
        h2h_run()
    h2h_drain()
    reps = max(6, min(30, steps // 4))
    dth = timed_steps(h2h_run, h2h_drain, reps, dist, ctx.red_dev)
    # the gathered region must hold exactly what the HBM-resident leg produces for the same frames (input batch 0 of the ring;
    # this rank's block; pair B-1 only where the halo is the true next frame)
    wl.counter[0] = 0
    wl.step()
    wl.sync_all()
    hb = wl.download(wl.outs[0])
    ok = (g_n[first:first + B] == hb["n"][:B]).all()
    for f in range(B):
        n = int(hb["n"][f])
        ok = ok and g_kp[first + f, :n].tobytes() == hb["kp"][f, :n].tobytes() and (g_de[first + f, :n] == hb["de"][f, :n]).all()
    npairs = B if first + B < total else B - 1
    for p in range(npairs):
        nq = int(hb["n"][p])
        ok = ok and (g_i0[first + p, :nq] == hb["i0"][p, :nq]).all() and (g_d0[first + p, :nq] == hb["d0"][p, :nq]).all() and \
            (g_i1[first + p, :nq] == hb["i1"][p, :nq]).all() and (g_d1[first + p, :nq] == hb["d1"][p, :nq]).all()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctx.red_dev if ctx.red_dev is not None else "cpu")
    if dist is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        raise SystemExit("bench.py: VERIFICATION FAILED -- %s: the gathered host region differs from the HBM-resident outputs" % wl.cfg["name"])
    if rank == 0 and world > 1:   # rank 0 sees every rank's block in the one region
        assert (g_n[:total] > 0).all(), "a rank's block is missing from the gathered region"
    up = B * W * H + (W * H if first + B < total else 0)
    down = B * scap * (28 + 32) + B * 4 + npairs * scap * 12
    job_s = dth / reps
    h2h = {"value": round(total * reps / dth, 1), "unit": "frames/s", "ms_per_job": round(job_s * 1e3, 3), "frames_per_job": total,
           "chunk_frames": chunk, "jobs_in_flight": 2, "pcie_bytes_up_per_rank": up, "pcie_bytes_down_per_rank": down, "gather": "device-to-host copies at precomputed "
           "offsets of one page-locked region shared by all ranks (%s)" % ("uvo_host_alloc" if world == 1 else "/dev/shm mapping + uvo_host_register" +
                                                                              ("" if region.registered else " [registration failed: pageable]")),
           "gathered_equals_hbm_resident": True}
    # what the leg does against what the rank's link can do: the upload is the long direction (PCIe is full duplex: the gather rides the
    # other one), so h2h_frac = upload bytes / (job time x the link's measured host-to-device rate)
    mine = {"rank": rank, "numa_node": ctx.numa_node, "bound_to_local_cpus": bool(ctx.numa_bound), "up_GBps_in_the_leg": round(up / job_s / 1e9, 2),
            "down_GBps_in_the_leg": round(down / job_s / 1e9, 2)}
    if link is not None:
        mine["link_GBps"] = link
        mine["h2h_frac"] = round(up / job_s / 1e9 / link["h2d_GBps"], 4) if link["h2d_GBps"] > 0 else None
    per_rank = [None] * world
    if dist is not None:
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]
    h2h["numa"] = per_rank   # per rank: the NUMA node of its GPU, whether its thread (frames, gather slice, sharder staging) was bound to that node's CPUs,
    #                          the measured link rates (all ranks copying at once) and the share of the link the leg's upload reaches
    if link is not None:
        h2h["link_GBps"] = per_rank[0]["link_GBps"]
        fr = [r["h2h_frac"] for r in per_rank if r.get("h2h_frac") is not None]
        h2h["h2h_frac"] = round(min(fr), 4) if fr else None
        h2h["h2h_frac_note"] = "upload bytes per job / (job time x the rank's measured page-locked host-to-device rate; all ranks probe their links at the same time); min over ranks"
    sh.close()
    region.close(rank, dist)
    return h2h


def config3_record(ctx, steps, link):
    """configs[3] on this run's clock: one GPU's share (batch 128 @ 1920x1080, 2000 feats) HBM-resident, verified against the oracle; with
    several ranks also the real shape -- world x 128 frames through the sharder with the host gather (src/Tracking.cc:946 per frame)."""
    cfg = CONFIGS[3]
    B = int(os.environ.get("UVO_BENCH_C3_BATCH", cfg["batch"]))
    first, halo = shard_frames(ctx.rank, ctx.world, B)
    frames = make_frames(ctx, cfg, B, first, halo, distinct=32)
    wl = HbmWorkload(ctx, cfg, B, frames, 2)
    m = wl.measure(steps, 2, n_serial=2)
    host = [wl.download(o) for o in wl.outs]
    rec = None
    if ctx.rank == 0:
        value = B * steps * ctx.world / m["dt"]
        n_kp = host[0]["n"][:B]
        alg = algorithmic_bytes_per_frame(cfg["W"], cfg["H"], float(n_kp.mean()))
        total_alg = sum(v for k, v in alg.items() if k not in ("k_pad_level0", "k_fast_blur", "k_fast_score_gauss", "k_pyr_tiles", "k_octree_gauss"))
        dom = m["dom"]
        dom_ms, dom_launches = m["ktimes"][dom]
        per_launch = alg.get(dom, 0) * (B + 1) / (dom_launches / steps)
        live_s = dom_ms * 1e-3 / dom_launches
        rec = {"value": round(value, 1), "unit": "frames/s", "n_gpus": ctx.world, "steps": steps, "ms_per_step": round(m["dt"] / steps * 1e3, 4),
               "workload": "%s share of one GPU: batch=%d synthetic %dx%d frames per rank (+1 halo), %d feats, %d levels, fastTh %d, extract + all-pairs knn-2 of "
                           "consecutive frames, HBM-resident I/O; frames = a 32-frame chain of the generator and its three mirror images"
                           % (cfg["name"], B, cfg["W"], cfg["H"], cfg["nfeat"], NLEVELS, FAST_TH),
               "mean_keypoints_per_frame": round(float(n_kp.mean()), 1), "step_spread": step_spread(m, steps),
               "whole_path_frac": round(total_alg * value / ctx.world / 1e9 / HBM_PEAK_GBS, 5),
               "whole_path_algorithmic_bytes_per_frame": int(total_alg),
               "roofline": {"bound": "hbm", "kernel": dom, "avg_launch_ms": round(live_s * 1e3, 5), "algorithmic_bytes_per_launch": int(per_launch),
                            "achieved": round(per_launch / live_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(per_launch / live_s / 1e9 / HBM_PEAK_GBS, 5),
                            "kernel_ms_per_step_unoverlapped": {k: round(v[0] / m["n_serial"], 4) for k, v in sorted(m["serial"].items())}},
               "verification": wl.verify(host, m["timed_variants"], [0, B - 1])}
        rec["verified_frames"] = rec["verification"]["frames"]
    if ctx.world > 1:   # the real shape: world x 128 frames, host in, host gather
        h2h = host_to_host_leg(ctx, wl, first, steps, link)
        if rec is not None:
            rec["host_to_host"] = h2h
    wl.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)   # default: 100 (config 2) / 30 (config 3) -- long enough that filling the two lanes is noise
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subrecords", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--h2h", action="store_true", help="run the host-to-host sharder leg even with --no-subrecords")
    ap.add_argument("--c3", action="store_true", help="run the configs[3] record even with --no-subrecords")
    ap.add_argument("--matcher-stream", choices=["lane", "own"], default="lane",
                    help="where the matching of a batch is enqueued: in the extracting lane's stream (default) or in the matcher's own stream behind events")
    ap.add_argument("--fast-mode", choices=["adaptive", "two_pass", "single_pass"], default="adaptive",
                    help="UVO_TUNE_FAST_MODE of the extractor (speed only; the keypoints are the same in every mode)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    W, H, NFEAT = cfg["W"], cfg["H"], cfg["nfeat"]
    B = args.batch or cfg["batch"]
    steps = args.steps or cfg["steps"]

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # UVO_BENCH_DRYRUN_ONE_GPU=1 (builder's dry run of the N-rank logic on a one-GPU box): every rank uses device 0 and the barrier /
    # max-over-ranks go through gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    dry = os.environ.get("UVO_BENCH_DRYRUN_ONE_GPU") == "1"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    uvo = importlib.import_module("u-vip-slam_amd")
    synth = importlib.import_module("u-vip-slam_amd.synth")
    workloads = importlib.import_module("u-vip-slam_amd.workloads")
    # Host placement before any host buffer exists: this rank's thread goes to the CPUs next to its GPU's PCIe link, so that the frames
    # it generates below and its slice of the gather region are first touched -- and page-locked -- on that NUMA node (N ranks on a
    # two-socket host would otherwise all pull their 84 MB per step from wherever the first rank's pages happened to land)
    all_cpus = os.sched_getaffinity(0)   # (the CPU baseline at the end runs on every core the process was given, not on one socket)
    numa_bound, numa_node = uvo.host_bind_near_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = Ctx()
    ctx.uvo, ctx.synth, ctx.torch, ctx.dist, ctx.rank, ctx.local_rank, ctx.world, ctx.dev, ctx.dry = uvo, synth, torch, dist, rank, local_rank, world, dev, dry
    ctx.red_dev = None if dry else dev   # where the tensors of the barrier-side reductions live (gloo: host)
    ctx.numa_bound, ctx.numa_node = numa_bound, numa_node

    # this rank's shard of the global sequence + the neighbour's first frame, in page-locked host memory (the host-to-host leg uploads
    # from here); slot B of the device copy is the halo
    first, halo = shard_frames(rank, world, B)
    frames = make_frames(ctx, cfg, B, first, halo, distinct=32 if (args.config == 3 and os.environ.get("UVO_BENCH_C3_DISTINCT", "1") == "1") else None)
    DEPTH = int(os.environ.get("UVO_PIPELINE_DEPTH", "2"))
    wl = HbmWorkload(ctx, cfg, B, frames, DEPTH, fast_mode=args.fast_mode, own_stream=args.matcher_stream == "own", env_knobs=True)
    d_imgs, d_ring, NRING = wl.d_ring[0], wl.d_ring, wl.NRING
    step, extract_only, sync_all = wl.step, wl.extract_only, wl.sync_all

    m = wl.measure(steps, args.warmup)
    dt, dom, serial, ktimes = m["dt"], m["dom"], m["serial"], m["ktimes"]
    # ---- the timed buffers, on the host (both lanes' last results) ----
    host = [wl.download(o) for o in wl.outs]
    n_live = max(8, min(20, steps // 4))
    live = wl.live_pass(n_live)
    n_kp = host[0]["n"][:B]

    # ---- verification of the timed outputs against the CPU oracle (rank 0; a mismatch fails the run) ----
    verified = None
    if rank == 0 and not args.no_verify:
        verified = wl.verify(host, m["timed_variants"], sorted(set([0, B // 4, B // 2 - 1, B - 2, B - 1])))   # (p, p + 1); B - 1 pairs with the halo frame

    # ---- what each rank's host link can do (all ranks at once), then the host-to-host leg ----
    sub = {}
    h2h = None
    link = None
    if not args.no_subrecords or args.h2h or args.c3:
        link = link_probe(ctx)
    if not args.no_subrecords or args.h2h:
        h2h = host_to_host_leg(ctx, wl, first, steps, link)

    # ---- configs[3] on this run's clock: one GPU's share as a sub-record (rank 0 alone); with several ranks every rank runs its share and
    # the sharder gathers world x 128 frames ----
    c3 = None
    if args.config == 2 and (not args.no_subrecords or args.c3) and os.environ.get("UVO_BENCH_C3", "1") == "1":
        c3 = config3_record(ctx, max(8, min(12, steps)), link)
        if rank == 0:
            sub["configs[3] per-GPU share" if world == 1 else "configs[3] over %d GPUs" % world] = c3

    # ---- sub-records on rank 0 (the other ranks idle at the barrier below) ----
    c4_cpu = None
    if rank == 0 and not args.no_subrecords:
        # extract only
        for _ in range(DEPTH):
            extract_only()
        sync_all()
        n_eo = max(10, steps // 3)
        t0 = time.perf_counter()
        for _ in range(n_eo):
            extract_only()
        sync_all()
        sub["extract_only"] = {"frames_per_s": round(B * n_eo / (time.perf_counter() - t0), 1), "note": "same batch (+1 halo frame extracted, not counted), no matching"}
        # configs[1]: batch-1 latency
        ex1 = uvo.ORBextractor(NFEAT, SCALE, NLEVELS, 0, FAST_TH, max_width=W, max_height=H, max_batch=1, device=local_rank)
        o1 = (torch.zeros((1, ex1.cap, 7), dtype=torch.float32, device=dev), torch.zeros((1, ex1.cap, 32), dtype=torch.uint8, device=dev),
              torch.zeros(1, dtype=torch.int32, device=dev))
        lat_dev, lat_host = [], []
        for i in range(260):
            t0 = time.perf_counter()
            ex1.extract_batch_device(d_imgs.data_ptr() + (i % B) * W * H, 1, W, H, o1[0].data_ptr(), o1[1].data_ptr(), o1[2].data_ptr(), ex1.cap)
            ex1.synchronize()
            lat_dev.append(time.perf_counter() - t0)
        for i in range(130):
            t0 = time.perf_counter()
            ex1(frames[i % B])
            lat_host.append(time.perf_counter() - t0)
        # the call Tracking makes per frame (src/Tracking.cc:896-946): 400 tracked keypoints fill the occupancy grid, the extractor tops up
        ext = uvo.ORBextractor(NFEAT, SCALE, NLEVELS, 0, FAST_TH, max_width=W, max_height=H, max_batch=1, max_input_keypoints=400, device=local_rank)
        kp1, _ = ex1(frames[0])
        tracked, lat_top = kp1[::max(1, len(kp1) // 400)][:400].copy(), []
        for i in range(130):
            t0 = time.perf_counter()
            ext.extract_tracked(frames[i % B], tracked, 20, NFEAT - len(tracked))
            lat_top.append(time.perf_counter() - t0)
        ext.close()
        sub["configs[1] batch-1 latency"] = {"ms_hbm_resident_median": round(float(np.median(lat_dev[60:])) * 1e3, 4),
                                             "ms_host_in_host_out_median": round(float(np.median(lat_host[30:])) * 1e3, 4),
                                             "ms_host_in_host_out_p95": round(float(np.percentile(lat_host[30:], 95)) * 1e3, 4),
                                             "ms_topup_tracked_host_in_host_out_median": round(float(np.median(lat_top[30:])) * 1e3, 4),
                                             "tracked_keypoints": int(len(tracked))}
        ex1.close()
        # configs[4]: 752x480 extract + isInFrustum + SearchByProjection vs 5000 map points, as one fused call
        W4, H4 = workloads.EUROC_W, workloads.EUROC_H
        img4 = synth.make_frame(31337, W4, H4)
        ex4 = uvo.ORBextractor(1000, SCALE, NLEVELS, 0, 7, max_width=W4, max_height=H4, device=local_rank)
        m4 = uvo.ORBmatcher(0.8, max_query=4096, max_map_points=8192, device=local_rank)
        kp4, de4 = ex4(img4)
        sf4 = ex4.mvScaleFactor.copy()
        mp = workloads.config4_local_map(kp4, de4, sf4)
        cam = uvo.CameraPose.make(mp["R"], mp["t"], mp["Ow"], workloads.EUROC_FX, workloads.EUROC_FY, workloads.EUROC_CX, workloads.EUROC_CY, (0, 0, W4, H4))
        # ... and the host work the reference's tracking thread does between two frames of this configuration: 10 IMU samples (200 Hz
        # against the 20 Hz camera) through IMUPreintegrator::update (src/IMU/IMUPreintegrator.cpp:81-140, restated in
        # tools/hoststress/) -- the "IMU-preintegration stress" of configs[4], timed inside the frame
        imu = workloads.ImuStress()
        stream = workloads.imu_stream(80)
        ts, ti = [], []
        for i in range(70):
            t0 = time.perf_counter()
            pre = imu.preintegrate(stream[i])
            t1 = time.perf_counter()
            k, d_ = ex4(img4)
            a = np.full(len(k), -1, np.int32)
            nm = m4.SearchPointsInFrustum(k, d_, a, cam, mp["xyz"], mp["normal"], mp["min_distance"], mp["max_distance"], None, mp["mp_desc"], sf4, 1.2, 0.5, 1.0)[0]
            ts.append(time.perf_counter() - t0)
            ti.append(t1 - t0)
        sub["configs[4] extract + fused frustum search"] = {"ms_per_frame_median": round(float(np.median(ts[10:])) * 1e3, 4), "matches": int(nm), "keypoints": len(kp4),
                                                            "map_points": 5000, "imu_samples_per_frame": int(stream.shape[1]),
                                                            "imu_preintegration_ms_per_frame": round(float(np.median(ti[10:])) * 1e3, 4),
                                                            "imu_delta_time_s": round(float(pre["delta_time"]), 4),
                                                            "note": "752x480, fastTh 7, host buffers in and out; the frame time includes the 10 host-side IMU updates"}
        cam_o = np.concatenate([mp["R"].reshape(9), mp["t"], mp["Ow"], np.float32([workloads.EUROC_FX, workloads.EUROC_FY, workloads.EUROC_CX, workloads.EUROC_CY]),
                                np.float32([0, W4, 0, H4])]).astype(np.float32)
        c4_cpu = (kp4, de4, sf4, mp, cam_o)
        ex4.close()
        m4.close()
        sub["device_copy_GBps"] = round(device_copy_gbps(torch, dev), 1)
        # the same step on the frames of rounds 1-2's generator, whose noise grows along a 32-frame chain (synth.make_sequence,
        # noise="cumulative": 5 -> 18 % of the pixels pass as FAST corners instead of a steady 4 %): extraction time follows the corner
        # density.  Last, because it overwrites the device frames.
        fr_c = synth.make_sequence(first, B + 1, W, H, n_shapes=cfg["n_shapes"], noise="cumulative")
        d_ring[0].copy_(torch.from_numpy(fr_c).to(dev))
        for k in range(1, NRING):
            d_ring[k].copy_(torch.flip(d_ring[0], dims=[d for d, on in ((2, k & 1), (1, k & 2)) if on]))
        for _ in range(DEPTH + 1):
            step()
        sync_all()
        n_c = max(10, steps // 3)
        t0 = time.perf_counter()
        for _ in range(n_c):
            step()
        sync_all()
        sub["frames with noise accumulating along a chain (rounds 1-2 generator)"] = {
            "frames_per_s": round(B * n_c / (time.perf_counter() - t0), 1),
            "note": "extract + match of the same batch size; every frame = the previous noisy frame resampled + N(0,2)"}

    if rank == 0:
        frames_total = B * steps * world
        value = frames_total / dt
        k_mean = float(n_kp.mean())
        alg = algorithmic_bytes_per_frame(W, H, k_mean)
        # dominant kernel by device time (all launches of a name together), its launches inside the timed region
        dom_ms, dom_launches = ktimes[dom]
        launches_per_step = dom_launches / steps          # k_resize_level is launched once per level, its model is for all 7 together
        avg_launch_s = dom_ms * 1e-3 / dom_launches
        bytes_per_launch = alg.get(dom, 0) * (B + 1) / launches_per_step
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        total_alg = sum(v for kname, v in alg.items() if kname not in ("k_pad_level0", "k_fast_blur", "k_fast_score_gauss", "k_pyr_tiles", "k_octree_gauss"))
        pmc, pmc_frames, pmc_src = load_pmc(args.config)
        scale = (B + 1) / pmc_frames if pmc_frames else 1.0   # the counters were taken at the config's default batch: per-launch figures scale with the frames

        def pmc_of(name, key="valu"):   # profiler name -> summed counters of the device kernels behind it (k_knn2 -> k_knn2_mfma, k_fast_cells -> + _list)
            hit = [v for k, v in pmc.items() if k == name or k.startswith(name + "_")]
            if not hit:
                return None, None
            tr = [h["traffic"] for h in hit if h["traffic"] is not None]
            va = [h[key] for h in hit if h.get(key) is not None]
            return (int(sum(tr) * scale) if tr else None), (sum(va) * scale if va else None)

        # ---- per-kernel table: algorithmic bytes per step, duration with the other lane's kernels beside it (live)
        # and alone on the chip, achieved GB/s and fraction of 8 TB/s on the live duration, HBM bytes from the counters and their ratio
        # to the algorithmic bytes.  Durations are per step (k_resize_level: its 7 launches together). ----
        per_kernel = {}
        for kname in sorted(set(live) | set(serial)):
            ab = int(alg.get(kname, 0) * (B + 1))
            live_ms = live[kname][0] / n_live if kname in live else None
            alone_ms = serial[kname][0] / m["n_serial"] if kname in serial else None
            if kname == dom:   # the dominant kernel's live duration comes from the timed region itself
                live_ms = ktimes[dom][0] / steps
            tr, _ = pmc_of(kname)
            if tr is not None and kname == "k_resize_level":
                tr *= 7    # the counter file holds the mean over its 7 launches
            gbps = ab / (live_ms * 1e-3) / 1e9 if live_ms and ab else None
            gbps_alone = ab / (alone_ms * 1e-3) / 1e9 if alone_ms and ab else None
            per_kernel[kname] = {"alg_bytes": ab, "live_ms": round(live_ms, 5) if live_ms is not None else None,
                                 "alone_ms": round(alone_ms, 5) if alone_ms is not None else None, "GBps": round(gbps, 1) if gbps else None,
                                 "frac": round(gbps / HBM_PEAK_GBS, 5) if gbps else None,
                                 "frac_alone": round(gbps_alone / HBM_PEAK_GBS, 5) if gbps_alone else None, "counter_bytes": tr,
                                 "ratio": round(tr / ab, 3) if tr and ab else None}
        traffic, valu_instr = pmc_of(dom)
        hbm_frac = achieved / HBM_PEAK_GBS
        alone_launch_s = serial[dom][0] / m["n_serial"] * 1e-3 / launches_per_step if dom in serial and serial[dom][0] > 0 else None
        roof_valu = None
        if valu_instr:
            ginstr = valu_instr / avg_launch_s / 1e9
            roof_valu = {"achieved": round(ginstr, 2), "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": round(ginstr / VALU_PEAK_GINSTR, 5),
                         "valu_wave_instructions_per_launch": int(valu_instr),
                         "lane_instructions_per_pixel": round(valu_instr * 64 / (alg["k_fast_score"] * (B + 1)), 2) if dom == "k_fast_score" else None}
            # the same kernel alone on the chip (no second pipeline lane beside it), and against the issue rate of the instruction class most
            # of its instructions belong to: on gfx950 only plain two-operand 32-bit / 16-bit ALU ops and fp32 add / mul / fma issue every
            # 2 cycles per SIMD; packed, three-operand, compare, 32-bit min / max, integer multiply and conversion instructions take 4
            # (measured: tools/ubench/valu_rate3.hip)
            if alone_launch_s:
                g1 = valu_instr / alone_launch_s / 1e9
                roof_valu["alone_on_the_chip"] = {"launch_ms": round(alone_launch_s * 1e3, 5), "achieved": round(g1, 2), "frac": round(g1 / VALU_PEAK_GINSTR, 5),
                                                  "frac_of_4_cycle_class_peak": round(g1 / (VALU_PEAK_GINSTR / 2), 5)}
        # the roof that actually binds the dominant kernel: instruction issue.  VALU + SALU wavefront-instructions per launch (counters)
        # / live launch duration, against the chip's issue peak of the 2-cycle class (and of the 4-cycle class most stencil arithmetic is in)
        issue_roofline = None
        _, salu_instr = pmc_of(dom, "salu")
        if valu_instr and avg_launch_s > 0:
            tot = valu_instr + (salu_instr or 0)
            gi = tot / avg_launch_s / 1e9
            issue_roofline = {"bound": "issue", "kernel": dom, "valu_wave_instructions": int(valu_instr), "salu_wave_instructions": int(salu_instr or 0),
                              "achieved": round(gi, 2), "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": round(gi / VALU_PEAK_GINSTR, 5),
                              "frac_of_4_cycle_class_peak": round(gi / (VALU_PEAK_GINSTR / 2), 5), "source": pmc_src,
                              "note": "peak = 1024 SIMDs x 2.4 GHz / 2 cycles (add / sub / logic / fp32 fma class); packed, three-operand, compare, "
                                      "convert and integer-multiply instructions issue every 4 cycles (tools/ubench/valu_rate3.hip), and the CU's one scalar "
                                      "unit makes a scalar instruction cost a vector slot at this density (tools/ubench/valu_issue.hip)"}
        whole_path_frac = total_alg * value / world / 1e9 / HBM_PEAK_GBS
        prof_tag = PROFILE_TAG if args.config == 2 else PROFILE_TAG + "_hd"
        csv_fracs = {}
        for key_, depth_ in (("alone", 1), ("live", 2)):
            got = rocprof_avg_ms("profiles/%s_kernel_stats_depth%d.csv" % (prof_tag, depth_), dom)
            if got:   # (the files were taken at the configuration's default batch)
                csv_bytes = alg.get(dom, 0) * (cfg["batch"] + 1) / launches_per_step
                csv_fracs[key_] = {"avg_launch_ms": round(got[0], 5), "launches_in_file": got[1], "frac": round(csv_bytes / (got[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 5), "traffic": traffic,
                    "kernel": dom, "avg_launch_ms": round(avg_launch_s * 1e3, 5), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    # the two operating points of the same launch: `frac` / `avg_launch_ms` = frac_live (HIP events inside the timed region, pipeline
                    # depth %d: the other lane's kernels run beside it); frac_alone = the launch alone on the chip (pipeline depth 1, untimed steps)
                    "avg_launch_ms_is": "live (pipeline depth %d, inside the timed region)" % DEPTH,
                    "frac_live": round(hbm_frac, 5),
                    "frac_alone": round(bytes_per_launch / alone_launch_s / 1e9 / HBM_PEAK_GBS, 5) if alone_launch_s else None,
                    "avg_launch_ms_alone": round(alone_launch_s * 1e3, 5) if alone_launch_s else None,
                    "rocprof_csv": {"alone": "profiles/%s_kernel_stats_depth1.csv" % prof_tag, "live": "profiles/%s_kernel_stats_depth2.csv" % prof_tag,
                                    # the same two fractions recomputed from the committed files (taken on the builder's box; pure kernel execution
                                    # time -- the HIP events of this run also see each launch's dispatch)
                                    "from_the_files": csv_fracs,
                                    "note": "rocprofv3 --kernel-trace --stats of this command with UVO_PIPELINE_DEPTH=1 resp. 2 and UVO_BENCH_SKIP_SERIAL=1 (every launch "
                                            "of a file ran at that one depth); AverageNs of the kernel's row x launches per step = avg_launch_ms_alone resp. avg_launch_ms"},
                    "traffic_source": pmc_src, "measured_device_copy_GBps": sub.get("device_copy_GBps"),
                    "per_kernel": per_kernel, "valu": roof_valu, "whole_path_GBps": round(total_alg * value / world / 1e9, 2),
                    "whole_path_algorithmic_bytes_per_frame": int(total_alg),
                    "kernel_ms_per_step_in_timed_region": {k: round(v[0] / steps, 4) for k, v in sorted(ktimes.items())},
                    "kernel_ms_per_step_unoverlapped": {k: round(v[0] / m["n_serial"], 4) for k, v in sorted(serial.items())},
                    "note": "bound says hbm because frac is the HBM fraction this record format asks for (algorithmic bytes per launch / live launch duration "
                            "/ 8 TB/s); the kernel itself is bound by instruction issue: issue_roofline"}
        out = {
            "metric": "frames/sec ORB extract+match, %dx%d @%d kp" % (W, H, NFEAT),
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 4),
            "step_spread": step_spread(m, steps, DEPTH),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: 1xMI355X per rank, batch=%d synthetic %dx%d mono frames (+1 halo frame: the neighbouring rank's first), %d feats, "
                                   "%d levels, fastTh %d, FullDetect extract + all-pairs 256-bit Hamming knn-2 of consecutive frames, HBM-resident I/O"
                                   % (cfg["name"], B, W, H, NFEAT, NLEVELS, FAST_TH),
                       "generator": "SURVEY.md 8(d): value-noise texture + %d shapes + N(0,3) sensor noise per frame; camera motion = a small affine per "
                                    "frame inside 32-frame chains (u-vip-slam_amd/synth.py make_sequence, noise='sensor')" % cfg["n_shapes"],
                       "batch_per_gpu": B, "sharding": "contiguous frame blocks of one global sequence, 1-frame halo from the neighbour, no collective",
                       "pipeline_depth": DEPTH, "mean_keypoints_per_frame": round(k_mean, 1),
                       "input_batches_in_rotation": NRING, "fast_mode": args.fast_mode},
            "verified_frames": verified["frames"] if verified else 0,
            "verification": verified,
            "roofline": roofline,
            "whole_path_frac": round(whole_path_frac, 5),
            "valu_frac": roof_valu["frac"] if roof_valu else None,
            "issue_roofline": issue_roofline,
            "host_to_host": h2h,
            "sub_records": sub,
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            os.sched_setaffinity(0, all_cpus)
            out["cpu_baseline"] = cpu_baseline(frames[:B], cfg, c4_cpu)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
