"""Host-to-device / device-to-host rate of page-locked memory on this box: one copy at a time vs several streams at once."""
import time, torch
n = 128 << 20
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
s = [torch.cuda.Stream() for _ in range(4)]
def run(k, up=True, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(k):
            with torch.cuda.stream(s[i]):
                (d[i].copy_(h[i], non_blocking=True) if up else h[i].copy_(d[i], non_blocking=True))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return k * reps * n / dt / 1e9
for k in (1, 2, 4):
    print("streams %d: H2D %.1f GB/s  D2H %.1f GB/s" % (k, run(k, True), run(k, False)))
# both directions at once
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s[0]): d[0].copy_(h[0], non_blocking=True)
    with torch.cuda.stream(s[1]): h[1].copy_(d[1], non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("H2D + D2H together: %.1f GB/s each" % (5 * n / dt / 1e9))
# the library's own page-locked allocations (uvo_host_alloc) and registered shared mappings (uvo_host_register), same copies
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
uvo = importlib.import_module("u-vip-slam_amd")
a = uvo.pinned_empty((n,), np.uint8); a[:] = 1
ta = torch.from_numpy(a)
def rate(src, dst, reps=8):
    dst.copy_(src, non_blocking=True); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); return reps * n / (time.perf_counter() - t0) / 1e9
print("torch pinned  H2D %.1f  D2H %.1f GB/s" % (rate(h[0], d[0]), rate(d[0], h[0])))
print("uvo_host_alloc H2D %.1f  D2H %.1f GB/s" % (rate(ta, d[0]), rate(d[0], ta)))
