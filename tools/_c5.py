import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
torch.zeros(1, device="cuda")
uvo = importlib.import_module("u-vip-slam_amd"); synth = importlib.import_module("u-vip-slam_amd.synth")
W, H = 752, 480
rng = np.random.default_rng(7)
img = synth.make_frame(31337, W, H)
ex = uvo.ORBextractor(1000, 1.2, 8, 0, 7, max_width=W, max_height=H)
m = uvo.ORBmatcher(0.8, max_query=4096, max_map_points=8192)
kp, de = ex(img)
sf = ex.mvScaleFactor.copy()
n, M = len(kp), 5000
fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
R, t, Ow = np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
src = rng.integers(0, n, M); z = rng.uniform(2, 12, M)
xyz = np.stack([(kp["x"][src] - cx) / fx * z, (kp["y"][src] - cy) / fy * z, z], 1).astype(np.float32)
nrm = (xyz / np.linalg.norm(xyz, axis=1, keepdims=True)).astype(np.float32)
dist = np.linalg.norm(xyz, axis=1)
mxd = (dist * sf[kp["octave"][src]]).astype(np.float32); mnd = (mxd / sf[7]).astype(np.float32)
mp_desc = np.packbits(np.unpackbits(de[src], axis=1) ^ (rng.random((M, 256)) < 0.06), axis=1)
cam = uvo.CameraPose.make(R, t, Ow, fx, fy, cx, cy, (0, 0, W, H))
def fused():
    a = np.full(n, -1, np.int32)
    return m.SearchPointsInFrustum(kp, de, a, cam, xyz, nrm, mnd, mxd, None, mp_desc, sf, 1.2, 0.5, 1.0)[0]
for _ in range(5): fused()
t0 = time.perf_counter()
for _ in range(50): fused()
print("fused search only ms", (time.perf_counter() - t0) / 50 * 1e3)
m.profile(True)
for _ in range(20): fused()
print({k: round(v[0] / v[1] * 1e3, 1) for k, v in m.kernel_times().items()})
