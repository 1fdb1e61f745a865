import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
torch.zeros(1, device="cuda")
uvo = importlib.import_module("u-vip-slam_amd"); synth = importlib.import_module("u-vip-slam_amd.synth")
import oracle_lib
o = oracle_lib.Oracle()
rng = np.random.default_rng(1)
for t in range(22):
    w, h = int(rng.integers(120, 900)), int(rng.integers(120, 700)); nlev = int(rng.integers(1, 9))
    scale = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.25, 1.4]))
    while nlev > 1 and min(w, h) / scale ** (nlev - 1) < 60: nlev -= 1
    nfeat = int(rng.integers(50, 2500)); th = int(rng.choice([3, 7, 12, 20, 20, 35])); kind = t % 5
    if kind == 3: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 4: img = (rng.integers(0, 30, (h, w)) + 100).astype(np.uint8)
    else: img = synth.make_frame(int(rng.integers(1 << 30)), w, h, n_shapes=max(20, w * h // 900))
print(w,h,nlev,scale,nfeat,th,kind)
ex = uvo.ORBextractor(nfeat, scale, nlev, 0, th, max_width=w, max_height=h); oe = o.extractor(nfeat, scale, nlev, th)
kg, dg = ex(img); ko, do = oe(img)
print("quota", oe.quota)
for l in range(nlev):
    cg = ex.read_candidates(l); co = oe.level_candidates(l)
    sg = set(map(tuple, cg.tolist())); so = set(map(tuple, np.asarray(co).tolist()))
    print("level", l, "cand gpu", len(cg), "oracle", len(co), "same set", sg == so, "kp gpu", int((kg["octave"]==l).sum()), "oracle", int((ko["octave"]==l).sum()), ex.level_dims(l))
