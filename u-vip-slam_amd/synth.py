"""Synthetic mono frames for the parity tests and bench.py.

A frame is the sum of (a) 3-octave value noise (smooth texture, amplitude 60), (b) random bright/dark
rectangles and discs (4-40 px, contrast +-30..+-90) that give FAST corners at several scales and
(c) N(0,3) pixel noise, clipped to [0,255].  `warp_frame` makes the "next" frame of a sequence by a small
random affine (<= 3 px shift, <= 2 deg rotation) so consecutive-frame matching has true correspondences.
Pure numpy; deterministic per seed.
"""
import numpy as np


def _value_noise(rng, h, w, cell):
    gh, gw = h // cell + 3, w // cell + 3
    g = rng.random((gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_frame(seed, w=640, h=512, n_shapes=400, noise=True):
    """One frame.  noise=False returns the float scene (texture + shapes) before the N(0,3) pixel noise, rounding and clipping; the
    random stream is the same either way."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 110.0, dtype=np.float32)
    for cell, amp in ((64, 30.0), (32, 20.0), (16, 10.0)):
        img += (_value_noise(rng, h, w, cell) - 0.5) * 2 * amp
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_shapes):
        cx, cy = rng.integers(0, w), rng.integers(0, h)
        sz = int(rng.integers(4, 41))
        contrast = float(rng.integers(30, 91)) * (1 if rng.random() < 0.5 else -1)
        kind = rng.integers(0, 3)
        x0, x1 = max(cx - sz, 0), min(cx + sz + 1, w)
        y0, y1 = max(cy - sz, 0), min(cy + sz + 1, h)
        if x1 <= x0 or y1 <= y0:
            continue
        sub = img[y0:y1, x0:x1]
        lx = xx[y0:y1, x0:x1] - cx
        ly = yy[y0:y1, x0:x1] - cy
        if kind == 0:  # axis-aligned rectangle
            hw, hh = sz / 2, max(2, sz / 2 * rng.uniform(0.4, 1.0))
            m = (np.abs(lx) <= hw) & (np.abs(ly) <= hh)
        elif kind == 1:  # rotated rectangle
            th = rng.uniform(0, np.pi)
            c, s = np.cos(th), np.sin(th)
            u = lx * c + ly * s
            v = -lx * s + ly * c
            m = (np.abs(u) <= sz / 2) & (np.abs(v) <= max(2, sz / 3))
        else:  # disc
            m = lx * lx + ly * ly <= (sz / 2) ** 2
        sub[m] += contrast
    if not noise:
        return img
    img += rng.normal(0.0, 3.0, size=(h, w)).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _step_affine(seed, w, h):
    """The small random affine of one sequence step (<= 3 px shift, <= 2 deg rotation about the image centre) as a 3x3 matrix that maps
    destination pixel coordinates to source coordinates -- the same draw as warp_frame(seed)."""
    rng = np.random.default_rng(seed)
    ang = np.deg2rad(rng.uniform(-2, 2))
    tx, ty = rng.uniform(-3, 3, size=2)
    c, s = np.cos(ang), np.sin(ang)
    cx, cy = w / 2, h / 2
    return np.array([[c, s, cx - tx - c * cx - s * cy], [-s, c, cy - ty + s * cx - c * cy], [0.0, 0.0, 1.0]])


def warp_frame(img, seed):
    """Previous frame warped by a small random affine (nearest-neighbour resample + fresh N(0,2) noise)."""
    rng = np.random.default_rng(seed)
    h, w = img.shape
    ang = np.deg2rad(rng.uniform(-2, 2))
    tx, ty = rng.uniform(-3, 3, size=2)
    c, s = np.cos(ang), np.sin(ang)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cx, cy = w / 2, h / 2
    sx = c * (xx - cx) + s * (yy - cy) + cx - tx
    sy = -s * (xx - cx) + c * (yy - cy) + cy - ty
    sx = np.clip(np.rint(sx), 0, w - 1).astype(np.int32)
    sy = np.clip(np.rint(sy), 0, h - 1).astype(np.int32)
    out = img[sy, sx].astype(np.float32) + rng.normal(0, 2.0, size=(h, w)).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def make_batch(n, w=640, h=512, seed0=1000):
    return np.stack([make_frame(seed0 + i, w, h) for i in range(n)])


def make_sequence(first, count, w=640, h=512, chain=32, n_shapes=400, seed_base=1000, noise="sensor"):
    """Frames [first, first + count) of the global synthetic sequence bench.py and the sharding tests use.  Frame g starts a new chain
    when g % chain == 0; inside a chain the camera moves by a small random affine per frame (seed seed_base + g), so consecutive frames
    truly correspond.  Any shard of the sequence can be generated on its own (it replays its chain from the chain's start), which is
    what lets every rank / device produce its frames and its neighbour's halo frame independently.

    noise="sensor" (default, SURVEY.md 8(d): N(0,3) pixel noise): a frame is the chain's noise-free scene sampled through the composed
    motion of the chain so far, plus fresh N(0,3) noise -- every frame has the same noise level, as frames of one camera do.
    noise="cumulative": the round-2 generator kept for comparison -- every frame is the previous *noisy* frame resampled plus N(0,2),
    so the noise grows along a chain (sigma 3 -> 11.5 over 32 frames) and with it the share of pixels that pass as FAST corners
    (5 % -> 18 % at threshold 7)."""
    out = []
    g = first - first % chain
    prev = None
    scene, T = None, None
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    while g < first + count:
        if noise == "cumulative":
            prev = make_frame(seed_base + g, w, h, n_shapes) if g % chain == 0 else warp_frame(prev, seed_base + g)
        else:
            if g % chain == 0:
                scene, T = make_frame(seed_base + g, w, h, n_shapes, noise=False), np.eye(3)
                if g >= first:
                    prev = make_frame(seed_base + g, w, h, n_shapes)  # the chain's first frame is make_frame itself
            else:
                T = T @ _step_affine(seed_base + g, w, h)           # destination pixel -> the scene's coordinates
                if g >= first:
                    sx = np.clip(np.rint(T[0, 0] * xx + T[0, 1] * yy + T[0, 2]), 0, w - 1).astype(np.int32)
                    sy = np.clip(np.rint(T[1, 0] * xx + T[1, 1] * yy + T[1, 2]), 0, h - 1).astype(np.int32)
                    rng = np.random.default_rng((seed_base + g) * 7919 + 17)
                    img = scene[sy, sx] + rng.normal(0.0, 3.0, size=(h, w)).astype(np.float32)
                    prev = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if g >= first:
            out.append(prev)
        g += 1
    return np.stack(out) if out else np.zeros((0, h, w), np.uint8)
