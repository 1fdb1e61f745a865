"""Synthetic workloads of the BASELINE.json configurations, shared by bench.py, tools/ and the tests.

configs[4]: one 752x480 EuRoC-sized frame (fastTh 7) against a local map of 10 key frames x 500 = 5000 map points: every map point is
the back-projection of one of the frame's own keypoints at a random depth (so that isInFrustum + SearchByProjection find true
correspondences), its descriptor that keypoint's descriptor with Binomial(256, 0.06) bit flips, its distance-invariance interval
the one the keypoint's octave implies.  Pure numpy, deterministic per seed.
"""
import numpy as np

EUROC_W, EUROC_H = 752, 480
EUROC_FX, EUROC_FY, EUROC_CX, EUROC_CY = 458.654, 457.296, 367.215, 248.375   # Data/Settings_VIORB.yaml:9-12


def config4_local_map(kp, desc, scale_factors, n_points=5000, seed=7):
    """-> dict(xyz, normal, min_distance, max_distance, mp_desc, R, t, Ow) for the keypoints / descriptors of the frame."""
    rng = np.random.default_rng(seed)
    n, M = len(kp), n_points
    sf = np.asarray(scale_factors, np.float32)
    src = rng.integers(0, n, M)
    z = rng.uniform(2, 12, M)
    xyz = np.stack([(kp["x"][src] - EUROC_CX) / EUROC_FX * z, (kp["y"][src] - EUROC_CY) / EUROC_FY * z, z], 1).astype(np.float32)
    nrm = (xyz / np.linalg.norm(xyz, axis=1, keepdims=True)).astype(np.float32)
    dist = np.linalg.norm(xyz, axis=1)
    mxd = (dist * sf[kp["octave"][src]]).astype(np.float32)
    mnd = (mxd / sf[len(sf) - 1]).astype(np.float32)
    flip = rng.random((M, 256)) < 0.06
    mp_desc = np.packbits(np.unpackbits(desc[src], axis=1) ^ flip, axis=1)
    return dict(xyz=xyz, normal=nrm, min_distance=mnd, max_distance=mxd, mp_desc=mp_desc, src=src,
                R=np.eye(3, dtype=np.float32), t=np.zeros(3, np.float32), Ow=np.zeros(3, np.float32))


def imu_stream(n_frames, rate_hz=200, frame_hz=20, seed=11):
    """configs[4] "IMU-preintegration stress": rate_hz / frame_hz samples (gyro rad/s, accel m/s^2, dt) per frame, a slowly turning
    and accelerating body with sensor noise.  -> (n_frames, samples_per_frame, 7) float64 rows [wx wy wz ax ay az dt]."""
    rng = np.random.default_rng(seed)
    per = rate_hz // frame_hz
    tt = np.arange(n_frames * per) / rate_hz
    w = np.stack([0.3 * np.sin(0.7 * tt), 0.2 * np.cos(0.5 * tt), 0.1 * np.sin(0.3 * tt + 1.0)], 1) + rng.normal(0, 0.002, (len(tt), 3))
    a = np.stack([0.5 * np.cos(0.9 * tt), 0.4 * np.sin(0.6 * tt), 9.81 + 0.2 * np.sin(1.1 * tt)], 1) + rng.normal(0, 0.02, (len(tt), 3))
    dt = np.full((len(tt), 1), 1.0 / rate_hz)
    return np.concatenate([w, a, dt], 1).reshape(n_frames, per, 7)


class ImuStress:
    """The restated IMUPreintegrator::update (tools/hoststress/imu_preintegrator.cpp; src/IMU/IMUPreintegrator.cpp:81-140) behind ctypes:
    the host work the reference's tracking thread does between frames in configs[4]."""

    def __init__(self):
        import ctypes
        import os
        import subprocess
        here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "hoststress")
        lib = os.path.join(here, "libimu_stress.so")
        src = os.path.join(here, "imu_preintegrator.cpp")
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", lib, src])
        self._L = ctypes.CDLL(lib)
        self._L.imu_preintegrate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        self._L.imu_preintegrate.restype = None

    def preintegrate(self, samples, reset_every=0, gyr_cov=(1.6968e-4) ** 2 * 200, acc_cov=(2.0e-3) ** 2 * 200):
        """samples (n, 7) float64 [wx wy wz ax ay az dt] -> dict(delta_P, delta_V, delta_R, delta_time, cov_trace)"""
        s = np.ascontiguousarray(samples, np.float64).reshape(-1, 7)
        out = np.zeros(17)
        self._L.imu_preintegrate(s.ctypes.data, len(s), int(reset_every), float(gyr_cov), float(acc_cov), out.ctypes.data)
        return dict(delta_P=out[:3].copy(), delta_V=out[3:6].copy(), delta_R=out[6:15].reshape(3, 3).copy(), delta_time=out[15], cov_trace=out[16])
