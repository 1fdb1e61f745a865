"""The scalar numerics the HIP kernels use (csrc/uvo_math.hpp), compiled for the host and compared bit-for-bit with
what the reference calls: libm sinf/cosf (src/ORBextractor.cc:160-161), cv::fastAtan2, cvRound (via the oracle)."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mh():
    src = os.path.join(ROOT, "tests", "emu", "math_host.cpp")
    lib = os.path.join(ROOT, "tests", "emu", "libmath_host.so")
    hdr = os.path.join(ROOT, "u-vip-slam_amd", "csrc", "uvo_math.hpp")
    if not os.path.exists(lib) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(lib):
        # same contract as the device build: no FMA contraction
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", lib, src, "-lpthread"])
    L = ctypes.CDLL(lib)
    L.mh_sincosf.argtypes = [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    L.mh_fast_atan2.restype = ctypes.c_float
    L.mh_fast_atan2.argtypes = [ctypes.c_float, ctypes.c_float]
    L.mh_cv_round.argtypes = [ctypes.c_float]
    L.mh_sincos_mismatches.restype = ctypes.c_long
    L.mh_sincos_mismatches.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    L.mh_logf.restype = ctypes.c_float
    L.mh_logf.argtypes = [ctypes.c_float]
    L.mh_logf_mismatches.restype = ctypes.c_long
    L.mh_logf_mismatches.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    return L


def _bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def test_sincosf_equals_libm_on_every_float_in_0_2pi(mh):
    """Exhaustive: every float in [0, 6.2832] (the whole domain angle*pi/180 can take, src/ORBextractor.cc:159)."""
    first = ctypes.c_uint32()
    bad = mh.mh_sincos_mismatches(0, _bits(6.2832), 4, ctypes.byref(first))
    assert bad == 0, "uvo_sincosf differs from libm on %d inputs, first bit pattern 0x%08x" % (bad, first.value)


def test_logf_equals_libm(mh):
    """`log(ratio)` of MapPoint::PredictScale (src/MapPoint.cc:381): every float in [2^-20, 2^20] -- far beyond any ratio
    of two scene distances -- plus the special cases (zero, subnormals, inf, nan, negatives)."""
    first = ctypes.c_uint32()
    bad = mh.mh_logf_mismatches(_bits(2.0 ** -20), _bits(2.0 ** 20), 4, ctypes.byref(first))
    assert bad == 0, "uvo_logf differs from libm on %d inputs, first bit pattern 0x%08x" % (bad, first.value)
    for lo, hi in ((0x00000000, 0x00000400), (0x007ffc00, 0x00800400), (0x7f7ffc00, 0x7f800400), (0x80000000, 0x80000400), (0xbf800000, 0xbf800400)):
        assert mh.mh_logf_mismatches(lo, hi, 1, ctypes.byref(first)) == 0, hex(first.value)


def test_fast_atan2_equals_oracle(mh, oracle):
    rng = np.random.default_rng(0)
    ys = rng.integers(-3_000_000, 3_000_000, 20000)
    xs = rng.integers(-3_000_000, 3_000_000, 20000)
    for y, x in list(zip(ys, xs)) + [(0, 0), (0, 1), (1, 0), (0, -1), (-1, 0), (5, 5), (-5, 5), (5, -5), (-5, -5)]:
        a = mh.mh_fast_atan2(float(y), float(x))
        b = oracle.fast_atan2(float(y), float(x))
        assert _bits(a) == _bits(b), (y, x, a, b)


def test_cv_round_half_to_even(mh):
    for v, r in ((0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3), (-2.5, -2), (17.0, 17)):
        assert mh.mh_cv_round(v) == r
