#!/usr/bin/env python3
"""Build the HIP library (libuvo.so, gfx950) in-tree.  No torch involved: plain hipcc.

  python u-vip-slam_amd/build.py            # build if sources are newer than the .so
  python u-vip-slam_amd/build.py --force

Flags that matter for bit-exactness (DESIGN.md "Numerics"):
  -ffp-contract=off                         x*b + y*a stays two rounded multiplies and one rounded add
  -fhip-fp32-correctly-rounded-divide-sqrt  IEEE fp32 divide in cv::fastAtan2 / the occupancy filter
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libuvo.so")
OBJ = os.path.join(HERE, "build")
SOURCES = ["pyramid.hip", "gauss.hip", "fast.hip", "octree.hip", "describe.hip", "hamming.hip", "search.hip", "match_engine.hip", "grider.hip", "extractor.cpp", "sharder.cpp", "matcher.cpp",
           "matcher_search.cpp", "matcher_batch.cpp", "bow.hip", "bow.cpp", "clahe.hip", "klt.hip", "numa.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fgpu-flush-denormals-to-zero" if False else "-fno-gpu-flush-denormals-to-zero", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-x", "hip"]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=True):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "uvo", "uvo.h"), __file__]
    extra = os.environ.get("UVO_EXTRA_FLAGS", "")
    stamp = os.path.join(OBJ, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == extra
    if not force and same_flags and not _newer(deps, OUT):
        return OUT
    os.makedirs(OBJ, exist_ok=True)

    def cc(src):
        obj = os.path.join(OBJ, src + ".o")
        cmd = [HIPCC] + FLAGS + os.environ.get("UVO_EXTRA_FLAGS", "").split() + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr[-3000:])
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-6000:])
    with open(stamp, "w") as fh:
        fh.write(extra)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
