"""The C-ABI library loads without a GPU and exports every symbol include/uvo/uvo.h declares; without a device the
product fails loudly instead of falling back to the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "uvo", "uvo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(uvo_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(uvo):
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(uvo.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libuvo.so does not export %s" % n
    assert sorted(uvo.ABI_SYMBOLS) == names, "python binding list out of sync with uvo.h"


def test_keypoint_layout_is_cv_keypoint(uvo):
    d = uvo.KEYPOINT_DTYPE
    assert d.itemsize == 28
    assert [d.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == [0, 4, 8, 12, 16, 20, 24]


def test_no_cpu_fallback_without_device(uvo):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(uvo.UvoError) as ei:
        uvo.ORBextractor(1000, 1.2, 8, 0, 20)
    assert ei.value.code == uvo.UVO_E_NODEVICE
    with pytest.raises(uvo.UvoError) as ei:
        uvo.ORBmatcher(0.8)
    assert ei.value.code == uvo.UVO_E_NODEVICE


def test_product_sources_never_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "u-vip-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("oracle/", "orb_oracle", "liborb_oracle", "oracle_lib", "import oracle", "oracle."):
                    assert needle not in txt, "%s mentions %s" % (os.path.join(dirpath, f), needle)
    for f in os.listdir(os.path.join(ROOT, "include", "uvo")):
        p = os.path.join(ROOT, "include", "uvo", f)
        if os.path.isfile(p):
            assert "oracle" not in open(p).read()
