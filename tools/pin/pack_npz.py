#!/usr/bin/env python3
"""Pin kit, step 3: pack the dumper's output directory into one .npz the tests read (tests/golden/reference_pins.npz).

  python tools/pin/pack_npz.py /tmp/pin_out tests/golden/reference_pins.npz
"""
import os
import sys

import numpy as np

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
DT = {"u1": np.uint8, "i2": np.int16, "i4": np.int32, "f4": np.float32, "f8": np.float64, "kp": KP}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {}
    for line in open(os.path.join(src, "manifest.txt")):
        tok = line.split()
        if not tok:
            continue
        name, dt, nd = tok[0], DT[tok[1]], int(tok[2])
        dims = [int(v) for v in tok[3:3 + nd]]
        a = np.fromfile(os.path.join(src, name + ".bin"), dtype=dt)
        out[name.replace("/", "__")] = a.reshape(dims)
    out["build_info"] = np.frombuffer(open(os.path.join(src, "build_info.txt"), "rb").read(), dtype=np.uint8)
    np.savez_compressed(dst, **out)
    print("packed %d arrays into %s" % (len(out), dst))


if __name__ == "__main__":
    main()
