#!/usr/bin/env python3
"""Pin kit, step 1: the inputs tools/pin/dump_reference.cpp reads -- the same seeded synthetic frames the tests use, plus the small
inputs of the primitive-level dumps.  Plain binary files + a whitespace-separated manifest (`cases.txt`), numpy only.

  python tools/pin/make_inputs.py --out /tmp/pin_in
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
synth = importlib.import_module("u-vip-slam_amd.synth")

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def frames():
    """name -> (image, nfeatures, fastTh)"""
    seq = synth.make_sequence(0, 3, 640, 512)
    out = {"c2_f0": (seq[0], 1000, 20), "c2_f1": (seq[1], 1000, 20), "c2_f2_th7": (seq[2], 1000, 7),
           "euroc": (synth.make_frame(31337, 752, 480), 1000, 7),
           "hd": (synth.make_frame(7000, 1920, 1080, n_shapes=2500), 2000, 20),
           "harbor400": (synth.make_frame(4711, 640, 512), 400, 20),
           "small": (synth.make_frame(99, 320, 256, n_shapes=120), 300, 20)}
    flat = np.full((256, 320), 128, np.uint8)
    out["flat"] = (flat, 300, 20)
    return out


def topup_inputs(w, h, nfeat, seed=1, n_in=330, d=20):
    """The caller side of src/Tracking.cc:925-946: tracked keypoints + their occupancy grid (Eigen::MatrixXi, column-major)."""
    rng = np.random.default_rng(seed)
    kin = np.zeros(n_in, KP)
    kin["x"], kin["y"] = rng.uniform(20, w - 21, n_in).astype(np.float32), rng.uniform(20, h - 21, n_in).astype(np.float32)
    kin["size"], kin["angle"], kin["octave"], kin["class_id"] = 31, -1, 0, -1
    rows, cols = h // d + 2, w // d + 2
    grid = np.zeros((rows, cols), np.int32, order="F")
    for k in kin:
        grid[int(k["y"] / d), int(k["x"] / d)] += 1
    return kin, grid, d, nfeat - n_in


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    lines = []

    def put(name, arr):
        arr = np.ascontiguousarray(arr)
        arr.tofile(os.path.join(a.out, name + ".bin"))
        return name + ".bin"

    rng = np.random.default_rng(2024)
    for name, (img, nfeat, th) in frames().items():
        h, w = img.shape
        lines.append("frame %s %d %d %d %d %s" % (name, w, h, nfeat, th, put("frame_" + name, img)))
    # top-up mode (FullDetect = false) on the harbor frame, parameters of Data/Settings_VI_Aqualoc_harbor.yaml:67-79
    kin, grid, d, need = topup_inputs(640, 512, 400)
    lines.append("topup harbor400 %d %d %d %d %d %s %s" % (len(kin), grid.shape[0], grid.shape[1], d, need, put("topup_kp", kin),
                                                          put("topup_grid", np.asfortranarray(grid).ravel(order="F"))))
    # cv::FAST on ROIs of c2_f0: x y w h threshold
    for k in range(12):
        rw, rh = int(rng.integers(20, 70)), int(rng.integers(20, 70))
        x, y = int(rng.integers(0, 640 - rw)), int(rng.integers(0, 512 - rh))
        lines.append("fast c2_f0 %d %d %d %d %d" % (x, y, rw, rh, 20 if k % 2 == 0 else 7))
    # ... and the border cases of an ROI: the smallest one with an interior pixel (7 x 7), interiors one pixel wide / high, and ROIs that
    # touch each corner and edge of the parent image (cv::FAST must not look outside the ROI; NMS rows at the ROI's first / last row)
    for x, y, rw, rh in ((0, 0, 7, 7), (0, 0, 7, 40), (0, 0, 40, 7), (0, 0, 36, 36), (640 - 36, 0, 36, 36), (0, 512 - 36, 36, 36), (640 - 36, 512 - 36, 36, 36),
                         (640 - 7, 512 - 7, 7, 7), (300, 0, 45, 9), (0, 200, 9, 45), (640 - 8, 100, 8, 60), (100, 512 - 8, 60, 8)):
        for th in (20, 7):
            lines.append("fast c2_f0 %d %d %d %d %d" % (x, y, rw, rh, th))
    # cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, 2) as Utils::ratioMatching calls it (include/utils.h:92-101): which train index
    # wins among EQUAL distances, first and second neighbour -- low-entropy descriptors (many exact ties), duplicated train rows, a
    # train set of one and of two rows
    def lowent(n, bits):
        d = np.zeros((n, 32), np.uint8)
        d[:, :bits // 8 + 1] = rng.integers(0, 256, (n, bits // 8 + 1), dtype=np.uint8) & np.uint8(0x0f)
        return d
    sets = {"ties": (lowent(300, 20), lowent(400, 20)), "dups": (rng.integers(0, 256, (100, 32), dtype=np.uint8),
                                                                 np.repeat(rng.integers(0, 256, (50, 32), dtype=np.uint8), 3, axis=0)),
            "one": (lowent(40, 12), lowent(1, 12)), "two": (lowent(40, 12), lowent(2, 12))}
    for name, (q, t) in sets.items():
        lines.append("knn %s %d %d %s %s" % (name, len(q), len(t), put("knn_%s_q" % name, q), put("knn_%s_t" % name, t)))
    # cv::GaussianBlur(roi, roi, Size(7, 7), 2, 2, BORDER_REFLECT_101) on the ROI of a padded parent (src/ORBextractor.cc:942), on noise tall
    # enough to hold exact .5 column sums (one pixel in 65 536): THE case that tells the two rounding contracts apart -- the generic column
    # filter rounds them up everywhere, an SSE2 build's SymmColumnVec_32s8u to even on the columns 0 .. (w & ~3) - 1 (UVO_TUNE_BLUR_ROUNDING).
    # Widths with a scalar tail of 0, 2 and 1 columns.  File = the padded parent, (h + 32) x (w + 32), REFLECT_101 of the noise.
    for k, (gw, gh) in enumerate(((320, 1200), (318, 1200), (157, 2400))):
        noise = np.random.default_rng(700 + k).integers(0, 256, (gh, gw)).astype(np.uint8)
        lines.append("gauss %d %d %d %s" % (k, gw, gh, put("gauss_%d_in" % k, np.pad(noise, 16, mode="reflect"))))
    # cv::fastAtan2 on a grid of (y, x) incl. zeros, equal magnitudes, negative and tiny values
    v = np.concatenate([np.float32([0, 1, -1, 1e-12, -1e-12, 3e7]), rng.normal(0, 50000, 2000).astype(np.float32), rng.integers(-200000, 200000, 2000).astype(np.float32)])
    yy, xx = rng.permutation(v)[:4000], rng.permutation(v)[:4000]
    yy[:6], xx[:6] = [0, 0, 1, -1, 5, -5], [0, 1, 0, 0, 5, 5]
    lines.append("atan2 %d %s %s" % (len(yy), put("atan2_y", yy.astype(np.float32)), put("atan2_x", xx.astype(np.float32))))
    # DistributeOctTree on candidate lists: n minX maxX minY maxY N level file   (keypoints: pt relative to (minX, minY), response)
    for k, (W, H, N, P) in enumerate([(614, 486, 217, 2600), (1894, 1054, 434, 9000), (153, 117, 60, 400), (614, 486, 217, 120), (300, 200, 50, 700)]):
        pts = np.unique(np.stack([rng.integers(0, W, P), rng.integers(0, H, P)], 1), axis=0)
        nC, nR = W // 30, H // 30
        wC, hC = -(-W // nC), -(-H // nR)
        j, i = np.minimum(np.maximum(pts[:, 0] - 3, 0) // wC, nC - 1), np.minimum(np.maximum(pts[:, 1] - 3, 0) // hC, nR - 1)
        pts = pts[np.lexsort((pts[:, 0], pts[:, 1], j, i))]            # the reference's candidate order: cell-major, raster inside a cell
        kp = np.zeros(len(pts), KP)
        kp["x"], kp["y"], kp["response"], kp["size"], kp["angle"], kp["class_id"] = pts[:, 0], pts[:, 1], rng.integers(7, 200, len(pts)), 7, -1, -1
        lines.append("octree %d %d %d %d %d %d %d %d %s" % (k, len(kp), 13, 13 + W, 13, 13 + H, N, 0, put("oct_%d_in" % k, kp)))
    # cv::Mat arithmetic of the projection prologues: R (3x3), P (3x1), t (3x1), CV_32F
    n = 64
    R = rng.normal(0, 1, (n, 3, 3)).astype(np.float32)
    P = (rng.normal(0, 5, (n, 3)) * rng.choice([1, 1e-3, 1e3], (n, 1))).astype(np.float32)
    t = rng.normal(0, 2, (n, 3)).astype(np.float32)
    lines.append("gemm %d %s %s %s" % (n, put("gemm_R", R), put("gemm_P", P), put("gemm_t", t)))
    # CLAHE (src/Tracking.cc:425-431) and the KLT step (src/FrameKTL.cc:76, src/Tracking.cc:1046-1047) on c2_f0 -> c2_f1
    lines.append("clahe c2_f0 4.0 12 12")
    lines.append("clahe small 4.0 12 12")
    seq = synth.make_sequence(0, 2, 640, 512)
    pts = np.stack([rng.uniform(30, 610, 800), rng.uniform(30, 480, 800)], 1).astype(np.float32)
    lines.append("klt c2_f0 c2_f1 21 21 5 30 0.01 0.0001 %d %s" % (len(pts), put("klt_pts", pts)))
    # Tracking::undistort_point (src/Tracking.cc:1265-1283): pin-hole (Data/Settings_VIORB.yaml:14-23) and fisheye (Settings_VI_Aqualoc_harbor.yaml:24-33,102)
    upts = np.stack([rng.uniform(-20, 772, 3000), rng.uniform(-20, 500, 3000)], 1).astype(np.float32)
    f = put("undistort_pts", upts)
    lines.append("undistort pinhole 0 %d 458.654 457.296 367.215 248.375 4 -0.28340811 0.07395907 0.00019359 1.76187114e-05 %s" % (len(upts), f))
    lines.append("undistort fisheye 1 %d 413.32595366596017 413.70198739483686 305.9507483284928 259.4439948946375 4 -0.06125568297136998 "
                 "-0.003796743395135256 0.027326634771204592 -0.030296403142887066 %s" % (len(upts), f))
    with open(os.path.join(a.out, "cases.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("wrote %d cases to %s" % (len(lines), a.out))


if __name__ == "__main__":
    main()
