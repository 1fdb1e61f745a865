"""Runs under the image's OTHER interpreter (/opt/conda/bin/python3.9: scikit-image 0.18.3, numpy 1.26) -- the test suite's own python
has no scikit-image.  Independent third-party implementations of three things the oracle restates from recall of OpenCV:
  corner_fast(n = 9)      the FAST-9/16 segment test (corner predicate only: scikit-image's response is not OpenCV's score);
  corner_orientations     the intensity-centroid angle over ORB's circular 31 x 31 patch (scikit-image's own OFAST mask);
  ORB sampling pattern    scikit-image ships the 256 published rBRIEF point pairs as a text file;
  steered rBRIEF          its descriptor loop (orb_cy._orb_loop: every pair rotated by the keypoint's angle, rounded, compared) on a given
                          image, keypoints and angles.
usage: skimage_probe.py in.npz out.npz      in: img (uint8 H x W), thresholds (ints), corners (N x 2 ints, row / col)
                                            optional: desc_img (uint8 H x W), desc_kp (M x 2 ints, row / col), desc_angle (M radians)"""
import sys

import numpy as np
import skimage
from skimage.feature import corner_fast, corner_orientations
from skimage.feature import orb as _orb
from skimage.feature import _orb_descriptor_positions as _pos

d = np.load(sys.argv[1])
img = d["img"].astype(np.float64)   # whole numbers in float64: corner_fast's `>` / `<` tests are exact
out = {"version": np.array(skimage.__version__)}
for t in d["thresholds"]:
    out["fast_%d" % int(t)] = (corner_fast(img, n=9, threshold=float(t)) > 0)
out["orientations"] = corner_orientations(img, d["corners"].astype(np.intp), _orb.OFAST_MASK)
out["ofast_mask"] = _orb.OFAST_MASK.astype(np.uint8)
out["pos0"] = _pos.POS0.astype(np.int32)
out["pos1"] = _pos.POS1.astype(np.int32)
if "score_upto" in d.files:
    # cornerScore = the largest threshold at which the pixel still is a corner: count the thresholds 0 .. score_upto it survives
    cnt = np.zeros(img.shape, np.int32)
    for t in range(int(d["score_upto"]) + 1):
        cnt += corner_fast(img, n=9, threshold=float(t)) > 0
    out["score"] = cnt - 1          # -1: not a corner even at threshold 0
if "desc_img" in d.files:
    from skimage.feature.orb_cy import _orb_loop
    out["descriptors"] = np.asarray(_orb_loop(np.ascontiguousarray(d["desc_img"], dtype=np.float64), np.ascontiguousarray(d["desc_kp"], dtype=np.intp),
                                              np.ascontiguousarray(d["desc_angle"], dtype=np.float64))).astype(np.uint8)
np.savez(sys.argv[2], **out)
