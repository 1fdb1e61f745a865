#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel.

usage: pmc_summary.py [--config N] [--calibration FILE] DIR [DIR ...]   (each DIR is searched recursively for *_counter_collection.csv;
FILE = the output of tools/fetch_calibration.py: FETCH_SIZE factors per load width, written into "_fetch_factor" per kernel)
Prints {"_note", "_config", "kernels": {name: mean per-launch value of every counter found + VGPR / LDS use + launch count}} -- the
format bench.py reads its `roofline.traffic` and VALU roof from (profiles/r*_pmc.json).  Template arguments are dropped from the
kernel names (k_octree<256> -> k_octree), so the names match the library's own per-kernel timers.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    args = sys.argv[1:]
    config = 2
    calib = None
    while args and args[0] in ("--config", "--calibration"):
        if args[0] == "--config":
            config = int(args[1])
        else:
            with open(args[1]) as fh:
                calib = json.load(fh)
        args = args[2:]
    for d in args:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].split("(")[0]
                    if "uvo::" not in name:
                        continue
                    name = name.split("uvo::")[-1].split("<")[0]
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    meta[name] = {"vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]), "lds": int(row["LDS_Block_Size"])}
    out = {}
    for k, ctrs in sorted(acc.items()):
        out[k] = dict(meta[k])
        for c, vals in sorted(ctrs.items()):
            out[k][c] = sum(vals) / len(vals)
            out[k]["launches"] = len(vals)
    note = ("rocprofv3 --pmc, separate passes (FETCH_SIZE / WRITE_SIZE / SQ_*), mean per launch over bench.py --config %d --steps 2 --warmup 1; "
            "FETCH_SIZE and WRITE_SIZE in KB as reported; HBM read bytes = _fetch_factor[kernel] * FETCH_SIZE_KB * 1024, the factor calibrated "
            "at the kernel's load width with tools/ubench/stream_read.hip (MI355X_MICROARCH.md states 2.0 for 16-B-per-lane streaming reads "
            "and leaves other widths to calibration); k_resize_level is the mean over its 7 launches per step; collected with "
            "tools/profile_round.sh" % config)
    frames = {2: 257, 3: 129}.get(config)   # bench.py's default batch of the config + its halo frame
    doc = {"_note": note, "_config": config, "_frames_per_launch": frames}
    if calib:
        ff = calib.get("fetch_factor", {})
        width = {"k_pad_level0": "16B", "k_resize_level": "4B", "k_fast_score": "4B", "k_fast_cells": "4B", "k_fast_cells_list": "1B", "k_gauss7": "4B",
                 "k_describe": "segments_40_of_704", "k_octree": "4B", "k_octree_gauss": "4B", "k_assemble": "4B", "k_knn2_mfma": "16B", "k_pyr_tiles": "4B"}
        doc["_fetch_factor"] = {k: ff[w] for k, w in width.items() if w in ff}
        # k_describe reads two planes of equal footprint: the row-major un-blurred one in 32-byte row segments (requests tallied exactly)
        # and the tiled blurred one in whole 128-byte lines (tallied at one half): the mean of the two factors
        if "segments_40_of_704" in ff and "16B" in ff:
            doc["_fetch_factor"]["k_describe"] = round((ff["segments_40_of_704"] + ff["16B"]) / 2, 4)
        doc["_fetch_factor"]["_default"] = ff.get("4B", 2.0)
        doc["_load_width"] = width
    doc["kernels"] = out
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
