#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel.

usage: pmc_summary.py [--config N] DIR [DIR ...]   (each DIR is searched recursively for *_counter_collection.csv)
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
    if args and args[0] == "--config":
        config = int(args[1])
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
            "FETCH_SIZE and WRITE_SIZE in KB as reported -- on gfx950 FETCH_SIZE under-reports reads by 2x (MI355X_MICROARCH.md), so HBM read "
            "bytes ~= 2 * FETCH_SIZE_KB * 1024; k_resize_level is the mean over its 7 launches per step; collected with tools/profile_round.sh" % config)
    frames = {2: 257, 3: 129}.get(config)   # bench.py's default batch of the config + its halo frame
    print(json.dumps({"_note": note, "_config": config, "_frames_per_launch": frames, "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
