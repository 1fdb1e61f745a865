#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel.

usage: pmc_summary.py DIR [DIR ...]   (each DIR is searched recursively for *_counter_collection.csv)
Prints, per kernel, the mean per-launch value of every counter found, plus VGPR / LDS use and launch count.
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
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].split("(")[0]
                    if "uvo::" not in name:
                        continue
                    name = name.split("uvo::")[-1]
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    meta[name] = {"vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]), "lds": int(row["LDS_Block_Size"])}
    out = {}
    for k, ctrs in sorted(acc.items()):
        out[k] = dict(meta[k])
        for c, vals in sorted(ctrs.items()):
            out[k][c] = sum(vals) / len(vals)
            out[k]["launches"] = len(vals)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
