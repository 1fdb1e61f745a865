#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes over tools/ubench/stream_read into calibration factors: bytes moved / (counter * 1024).
Streaming kernels move exactly the bytes they read; the row-segment kernel (k_describe's pattern: 40 bytes of every 704) moves one
64-byte line per segment -- its factor says how FETCH_SIZE counts 64-byte requests (measured 1.0: a request is tallied at 64 bytes
whatever its size, which is also why full 128-byte streaming requests read as one half).
usage: fetch_calibration.py DIR_FETCH DIR_WRITE  -> JSON {"fetch_factor": {"1B": .., "4B": .., "8B": .., "16B": .., "segments_40_of_704": ..},
"write_factor": {...}}.  bench.py / tools/pmc_summary.py multiply a kernel's FETCH_SIZE by the factor of its load width."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

BYTES = 1 << 30
KNOWN = {"k_read<unsigned char>": ("1B", BYTES), "k_read<unsigned int>": ("4B", BYTES), "k_read<HIP_vector_type<unsigned int, 2u>>": ("8B", BYTES),
         "k_read<HIP_vector_type<unsigned int, 4u>>": ("16B", BYTES), "k_read_segments": ("segments_40_of_704", (BYTES // 704) * 64),   # one 64-byte line per 40-byte row segment (pitch 704 = 11 lines)
         "k_write<unsigned char>": ("1B", BYTES), "k_write<unsigned int>": ("4B", BYTES), "k_write<HIP_vector_type<unsigned int, 4u>>": ("16B", BYTES)}


def collect(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace(" >", ">")
            acc[name].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"_note": "tools/ubench/stream_read.hip: 1 GiB streamed once per kernel; factor = known bytes / (counter KB * 1024)", "fetch_factor": {}, "write_factor": {},
           "raw_KB": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}}
    for name, (label, nbytes) in KNOWN.items():
        if name.startswith("k_read") and name in fetch and fetch[name] > 0:
            out["fetch_factor"][label] = round(nbytes / (fetch[name] * 1024), 4)
        if name.startswith("k_write") and name in write and write[name] > 0:
            out["write_factor"][label] = round(nbytes / (write[name] * 1024), 4)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
