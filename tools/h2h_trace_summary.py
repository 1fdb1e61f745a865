#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/h2h_trace.py: busy time of copies per direction, of kernels, and
of their union over the traced span.   python tools/h2h_trace_summary.py DIR"""
import csv, glob, os, sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


d = sys.argv[1]
kern, copies = [], defaultdict(list)
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        copies[r["Direction"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
# steady state: from the start of the 9th host-to-device copy (3 warm-up jobs x 2 chunks + the first two timed chunks) to the last copy's end
h2d = sorted(copies.get("MEMORY_COPY_HOST_TO_DEVICE", []))
big = [c for c in h2d if c[1] - c[0] > 200000]   # the frame uploads (42 MB), not the small table copies
lo, t1 = big[8][0], max(e for _, e in big)
span = t1 - lo
njobs = (len(big) - 8) / 2.0
kk = [(max(s, lo), min(e, t1)) for s, e, _ in kern if e > lo and s < t1]
print("window %.2f ms = %.3f ms per job (%d uploads)" % (span / 1e6, span / 1e6 / njobs, len(big) - 8))
print("kernels busy (union) %.2f ms = %.0f %%" % (union(kk) / 1e6, 100 * union(kk) / span))
allc = []
for k, v in copies.items():
    vv = [(max(s, lo), min(e, t1)) for s, e in v if e > lo and s < t1]
    allc += vv
    print("copies %-28s n=%4d  sum %.2f ms  union %.2f ms = %.0f %%  mean %.1f us" % (k, len(vv), sum(e - s for s, e in vv) / 1e6, union(vv) / 1e6, 100 * union(vv) / span, sum(e - s for s, e in vv) / max(len(vv), 1) / 1e3))
print("copies any direction (union) %.2f ms = %.0f %%" % (union(allc) / 1e6, 100 * union(allc) / span))
print("copies or kernels (union) %.2f ms = %.0f %%" % (union(allc + kk) / 1e6, 100 * union(allc + kk) / span))
per = defaultdict(list)
for s, e, n in kern:
    if e > lo and s < t1:
        per[n.replace("void ", "").replace("uvo::", "")[:28]].append(e - s)
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("  %-28s n=%5d  mean %.1f us  total %.2f ms" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
