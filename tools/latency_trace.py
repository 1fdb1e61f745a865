#!/usr/bin/env python3
"""Per-call timeline of the batch-1 path from a rocprofv3 --kernel-trace run of tools/latency.py (UVO_LAT_TRACE=1): for the calls of the
device-resident loop, every kernel's start relative to the call's first kernel, its duration and the gap in front of it (medians over
the calls), and the span first kernel start -> last kernel end.
   UVO_LAT_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 tools/latency.py;  python3 tools/latency_trace.py /tmp/lt"""
import csv, glob, json, os, sys
import numpy as np

d = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uvo::", "").split("<")[0]
        if not n.startswith("k_"):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
calls, cur = [], []
for r in rows:
    cur.append(r)
    if r[2] == "k_describe":
        calls.append(cur)
        cur = []
calls = calls[-250:]    # the device-resident loop (the host-buffer calls come first)
sig = [tuple(k[2] for k in c) for c in calls]
common = max(set(sig), key=sig.count)
calls = [c for c, s in zip(calls, sig) if s == common]
out = []
for i, name in enumerate(common):
    st = np.array([c[i][0] - c[0][0] for c in calls]) / 1e3
    du = np.array([c[i][1] - c[i][0] for c in calls]) / 1e3
    gap = np.array([c[i][0] - max(k[1] for k in c[:i]) if i else 0 for c in calls]) / 1e3
    streams = sorted({c[i][3] for c in calls})
    out.append({"kernel": name, "start_us": round(float(np.median(st)), 1), "dur_us": round(float(np.median(du)), 1), "gap_before_us": round(float(np.median(gap)), 1), "stream": ",".join(streams)})
span = np.array([max(k[1] for k in c) - c[0][0] for c in calls]) / 1e3
period = np.diff(np.array([c[0][0] for c in calls])) / 1e3
print(json.dumps({"calls": len(calls), "span_us_median": round(float(np.median(span)), 1), "call_period_us_median": round(float(np.median(period)), 1),
                  "sum_dur_us": round(sum(o["dur_us"] for o in out), 1), "sum_gap_us": round(sum(o["gap_before_us"] for o in out), 1), "timeline": out}))
