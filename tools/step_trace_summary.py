#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace run of the default bench command: how much of the steady-state span has 0 / 1 / 2+
kernels in flight, per-queue busy time and the gaps between consecutive kernels of a queue.
   rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 bench.py --no-cpu-baseline --no-subrecords --no-verify
   python tools/step_trace_summary.py /tmp/st"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uvo::", "")
        if not n.startswith("k_"):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo, hi = t0 + (t1 - t0) * 0.3, t0 + (t1 - t0) * 0.9   # steady state: the timed steps, away from warm-up and drain
win = [(max(s, lo), min(e, hi), n, q, st) for s, e, n, q, st in rows if e > lo and s < hi]
span = hi - lo
ev = []
for s, e, *_ in win:
    ev.append((s, 1)), ev.append((e, -1))
ev.sort()
depth, last, hist = 0, lo, defaultdict(float)
for t, dlt in ev:
    hist[min(depth, 4)] += t - last
    depth += dlt
    last = t
hist[min(depth, 4)] += hi - last
nfs = sum(1 for _, _, n, *_ in win if n == "k_fast_score")
print("span %.2f ms, %d k_fast_score launches: %.3f ms per launch" % (span / 1e6, nfs, span / 1e6 / max(nfs, 1)))
print("kernels in flight:  " + "  ".join("%d%s: %.1f %%" % (k, "+" if k == 4 else "", 100 * v / span) for k, v in sorted(hist.items())))
byq = defaultdict(list)
for s, e, n, q, st in win:
    byq[(q, st)].append((s, e, n))
for key, v in sorted(byq.items()):
    v.sort()
    busy = sum(e - s for s, e, _ in v)
    gaps = [(v[i + 1][0] - v[i][1], v[i][2], v[i + 1][2]) for i in range(len(v) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print("queue %s stream %s: %d kernels, busy %.1f %%, gaps: n %d sum %.1f %% mean %.1f us" % (key[0], key[1], len(v), 100 * busy / span, len(pos), 100 * sum(g[0] for g in pos) / span, sum(g[0] for g in pos) / max(len(pos), 1) / 1e3))
    agg = defaultdict(list)
    for g, a, b in pos:
        agg[a + " -> " + b].append(g)
    for k, gs in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("      %-44s n %4d mean %6.1f us  total %.2f ms" % (k, len(gs), sum(gs) / len(gs) / 1e3, sum(gs) / 1e6))
