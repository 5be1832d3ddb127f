#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 --kernel-trace run (the multi-context default bench): for each kernel name the union of its
dispatch intervals, how much of it overlaps the union of k_accumulate, the mean number of dispatches in flight, and the gaps.
   python tools/timeline.py <dir with *kernel_trace.csv> [window_ms]      (the busiest window of that length is analysed)"""
import csv, glob, re, sys
from collections import defaultdict

def union(iv):
    iv = sorted(iv); out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out

def length(u): return sum(b - a for a, b in u)

def inter(u, v):
    i = j = 0; tot = 0
    while i < len(u) and j < len(v):
        a = max(u[i][0], v[j][0]); b = min(u[i][1], v[j][1])
        if b > a: tot += b - a
        if u[i][1] < v[j][1]: i += 1
        else: j += 1
    return tot

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("avrf::", "").strip()
        name = re.sub(r"<.*", "", name)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
# the busiest window of W ms (default 600): the timed region of the bench, not its warm-up / latency / end-to-end phases
W = (float(sys.argv[2]) if len(sys.argv) > 2 else 600.0) * 1e6
acc_iv = sorted((a, b) for a, b, n in rows if n == "k_accumulate")
best, j, run = (0, t0), 0, 0                                  # the window in which k_accumulate accumulates the most running time
for i, (a, b) in enumerate(acc_iv):
    run += b - a
    while acc_iv[j][0] < a - W:
        run -= acc_iv[j][1] - acc_iv[j][0]; j += 1
    if run > best[0]: best = (run, acc_iv[j][0])
lo, hi = best[1], best[1] + W
rows = [(max(a, lo), min(b, hi), n) for a, b, n in rows if b > lo and a < hi]
wall = hi - lo
by = defaultdict(list)
for a, b, n in rows: by[n].append((a, b))
acc = union(by.get("k_accumulate", []))
allu = union([(a, b) for a, b, _ in rows])
print(f"window {wall/1e6:.2f} ms, {len(rows)} dispatches; some kernel running {length(allu)/wall:.3f} of the time; k_accumulate running {length(acc)/wall:.3f}")
# in-flight count
ev = sorted([(a, 1) for a, b, _ in rows] + [(b, -1) for a, b, _ in rows])
cur = 0; last = lo; area = 0
for t, d in ev:
    area += cur * (t - last); last = t; cur += d
print(f"mean dispatches in flight {area/wall:.2f}")
ev = sorted([(a, 1) for a, b in by.get('k_accumulate', [])] + [(b, -1) for a, b in by.get('k_accumulate', [])])
cur = 0; last = lo; hist = defaultdict(int)
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("k_accumulate dispatches in flight: " + ", ".join(f"{k}: {v/wall:.3f}" for k, v in sorted(hist.items())))
print(f"{'kernel':28s} {'calls':>6s} {'mean us':>9s} {'union/wall':>10s} {'inside k_acc':>12s}")
for n, iv in sorted(by.items(), key=lambda kv: -length(union(kv[1]))):
    u = union(iv)
    print(f"{n[:28]:28s} {len(iv):6d} {sum(b-a for a,b in iv)/len(iv)/1e3:9.1f} {length(u)/wall:10.3f} {inter(u, acc)/max(1,length(u)):12.3f}")
