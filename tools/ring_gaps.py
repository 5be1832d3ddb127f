#!/usr/bin/env python3
"""Device idle time inside the LAST burst of kernels of a rocprofv3 --kernel-trace run (a ring-proving pass of tools/ring_bench.py):
   python tools/ring_gaps.py <dir with *kernel_trace.csv> [burst gap ms = 3]
prints the burst's span, the union of its dispatch intervals, and every idle gap above 0.1 ms with the kernels on either side."""
import csv, glob, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
split = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 3e6
# bursts: maximal runs of dispatches with no idle gap above `split`
bursts, cur, end = [], [], None
for a, b, n in rows:
    if end is not None and a - end > split:
        bursts.append(cur); cur = []
    cur.append((a, b, n)); end = b if end is None else max(end, b)
    if not cur[:-1]: end = b
bursts.append(cur)
big = [b for b in bursts if len(b) > 50]
for bi, burst in enumerate(big[-3:]):
    t0 = burst[0][0]; t1 = max(b for _, b, _ in burst)
    busy = 0; e = t0; gaps = []
    last = None
    for a, b, n in burst:
        if a > e:
            if a - e > 1e5: gaps.append((e - t0, a - e, last, n))
            busy += b - a; e = b
        elif b > e:
            busy += b - e; e = b
        if last is None or b >= e: last = n
    print(f"burst {bi}: {len(burst)} dispatches, span {(t1 - t0) / 1e6:.2f} ms, device busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
    for at, g, p, n in gaps:
        print(f"   at {at / 1e6:7.2f} ms idle {g / 1e6:5.2f} ms   after {p}   before {n}")
    from collections import defaultdict
    tot = defaultdict(lambda: [0, 0])
    for a, b, n in burst:
        tot[n][0] += b - a; tot[n][1] += 1
    for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"   {t / 1e6:7.2f} ms x{c:3d}  {n}")
