#!/usr/bin/env python3
"""Two windows of one kernel trace of bench.py (the resident headline loop, then the H2D-inclusive loop): batches per second, device busy
union, mean duration and summed time of every kernel per batch in each.   python legs_compare.py <dir> """
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("avrf::", "")[:34]))
rows.sort()
acc = [r for r in rows if r[2].startswith("k_accumulate")]
t_end = acc[-1][1]
# the e2e loop is the last burst of k_accumulate launches; the resident loop the one before (separated by a gap > 20 ms)
bursts = []; cur = [acc[0]]
for a in acc[1:]:
    if a[0] - cur[-1][1] > 20e6: bursts.append(cur); cur = []
    cur.append(a)
bursts.append(cur)
big = [b for b in bursts if len(b) > 300]
for name, b in (("resident", big[-2]), ("from host", big[-1])):
    t0, t1 = b[len(b) // 4][0], b[-len(b) // 8][1]            # the steady middle
    sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    nb = sum(1 for r in sel if r[2].startswith("k_accumulate"))
    busy = 0; e = 0
    for a_, b_, _ in sel:
        if a_ > e: busy += b_ - a_; e = b_
        elif b_ > e: busy += b_ - e; e = b_
    tot = defaultdict(lambda: [0, 0])
    for a_, b_, n in sel: tot[n][0] += b_ - a_; tot[n][1] += 1
    print(f"{name}: {nb} batches in {(t1 - t0) / 1e6:.1f} ms = {(t1 - t0) / 1e3 / nb:.1f} us per batch; device busy {busy / (t1 - t0):.3f}; kernel time per batch {sum(v[0] for v in tot.values()) / nb / 1e3:.0f} us")
    for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:13]:
        print(f"    {n:36s} {t / nb / 1e3:7.1f} us per batch, {c / nb:5.2f} launches per batch, mean {t / c / 1e3:7.1f} us")
