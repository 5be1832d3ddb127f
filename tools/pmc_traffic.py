#!/usr/bin/env python3
"""Folds two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) into profiles/<name>.json:
   tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<command>"
Per kernel: average KB per dispatch of each counter; plus k_accumulate's bytes per launch as bench.py's
roofline.traffic reads it (FETCH + WRITE, KB * 1024)."""
import csv, glob, json, re, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            a = acc[name]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"command": sys.argv[4],
       "unit": "KB per dispatch as reported by rocprofv3 (TCC_EA0 request counters; Infinity-Cache hits are counted; no x2 correction "
               "applied: the accumulate kernel's reads are 96-byte per-lane gathers, not 16 B/lane coalesced streams)",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0]), write.get(k, [0, 0])
    out["kernels"][k] = {"FETCH_SIZE_KB_avg": round(f[0] / max(1, f[1]), 1), "WRITE_SIZE_KB_avg": round(w[0] / max(1, w[1]), 1), "dispatches": max(f[1], w[1])}
acc = [k for k in out["kernels"] if "k_accumulate" in k and "TeCurve" in k]
if acc:
    v = out["kernels"][acc[0]]
    out["k_accumulate_hbm_bytes_per_launch"] = int((v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"]) * 1024)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
