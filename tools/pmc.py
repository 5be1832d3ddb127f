#!/usr/bin/env python3
"""Counter passes of one command, folded into one json: every pass is its own `rocprofv3 --kernel-trace --pmc ...` run (never
combined with the tracing domains gpurun refuses), per kernel the mean counter value per dispatch.
   python tools/pmc.py OUT.json "SQ_WAVE_CYCLES SQ_WAIT_ANY ..." ["TCC_HIT_sum TCC_MISS_sum" ...] -- python3 tools/ring_bench.py 1024 512 1
Derived (when the counters are there): valu_busy_pct (VALUBusy), wait_any / wait_inst / active shares of SQ_WAVE_CYCLES,
l2_hit_rate, traffic_bytes (FETCH_SIZE + WRITE_SIZE, KB x 1024 as rocprofv3 reports them)."""
import csv, glob, json, os, re, shutil, subprocess, sys, tempfile
from collections import defaultdict

args = sys.argv[1:]
sep = args.index("--")
out_path, passes, cmd = args[0], args[1:sep], args[sep + 1:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
os.environ.setdefault("TMPDIR", "/tmp")
for k, counters in enumerate(passes):
    d = tempfile.mkdtemp(prefix="pmc", dir="/tmp")
    # every pass under a timeout, in its own process group: a hung workload is killed with the profiler (it must not outlive the pass and
    # run into the next one's counters); the program after `--` stays the interpreter itself (no shell, no env hop)
    import signal
    pr = subprocess.Popen(["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
                          cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        _, err = pr.communicate(timeout=float(os.environ.get("AVRF_PMC_TIMEOUT", "900")))
    except subprocess.TimeoutExpired:
        try:
            os.killpg(pr.pid, signal.SIGKILL)                      # the exact group this pass started
        except ProcessLookupError:
            pass
        _, err = pr.communicate()
        err = (err or "") + "\n(timed out: process group killed)"
    if pr.returncode != 0:
        print("pass", k, "failed:", (err or "")[-800:], file=sys.stderr)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("avrf::", "").strip()
            a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    shutil.rmtree(d, ignore_errors=True)
out = {"command": "rocprofv3 --kernel-trace --pmc <pass> -- " + " ".join(cmd), "passes": passes, "unit": "mean counter value per dispatch", "kernels": {}}
for name, cs in acc.items():
    v = {c: round(x[0] / max(1, x[1]), 2) for c, x in cs.items()}
    v["dispatches"] = max(x[1] for x in cs.values())
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        for c, key in (("SQ_WAIT_ANY", "wait_any_share"), ("SQ_WAIT_INST_ANY", "wait_inst_share"), ("SQ_ACTIVE_INST_ANY", "active_share"),
                       ("SQ_ACTIVE_INST_VALU", "active_valu_share")):
            if c in v:
                v[key] = round(v[c] / wc, 4)
    if "TCC_HIT_sum" in v and v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0) > 0:
        v["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0)), 4)
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        v["traffic_bytes"] = int((v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024)
    out["kernels"][name] = v
acc_te = [k for k in out["kernels"] if "k_accumulate" in k and "TeCurve" in k and "traffic_bytes" in out["kernels"][k]]
if acc_te:
    out["k_accumulate_hbm_bytes_per_launch"] = out["kernels"][acc_te[0]]["traffic_bytes"]
json.dump(out, open(out_path, "w"), indent=1)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", kv[1].get("FETCH_SIZE", 0))) * kv[1]["dispatches"])[:12]
for name, v in top:
    print(name[:60].ljust(60), {k: v[k] for k in v if k in ("dispatches", "VALUBusy", "wait_any_share", "wait_inst_share", "active_share", "active_valu_share", "l2_hit_rate", "traffic_bytes")})
