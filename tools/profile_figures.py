#!/usr/bin/env python3
"""The figures DESIGN.md / README.md quote, read from profiles/<round>_*: python tools/profile_figures.py [r5]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r5"
P = lambda n: os.path.join(ROOT, "profiles", f"{R}_{n}")
d = json.load(open(P("bench_default.json")))
r, am, h = d["roofline"], d["additional_metrics"], d["host"]
M = lambda v: round(v / 1e6, 1)
print("head", d["head"][:8], "| cpu", h["cpu_model"], "| threads", h["host_threads_per_rank"], "slots", h["slots_per_rank"], "lanes", h["lanes_per_rank"])
print(f"headline {M(d['value'])} M/s, {d['ms_per_step']:.3f} ms per batch; e2e {M(d['e2e_value'])} M/s; host hash {h['host_cpu_us_per_step']['hash']:.0f} us of a core per batch")
print(f"k_accumulate {r['kernel_avg_ms']:.4f} ms = {r['achieved']:.2f} of {r['peak']:.2f} T mad/s = {r['frac']:.3f}; bare loop {r['madd_loop']['achieved_Gadd_per_s']:.1f} of {r['madd_loop']['isolated_madd_Gadd_per_s']:.1f} G add/s = {r['madd_loop']['frac']:.2f}; "
      f"mads/add {r['multiply_adds_per_mixed_addition']}; HBM {r['hbm']['achieved']:.0f} GB/s = {r['hbm']['frac']:.4f}; counters {r['traffic'] / 1e6:.0f} MB per launch")
s = d["step_breakdown_us"]["one_context_alone"]
print(f"one batch alone: {s['total'] / 1e3:.2f} ms = host chain {s['host_weight_transcript'] / 1e3:.2f} + prepare {s['device_prepare_hash'] / 1e3:.2f} + msm {s['device_msm'] / 1e3:.2f}")
print(f"cpu_baseline {d['cpu_baseline']['value'] / 1e3:.1f} k/s one thread; {d['cpu_baseline_all_cores']['value'] / 1e3:.0f} k/s on {d['cpu_baseline_all_cores']['cores']}")
p = d["projected_8gpu"]
print(f"projected: thin {M(p['thin']['value'])} M/s = {p['thin']['frac_of_1gpu_rate']:.2f} -> {p['thin']['predicted_speedup_8gpu']:.2f}x ({p['thin']['plan']}); ring share {p['ring']['t_prove_s'] * 1e3:.1f} ms = "
      f"{p['ring']['ring_vrf_proofs_per_sec'] / 1e3:.1f} k/s -> {p['ring']['predicted_speedup_8gpu']:.2f}x")
print(f"ring: {am['ring_vrf_proofs_per_sec'] / 1e3:.2f} k proofs/s (passes {am['passes']['prove']['pass_times_s']}), batch verify {am['ring_vrf_batch_verifications_per_sec'] / 1e3:.0f} k/s "
      f"(passes {am['passes']['batch_verify']['pass_times_s']}), independent {am['ring_vrf_independent_verifications_per_sec'] / 1e3:.0f} k/s, one {am['ring_verify_single_ms']:.2f} ms, index {am['ring_index_ms']:.1f} ms")
print(f"ring roofline {am['roofline']['achieved']:.2f} of {am['roofline']['peak']:.2f} G add/s = {am['roofline']['frac']:.2f}")
c4 = am["configs4_shape"]
print(f"bn254 ring: {c4['ring_vrf_proofs_per_sec'] / 1e3:.2f} k proofs/s, batch {c4['ring_vrf_batch_verifications_per_sec'] / 1e3:.0f} k/s, independent {c4['ring_vrf_independent_verifications_per_sec'] / 1e3:.0f} k/s; "
      f"roofline {c4['roofline']['achieved']:.2f} of {c4['roofline']['peak']:.2f} = {c4['roofline']['frac']:.2f}")
c2 = am["configs2_shape"]
print(f"pedersen: prove {M(c2['pedersen_proofs_per_sec'])}, verify {M(c2['pedersen_verifications_per_sec'])}, batch {M(c2['pedersen_batch_verifications_per_sec'])}; pool resident "
      f"{M(c2['pedersen_batch_resident']['pedersen_batch_verifications_per_sec'])}, pinned {M(c2['pedersen_batch_resident']['from_pinned_host_buffers_per_sec'])} M/s")
n1 = am["configs0_shape"]["n=1"]
print("n=1 ms:", {k: round(v, 3) for k, v in n1.items()})
print("validate (bench line):", {k: M(v) for k, v in am["validate_yes"].items() if isinstance(v, float)})
print("validate_bench:", {k: M(v) for k, v in json.load(open(P("validate_bench.json"))).items() if isinstance(v, (int, float))})
# (the measurement kernels of libavrf_probe.so -- the clock probe's sleeping wave, the instruction streams, the bare addition loops -- and
# the once-per-SRS table builds are listed in the CSVs but are not part of a step / a proof: shares below are of the rest)
SKIP = ("k_clock", "k_stream", "k_madd_loop", "k_g1_multiples", "k_g1_table", "k_fixed_table")
for f in ("single_context", "per_item", "ring_prove_2048", "ring_bn254_prove_1024"):
    rows = [x for x in csv.DictReader(open(P(f + "_kernel_stats.csv"))) if not any(t in x["Name"] for t in SKIP)]
    tot = sum(int(x["TotalDurationNs"]) for x in rows) or 1
    nm = lambda x: x["Name"].replace("void avrf::", "").replace("avrf::", "").split("(")[0]
    short = lambda x: nm(x).split("<")[0] + ("<G1>" if "G1" in nm(x) and "k_accumulate<" in nm(x) else "")
    print(f + ": " + "; ".join(f"{short(x)} {float(x['AverageNs']) / 1e3:.1f} us x {x['Calls']} ({100.0 * int(x['TotalDurationNs']) / tot:.1f}%)" for x in rows[:12]))
for name in ("pmc_thin", "pmc_ring"):
    k = json.load(open(P(name + ".json")))["kernels"]
    for n, v in k.items():
        if "k_accumulate" in n:
            print(name, n[:40], "VALUBusy", v["VALUBusy"], "wait_any", v["wait_any_share"], "wait_inst", v["wait_inst_share"], "L2 hit", v.get("l2_hit_rate"), "traffic MB", round(v["traffic_bytes"] / 1e6))
print(open(P("latency_report.txt")).read())
