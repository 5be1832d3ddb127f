#!/usr/bin/env python3
"""The ring leg as bench.py measures it (4 contexts sharing the GPU, sharded by index), prove + batch verify only:
   python tools/ring4_bench.py [suite 0|1] [ring_size] [n_proofs] [reps] [contexts]      -- for A/B runs under AVRF_RING_* environment knobs"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ark_vrf_amd import _native as nat  # noqa: E402

suite = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ring = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n_ctx = int(sys.argv[5]) if len(sys.argv) > 5 else 4
if os.environ.get("AVRF_RING4_CPUS"):         # the per-rank CPU share of an 8-rank node (bench.py projected_child)
    os.sched_setaffinity(0, bench.pick_cpus(int(os.environ["AVRF_RING4_CPUS"])))
import torch  # noqa: E402
torch.cuda.set_device(0)                     # (torch's runtime first, as bench.py's ranks do; then the library's device flags)
assert nat.set_blocking_sync(0, True) == 0
m = bench.ring_metrics(nat, 0, n_proofs=n, ring_size=ring, n_ctx=n_ctx, suite=suite, quick=True, reps=reps)
print(json.dumps({"proofs_per_sec": round(m["ring_vrf_proofs_per_sec"]), "batch_verifications_per_sec": round(m["ring_vrf_batch_verifications_per_sec"]),
                  "prove_passes_s": m["passes"]["prove"]["pass_times_s"]}))
