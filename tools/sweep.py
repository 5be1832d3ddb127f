#!/usr/bin/env python3
"""Sweeps MSM plan parameters (window bits, segment length) on the bench workload and prints
per-step timing breakdowns.  Usage: python tools/sweep.py [items]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ark_vrf_amd import _native as nat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = nat.Context(0, 0)
batch, raw = bench.make_batch(ctx, nat, n, 0)
assert ctx.thin_batch_stage(batch) == 0
for c in [11, 12, 13, 14, 15, 16]:
    for seg in [8, 16, 32]:
        os.environ["AVRF_MSM_C"] = str(c)
        os.environ["AVRF_MSM_SEG"] = str(seg)
        for _ in range(2):
            assert ctx.thin_batch_run() == 0
        nat.lib().avrf_kernel_stats(ctx._h, 1, None, None, None)
        t0 = time.perf_counter()
        K = 5
        for _ in range(K):
            assert ctx.thin_batch_run() == 0
        dt = (time.perf_counter() - t0) / K
        ms, cnt = C.c_double(0), C.c_uint64(0)
        nat.lib().avrf_kernel_stats(ctx._h, 0, C.byref(ms), C.byref(cnt), None)
        tm = ctx.last_timing()
        print(f"c={c:2d} seg={seg:2d}  step {dt*1e3:7.2f} ms  accumulate {ms.value/cnt.value:6.3f} ms  hash(host) {tm[2]/1e3:5.2f}  msm(total) {tm[4]/1e3:6.2f} ms", flush=True)
