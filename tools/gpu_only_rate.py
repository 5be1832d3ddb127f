#!/usr/bin/env python3
"""Experiment: the headline workload with the host weight hash switched off (AVRF_EXPERIMENT_SKIP_HASH=1: wrong weights, the
verdicts are ignored) -- what the GPU side alone sustains with S contexts.  Tells how far the host hash is from mattering.
The switch is NOT in the product source: apply tools/patches/experiment_skip_hash.patch to a scratch copy of the tree and build that
(`git apply tools/patches/experiment_skip_hash.patch && make -C ark_vrf_amd/csrc EXTRA=-DAVRF_EXPERIMENTS OUT=/tmp/libavrf_exp.so
OBJDIR=/tmp/obj_exp && git checkout ark_vrf_amd/csrc/capi.hip`, then AVRF_LIB_PATH=/tmp/libavrf_exp.so); against the shipped build this
script measures the ordinary pipeline."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import bench  # noqa: E402
from ark_vrf_amd import _native as nat  # noqa: E402

n = 65536
for streams in [int(x) for x in sys.argv[1:]] or [4, 8, 16]:
    ctxs = [nat.Context(0) for _ in range(streams)]
    batches = [bench.make_batch(c, nat, n, start=i * n) for i, c in enumerate(ctxs)]
    for c, (b, _) in zip(ctxs, batches):
        assert c.thin_batch_stage(b) == 0
        c.thin_batch_run()
    stop = time.perf_counter() + 1.0
    counts = [0] * streams

    def worker(i):
        while time.perf_counter() < stop:
            ctxs[i].thin_batch_run(); counts[i] += 1
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(streams)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"{streams} contexts: {sum(counts)} batches in {dt:.2f} s -> {sum(counts) * n / dt / 1e6:.1f} M items/s, {dt / sum(counts) * 1e3:.3f} ms per batch", flush=True)
    for c in ctxs:
        c.close()
