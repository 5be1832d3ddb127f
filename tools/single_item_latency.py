#!/usr/bin/env python3
"""Latency of ONE Thin-VRF verification / proof through the C ABI (BASELINE configs[0]; reference 188 / 182 us on a CPU core,
benches/SUMMARY.md:53-54), and of small calls: python tools/single_item_latency.py [suite]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from helpers import nat_batch
from ark_vrf_amd import _native as nat
suite = int(sys.argv[1]) if len(sys.argv) > 1 else 0
c = nat.Context(suite)
for n in (1, 2, 16, 64, 512, 2048):
    b = orc.gen_batch(suite, 0, n)
    vb, pb = nat_batch(b), nat_batch(b, with_sks=True, with_proofs=False)
    for _ in range(3):
        assert c.thin_verify(vb) == [0] * n and c.thin_prove(pb) == b["proofs"]
    def best(fn, reps=30):
        t = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); t = min(t, time.perf_counter() - t0)
        return t
    print(f"suite {suite} n={n}: thin verify {best(lambda: c.thin_verify(vb)) * 1e3:.3f} ms, thin prove {best(lambda: c.thin_prove(pb)) * 1e3:.3f} ms", flush=True)

for n in (1, 64):
    b = orc.gen_batch(suite, 1, n)
    pb = nat_batch(b, with_sks=True, with_proofs=False); vb = nat_batch(dict(b, pks_xy=b""))
    for _ in range(3):
        assert c.pedersen_prove(pb)[0] == b["proofs"] and c.pedersen_verify(vb) == [0] * n
    print(f"suite {suite} n={n}: pedersen verify {best(lambda: c.pedersen_verify(vb)) * 1e3:.3f} ms, pedersen prove {best(lambda: c.pedersen_prove(pb)) * 1e3:.3f} ms", flush=True)
