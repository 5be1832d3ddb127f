#!/usr/bin/env python3
"""The pool fed with serialize_compressed bytes (avrf_pool_submit_wire) at BASELINE configs[1] size, alone, for kernel traces:
   python tools/pool_wire_bench.py [validate 0|1] [seconds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ark_vrf_amd import _native as nat  # noqa: E402
import oracle as orc  # noqa: E402

v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
n = 65536
b = orc.gen_batch(0, 0, n, threads=16)
comp = lambda xy_all: b"".join(orc.point_compress(0, xy_all[64 * i: 64 * i + 64]) for i in range(len(xy_all) // 64))
pks_c = comp(b["pks_xy"]); ios_c = comp(b["ios_xy"])
pr = b["proofs"]
proofs_c = b"".join(orc.point_compress(0, pr[96 * j: 96 * j + 64]) + pr[96 * j + 64: 96 * j + 96] for j in range(n))
pool = nat.Pool(0, kind=1, slots=48, lanes=10, threads=6, hash_group=8)
pw = nat.PinnedBatch(n, ios_c, b["io_counts"], b["ads"], b["ad_lens"], pks_xy=pks_c, proofs=proofs_c)
tk = [pool.submit_wire(pw, validate=v) for _ in range(48)]
assert all(pool.wait(t) == 0 for t in tk)
done, mism, sec = pool.cycle(steps_block=96, min_seconds=secs, from_host=True, expect=0)
assert mism == 0
print(f"pool wire validate={v}: {done * n / sec / 1e6:.2f} M items/s ({sec / done * 1e3:.3f} ms per batch)")
pool.close()
