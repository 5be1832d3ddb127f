#!/usr/bin/env python3
"""BASELINE configs[2] shape: Pedersen-VRF Bandersnatch, 65 536 independent items on one context:
prove (a5), independent verify (a6), one (5N+2)-term batch verification (a7); plus Thin prove / verify (a3, a4)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ark_vrf_amd import _native as nat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
suite = int(sys.argv[2]) if len(sys.argv) > 2 else 0
R_ORDER = {0: bench.R_BANDERSNATCH, 1: 2736030358979909402780800718157159386076813972158567259200215660948447373041,
           2: 6554484396890773809930967563523245729705921265872317281365359162392183254199, 3: 2 ** 252 + 27742317777372353535851937790883648493,
           4: bench.R_BANDERSNATCH}
bench.R_BANDERSNATCH = R_ORDER[suite]
ctx = nat.Context(suite)
print(f"suite {suite}, {n} items, one context")
sks = bench.derive_scalars(b"ped-bench-sk", 0, n, bench.R_BANDERSNATCH)
pks = ctx.scalar_mul_base(sks)
inputs = ctx.scalar_mul_base(bench.derive_scalars(b"ped-bench-in", 0, n, bench.R_BANDERSNATCH))
outs = ctx.scalar_mul(sks, inputs)
ios = b"".join(inputs[64 * j: 64 * j + 64] + outs[64 * j: 64 * j + 64] for j in range(n))
ads = [b"ad-%d" % j for j in range(n)]
adb, adl = b"".join(ads), [len(a) for a in ads]


def timed(f, reps=3):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t) / reps, r


import ctypes as C  # noqa: E402
o256, o96, o32, st32 = (C.c_uint8 * (256 * n))(), (C.c_uint8 * (96 * n))(), (C.c_uint8 * (32 * n))(), (C.c_int32 * n)()
dev = lambda: ctx.last_timing()[0] / 1e3                            # launch .. results on the host, ms
pb = nat.Batch(n, ios, [1] * n, adb, adl, pks_xy=pks, sks=sks)
ped, blind = ctx.pedersen_prove(pb)
t, rc = timed(lambda: ctx.pedersen_prove_into(pb, o256, o32))
assert rc == 0 and bytes(o256) == ped
print(f"pedersen prove        {n / t:12.0f} /s  ({t * 1e3:.2f} ms per {n} from host buffers; {dev():.2f} ms after staging)")
vb = nat.Batch(n, ios, [1] * n, adb, adl, proofs=ped)
t, rc = timed(lambda: ctx.pedersen_verify_into(vb, st32))
print(f"pedersen verify       {n / t:12.0f} /s  ({t * 1e3:.2f} ms; {dev():.2f} ms after staging), all ok: {rc == 0 and not any(st32)}")
ctx.pedersen_batch_stage(vb)
t, st = timed(lambda: ctx.pedersen_batch_run())
print(f"pedersen batch verify {n / t:12.0f} /s  ({t * 1e3:.2f} ms), status {st}")
proofs = ctx.thin_prove(pb)
t, rc = timed(lambda: ctx.thin_prove_into(pb, o96))
assert rc == 0 and bytes(o96) == proofs
print(f"thin prove            {n / t:12.0f} /s  ({t * 1e3:.2f} ms; {dev():.2f} ms after staging)")
tb = nat.Batch(n, ios, [1] * n, adb, adl, pks_xy=pks, proofs=proofs)
t, rc = timed(lambda: ctx.thin_verify_into(tb, st32))
print(f"thin verify           {n / t:12.0f} /s  ({t * 1e3:.2f} ms; {dev():.2f} ms after staging), all ok: {rc == 0 and not any(st32)}")
