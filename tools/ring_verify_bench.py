#!/usr/bin/env python3
"""Times ring verification of n proofs over a ring of `ring` keys on one context: the batch verifier and n independent
verifications (device pairings): tools/ring_verify_bench.py [ring] [n]   (AVRF_RING_TRACE=1 prints the phases)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ark_vrf_amd import _native as nat
from ark_vrf_amd.ring import RingSetup, ring_batch_verify, ring_verify_each
ring = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = nat.Context(0)
srs = open(os.path.join(ROOT, "tests", "golden", "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()
setup = RingSetup(ctx, srs, ring)
sks = bench.derive_scalars(b"rv-sk", 0, ring, bench.R_BANDERSNATCH)
pks = ctx.scalar_mul_base(sks); pkl = [pks[64 * i: 64 * i + 64] for i in range(ring)]
key = setup.index(pkl)
idx = [(7 * j + 3) % ring for j in range(n)]
inputs = ctx.scalar_mul_base(bench.derive_scalars(b"rv-in", 0, n, bench.R_BANDERSNATCH))
psk = b"".join(sks[32 * k: 32 * k + 32] for k in idx); ppk = b"".join(pkl[k] for k in idx)
outs = ctx.scalar_mul(psk, inputs)
ios = b"".join(inputs[64 * j: 64 * j + 64] + outs[64 * j: 64 * j + 64] for j in range(n))
ads = [b"ad-%d" % j for j in range(n)]
ped, blind = ctx.pedersen_prove(nat.Batch(n, ios, [1] * n, b"".join(ads), [len(a) for a in ads], pks_xy=ppk, sks=psk))
proofs = key.prove(idx, [blind[32 * j: 32 * j + 32] for j in range(n)])
ybs = [ped[256 * j: 256 * j + 64] for j in range(n)]
for name, fn in (("batch verifier", lambda: ring_batch_verify(setup, [key.commitment], None, ybs, proofs)),
                 ("independent (per-proof verdicts)", lambda: max(ring_verify_each(setup, [key.commitment], None, ybs, proofs)))):
    assert fn() == 0
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); assert fn() == 0; best = min(best, time.perf_counter() - t)
    print(f"{name}: {n} proofs in {best*1e3:.2f} ms -> {n/best:.0f} verifications/s (ring half, one context)", flush=True)
