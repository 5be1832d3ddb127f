#!/usr/bin/env python3
"""Times the Ring-VRF path (BASELINE configs[3] shape: Bandersnatch / BLS12-381, ring 1024 -> N = 2048)."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ark_vrf_amd import _native as nat  # noqa: E402
from ark_vrf_amd.ring import RingSetup  # noqa: E402

ring = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nproofs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
suite = int(sys.argv[4]) if len(sys.argv) > 4 else 0                # 0 Bandersnatch/BLS12-381, 1 Baby-JubJub/BN254
if os.environ.get("AVRF_BLOCKING") == "1":                          # waiting host threads sleep in the driver (what bench.py's ranks do)
    assert nat.set_blocking_sync(0, True) == 0
ctx = nat.Context(suite)
srs = open(os.path.join(ROOT, "tests", "golden", ["bls12-381-srs-2-11-uncompressed-zcash.bin", "bn254-testing-2-9-uncompressed.bin"][suite]), "rb").read()
fq = 48 if suite == 0 else 32
have = int.from_bytes(srs[:8], "little")
if have < 3 * (1 << (ring + 4 + (253 if suite == 0 else 251) - 1).bit_length()) + 1:
    # no file that large: generate powers of a fixed tau from the file's generators
    from ark_vrf_amd.ring import srs_generate
    t = time.perf_counter()
    srs = srs_generate(ctx, suite, 0x1234567890abcdef1234567890abcdef, srs[8: 8 + 2 * fq], srs[8 + have * 2 * fq + 8: 8 + have * 2 * fq + 8 + 4 * fq], ring)
    print(f"generated SRS: {int.from_bytes(srs[:8], 'little')} G1 powers in {(time.perf_counter() - t)*1e3:.1f} ms")
R_SUB = bench.R_BANDERSNATCH if suite == 0 else 0x60c89ce5c263405370a08b6d0302b0bab3eedb83920ee0a677297dc392126f1
bench.R_BANDERSNATCH = R_SUB
t = time.perf_counter(); setup = RingSetup(ctx, srs, ring); t_setup = time.perf_counter() - t
sks = bench.derive_scalars(b"ring-bench-sk", 0, ring, bench.R_BANDERSNATCH)
pks = ctx.scalar_mul_base(sks)
pkl = [pks[64 * i: 64 * i + 64] for i in range(ring)]
t = time.perf_counter(); key = setup.index(pkl); t_index = time.perf_counter() - t
t = time.perf_counter(); key2 = setup.index(pkl); t_index2 = time.perf_counter() - t
bl = [bench.derive_scalars(b"ring-bench-bl", i, 1, bench.R_BANDERSNATCH) for i in range(nproofs)]
key.prove([3], bl[:1])
t = time.perf_counter(); proofs = key.prove([3] * nproofs, bl); t_prove = (time.perf_counter() - t) / nproofs
print(f"ring {ring} (N={setup.domain_size}): setup {t_setup*1e3:.1f} ms, index {t_index*1e3:.1f} / {t_index2*1e3:.1f} ms, prove {t_prove*1e3:.1f} ms/proof "
      f"({1/t_prove:.1f} proofs/s, one context, one host thread)")

nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if len(sys.argv) > 3:
    # independent contexts (stream + SRS tables + scratch each), one host thread per context, proofs split evenly
    from concurrent.futures import ThreadPoolExecutor
    ctxs = [ctx] + [nat.Context(suite) for _ in range(nctx - 1)]
    setups = [setup] + [RingSetup(c, srs, ring) for c in ctxs[1:]]
    keys = [key] + [su.index(pkl) for su in setups[1:]]
    for k in keys:
        k.prove([3], bl[:1])
    per = nproofs // nctx
    def work(i):
        return keys[i].prove([3] * per, bl[i * per:(i + 1) * per])
    pool = ThreadPoolExecutor(nctx)
    list(pool.map(work, range(nctx)))                        # warm (scratch allocation)
    import gc
    dts = []; cpu = []
    for _ in range(int(os.environ.get("RING_BENCH_PASSES", "5"))):
        gc.collect(); gc.disable()
        c0 = time.process_time()
        t = time.perf_counter(); res = list(pool.map(work, range(nctx))); dts.append(time.perf_counter() - t)
        cpu.append(time.process_time() - c0)
        gc.enable()
    assert res[0][0] == proofs[0]
    print(f"  {nctx} contexts x {per} proofs: best {per * nctx / min(dts):.1f} proofs/s ({min(dts) * 1e3:.1f} ms), mean {per * nctx * len(dts) / sum(dts):.1f}; passes ms {[round(x * 1e3, 1) for x in dts]}; host CPU ms per pass {[round(x * 1e3, 1) for x in cpu]}")
