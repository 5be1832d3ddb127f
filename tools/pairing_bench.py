"""Throughput of the device pairing check (avrf_ring_pairing_check): tools/pairing_bench.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ark_vrf_amd import _native as nat
from ark_vrf_amd.ring import RingSetup, pairing_check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for suite, f in ((0, "bls12-381-srs-2-11-uncompressed-zcash.bin"), (1, "bn254-testing-2-9-uncompressed.bin")):
    raw = open(os.path.join(ROOT, "tests", "golden", f), "rb").read()
    fq = 48 if suite == 0 else 32
    p = [0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
         21888242871839275222246405745257275088696311157297823662689037894645226208583][suite]

    def g1(i):                                       # i-th `powers_in_g1` entry of the URS file as (x, y)
        e = raw[8 + 2 * fq * i: 8 + 2 * fq * (i + 1)]
        if suite == 0:
            return int.from_bytes(e[:fq], "big"), int.from_bytes(e[fq:], "big")
        y = bytearray(e[fq:]); y[-1] &= 0x3f
        return int.from_bytes(e[:fq], "little"), int.from_bytes(y, "little")
    ctx = nat.Context(suite); setup = RingSetup(ctx, raw, 8)
    le = lambda P: P[0].to_bytes(fq, "little") + P[1].to_bytes(fq, "little")
    pts = [g1(i) for i in range(501)]
    A = [le(pts[1 + i % 500]) for i in range(n)]                      # tau^(k+1) g1
    B = [le((pts[i % 500][0], (-pts[i % 500][1]) % p)) for i in range(n)]   # -tau^k g1
    assert pairing_check(setup, A[:8], B[:8]) == [1] * 8
    for m in (1, 64, 1024, n):
        t = time.perf_counter(); ok = pairing_check(setup, A[:m], B[:m]); dt = time.perf_counter() - t
        assert all(ok)
        print(f"suite {suite}: {m} checks in {dt*1e3:.2f} ms -> {m/dt:.0f} checks/s ({2*m/dt:.0f} pairings/s)", flush=True)
