"""Throughput of the device pairing check (avrf_ring_pairing_check): tools/pairing_bench.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ark_vrf_amd import _native as nat
from ark_vrf_amd.ring import RingSetup, pairing_check
from oracle import ring_py as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for suite, f in ((0, "bls12-381-srs-2-11-uncompressed-zcash.bin"), (1, "bn254-testing-2-9-uncompressed.bin")):
    raw = open(os.path.join(ROOT, "tests", "golden", f), "rb").read()
    s = R.SUITES[suite]; srs = R.Srs(s, raw)
    ctx = nat.Context(suite); setup = RingSetup(ctx, raw, 8)
    le = lambda P: P[0].to_bytes(s.fp_bytes, "little") + P[1].to_bytes(s.fp_bytes, "little")
    A = [le(srs.g1[1 + i % 500]) for i in range(n)]
    B = [le((srs.g1[i % 500][0], (-srs.g1[i % 500][1]) % s.p)) for i in range(n)]
    assert pairing_check(setup, A[:8], B[:8]) == [1] * 8
    for m in (1, 64, 1024, n):
        t = time.perf_counter(); ok = pairing_check(setup, A[:m], B[:m]); dt = time.perf_counter() - t
        assert all(ok)
        print(f"suite {suite}: {m} checks in {dt*1e3:.2f} ms -> {m/dt:.0f} checks/s ({2*m/dt:.0f} pairings/s)", flush=True)
