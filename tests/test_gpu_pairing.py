"""GPU parity of the device pairing (pairing.hip: Miller loop + final exponentiation, BLS12-381 and BN254) through
avrf_ring_pairing_check, against the relations the reference's own SRS files satisfy and against the pure-Python pairing of
oracle/pairing_py.py (itself pinned by tests/test_oracle_ring.py): arkworks `Pairing::multi_pairing` as reached from
RingVerifier::verify (src/ring.rs:242)."""
import os
import random

import pytest

from oracle import pairing_py as PP
from oracle import ring_py as R

pytestmark = pytest.mark.gpu
SRS = {0: "bls12-381-srs-2-11-uncompressed-zcash.bin", 1: "bn254-testing-2-9-uncompressed.bin"}


def le_xy(s, P):
    if P is None:
        return bytes(2 * s.fp_bytes)
    return P[0].to_bytes(s.fp_bytes, "little") + P[1].to_bytes(s.fp_bytes, "little")


@pytest.fixture(scope="module")
def env(golden_dir):
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import RingSetup
    out = {}
    for suite, f in SRS.items():
        raw = open(os.path.join(golden_dir, f), "rb").read()
        ctx = nat.Context(suite)
        out[suite] = (ctx, RingSetup(ctx, raw, 8), R.Srs(R.SUITES[suite], raw))
    return out


@pytest.mark.parametrize("suite", [0, 1])
def test_srs_relations(env, suite):
    """e(tau^i g1, g2) == e(tau^(i-1) g1, tau g2) for the powers of the reference's SRS files; any mismatch is detected."""
    from ark_vrf_amd.ring import pairing_check
    ctx, setup, srs = env[suite]
    s = R.SUITES[suite]
    p = s.p
    neg = lambda P: None if P is None else (P[0], (-P[1]) % p)
    add = lambda P, Q: R.g1_affine(p, R.g1_add(p, P + (1,), Q + (1,)))
    g = srs.g1
    A = [g[1], g[2], g[700], add(g[3], g[5]), g[4], g[9], None, None, g[1]]
    B = [neg(g[0]), neg(g[1]), neg(g[699]), neg(add(g[2], g[4])), neg(g[4]), neg(g[7]), None, neg(g[1]), None]
    want = [1, 1, 1, 1, 0, 0, 1, 0, 0]
    got = pairing_check(setup, [le_xy(s, P) for P in A], [le_xy(s, P) for P in B])
    assert got == want
    # 300 checks in one launch (more than one wave per SIMD row): every third one is broken
    n = 300
    A2 = [g[1 + (i % 50)] for i in range(n)]
    B2 = [neg(g[(i % 50) + (1 if i % 3 == 0 else 0)]) for i in range(n)]
    assert pairing_check(setup, [le_xy(s, P) for P in A2], [le_xy(s, P) for P in B2]) == [0 if i % 3 == 0 else 1 for i in range(n)]


@pytest.mark.parametrize("suite", [0, 1])
def test_matches_oracle_pairing(env, suite):
    """Random G1 arguments (multiples of g1 and of tau g1): the verdict of both forms of avrf_ring_pairing_check (<= 16 checks: host pool
    with the tabulated G2 lines; more: device Miller loops) equals oracle/pairing_py.pairing_product_is_one."""
    from ark_vrf_amd.ring import pairing_check
    ctx, setup, srs = env[suite]
    s = R.SUITES[suite]
    p = s.p
    PP.use_curve("bls12_381" if suite == 0 else "bn254")
    try:
        dec = PP.g2_decode_zcash_uncompressed if suite == 0 else PP.g2_decode_arkworks_uncompressed
        q0, q1 = dec(srs.g2_raw[0]), dec(srs.g2_raw[1])
        rng = random.Random(99 + suite)
        mul = lambda P, k: R.g1_affine(p, R.g1_mul(p, P + (1,), k % s.r))
        A, B, want = [], [], []
        for i in range(4):
            k = rng.randrange(1, s.r)
            a = mul(srs.g1[1], k)                                            # k tau g1
            b = mul(srs.g1[0], (-k if i % 2 == 0 else -k + 1))               # -k g1  (or -(k-1) g1: broken)
            A.append(a); B.append(b)
            want.append(1 if PP.pairing_product_is_one([(a, q0), (b, q1)]) else 0)
        assert want == [1, 0, 1, 0]
        assert pairing_check(setup, [le_xy(s, P) for P in A], [le_xy(s, P) for P in B]) == want          # few checks: finished on the host pool
        assert pairing_check(setup, [le_xy(s, P) for P in A] * 8, [le_xy(s, P) for P in B] * 8) == want * 8  # 32 checks: the device pairing kernel
    finally:
        PP.use_curve("bls12_381")


def test_rejects_noncanonical_coordinate(env):
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import pairing_check
    ctx, setup, srs = env[0]
    s = R.SUITES[0]
    bad = (s.p).to_bytes(48, "little") + (1).to_bytes(48, "little")
    with pytest.raises(nat.AvrfError, match="-> 2"):
        pairing_check(setup, [bad], [le_xy(s, srs.g1[0])])
