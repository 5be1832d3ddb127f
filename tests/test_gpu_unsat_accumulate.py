"""The bucket accumulation on unsaturated limbs (csrc/fpu.h, fpu_te.h, fpu_g1.h; round 5) through the C ABI, on inputs built to
reach every branch the random full-size tests reach only by accident:

  twisted Edwards (avrf_msm_te = msm_unchecked of src/thin.rs:319, src/pedersen.rs:420): the SIGN of a lane's accumulator flips
    whenever consecutive entries of its share have opposite digit signs; a share may start, continue and end on the identity
    point, on copies of one base, on a base followed by its negation; partial sums leave as raw limbs and are made canonical
    by their readers (k_bucket_sum, the heavy-bucket quads);
  G1 (avrf_g1_msm, the KZG MSMs of src/ring.rs:220,404,416,731) ABOVE the single-launch size: the XYZZ law is not complete --
    P = Q (doubling, saturated form + conversion), P = -Q (identity flag), a base at infinity, an accumulator that became the
    identity and goes on -- detected by the exact limb test on P^2.

Expected values: oracle/orc_msm.c (naive double-and-add) and oracle/ring_py.py (big-int Pippenger)."""
import os
import random

import pytest

import oracle as orc
from oracle import ring_py as R
from helpers import IDENTITY_XY, R_ORDER, rand_points_xy

pytestmark = pytest.mark.gpu
TE_SUITES = [orc.BANDERSNATCH, orc.BABYJUBJUB, orc.JUBJUB, orc.ED25519, orc.BANDERSNATCH_SW, orc.BANDERSNATCH_SHAKE128, orc.TESTING_SHA256]
SRS = {0: "bls12-381-srs-2-11-uncompressed-zcash.bin", 1: "bn254-testing-2-9-uncompressed.bin"}


def neg_xy(suite, xy):
    """-(x, y) = (-x, y) on a twisted Edwards curve; the field modulus from the oracle's identity-preserving round trip"""
    p = {orc.BANDERSNATCH: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
         orc.BABYJUBJUB: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
         orc.JUBJUB: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
         orc.ED25519: 2 ** 255 - 19, orc.TESTING_SHA256: 2 ** 255 - 19}
    p[orc.BANDERSNATCH_SW] = p[orc.BANDERSNATCH_SHAKE128] = p[orc.BANDERSNATCH]
    x = int.from_bytes(xy[:32], "little")
    return ((-x) % p[suite]).to_bytes(32, "little") + xy[32:]


@pytest.mark.parametrize("suite", TE_SUITES)
def test_te_pippenger_signs_duplicates_identity(suite):
    from ark_vrf_amd import _native as nat
    rng = random.Random(500 + suite)
    r = R_ORDER[suite]
    pts = rand_points_xy(rng, suite, 24)
    bases, ks = [], []

    def add(p, k):
        bases.append(p); ks.append(k % r)
    for i in range(1500):                                              # random terms, both digit signs in every window
        add(pts[i % 24], rng.randrange(r))
    k0 = rng.randrange(r)
    for i in range(300):                                               # one base, one scalar: every bucket run is P + P + P ...
        add(pts[0], k0)
    for i in range(150):                                               # P then -P with the same scalar: runs that pass through the identity
        add(pts[1 + i % 5], k0 + i); add(neg_xy(suite, pts[1 + i % 5]), k0 + i)
    for i in range(100):                                               # the identity point as a base, and zero scalars
        add(IDENTITY_XY, rng.randrange(r)); add(pts[7], 0)
    for i in range(200):                                               # scalars at the ends of the range, one hot digit, r - 1
        add(pts[8 + i % 9], [1, r - 1, (1 << 200) % r, r - 2, (1 << 252) % r][i % 5])
    order = list(range(len(ks))); rng.shuffle(order)
    bx = b"".join(bases[i] for i in order); sc = b"".join(ks[i].to_bytes(32, "little") for i in order)
    assert len(order) > 2048                                           # above the single-launch MSM: the Pippenger chain runs
    ctx = nat.Context(suite)
    try:
        assert ctx.msm(bx, sc) == orc.msm(suite, bx, sc)
    finally:
        ctx.close()


def test_te_pippenger_one_base_many_times_full_windows():
    """2^17 copies of G with one scalar, then with alternating signs: every lane's whole share is one bucket / two buckets"""
    from ark_vrf_amd import _native as nat
    suite, n = orc.BANDERSNATCH, 1 << 17
    r = R_ORDER[suite]
    g = orc.point_decompress(suite, orc.suite_point(suite, 0))[1]
    k = 0x1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcd % r
    ctx = nat.Context(suite)
    try:
        want = orc.point_decompress(suite, orc.smul(suite, (k * n % r).to_bytes(32, "little"), orc.suite_point(suite, 0)))[1]
        assert ctx.msm(g * n, k.to_bytes(32, "little") * n) == want
        both = (g + neg_xy(suite, g)) * (n // 2)
        assert ctx.msm(both, k.to_bytes(32, "little") * n) == IDENTITY_XY
    finally:
        ctx.close()


def le(s, P):
    n = s.fp_bytes
    return bytes(2 * n) if P is None else P[0].to_bytes(n, "little") + P[1].to_bytes(n, "little")


@pytest.mark.parametrize("suite", [0, 1])
def test_g1_pippenger_exceptional_cases(golden_dir, suite):
    from ark_vrf_amd import _native as nat
    s = R.SUITES[suite]
    srs = R.Srs(s, open(os.path.join(golden_dir, SRS[suite]), "rb").read())
    rng = random.Random(77 + suite)
    g1 = srs.g1
    pts, ks = [], []
    for i in range(2600):                                              # distinct points, random scalars
        pts.append(g1[i % len(g1)]); ks.append(rng.randrange(s.r))
    k0 = rng.randrange(s.r)
    for i in range(300):                                               # P = Q in every window: the doubling branch, then ordinary additions onto 2P
        pts.append(g1[5]); ks.append(k0)
    for i in range(120):                                               # P then -P: the accumulator becomes the identity and goes on
        P = g1[9 + i % 4]
        pts.append(P); ks.append(k0 + 3 * i); pts.append((P[0], (-P[1]) % s.p)); ks.append(k0 + 3 * i)
    for i in range(80):                                                # bases at infinity, zero scalars
        pts.append(None); ks.append(rng.randrange(s.r)); pts.append(g1[11]); ks.append(0)
    for i in range(120):                                               # extreme scalars
        pts.append(g1[20 + i % 7]); ks.append([1, s.r - 1, (1 << 200) % s.r, (1 << 254) % s.r][i % 4])
    order = list(range(len(ks))); rng.shuffle(order)
    pts = [pts[i] for i in order]; ks = [ks[i] % s.r for i in order]
    assert len(ks) > 2048
    ctx = nat.Context(suite)
    try:
        got = ctx.g1_msm(b"".join(le(s, P) for P in pts), b"".join(k.to_bytes(32, "little") for k in ks))
        assert got == le(s, R.g1_affine(s.p, R.g1_msm(s.p, pts, ks)))
        # everything cancels: the sum of k P and k (-P) over many P is the point at infinity
        pts2 = [g1[i // 2] if i % 2 == 0 else (g1[i // 2][0], (-g1[i // 2][1]) % s.p) for i in range(3000)]
        ks2 = [1 + (i // 2) * 0x9e3779b97f4a7c15 % s.r for i in range(3000)]
        assert ctx.g1_msm(b"".join(le(s, P) for P in pts2), b"".join(k.to_bytes(32, "little") for k in ks2)) == le(s, None)
    finally:
        ctx.close()
