"""GPU parity: avrf_g1_msm (XYZZ Pippenger on BLS12-381 / BN254 G1) == the ring oracle's big-int MSM.
These are the KZG commit / open MSMs inside w3f-ring-proof (src/ring.rs:220,404,416,731); bases are
the reference's own SRS points (tests/golden/*.bin)."""
import os
import random

import pytest

from oracle import ring_py as R

pytestmark = pytest.mark.gpu
SRS = {0: "bls12-381-srs-2-11-uncompressed-zcash.bin", 1: "bn254-testing-2-9-uncompressed.bin"}


@pytest.fixture(scope="module")
def env(golden_dir):
    from ark_vrf_amd import _native as nat
    out = {}
    for i in (0, 1):
        s = R.SUITES[i]
        srs = R.Srs(s, open(os.path.join(golden_dir, SRS[i]), "rb").read())
        out[i] = (s, srs, nat.Context(i))
    return out


def le(s, P):
    n = s.fp_bytes
    return bytes(2 * n) if P is None else P[0].to_bytes(n, "little") + P[1].to_bytes(n, "little")


def check(s, ctx, pts, ks):
    got = ctx.g1_msm(b"".join(le(s, P) for P in pts), b"".join(k.to_bytes(32, "little") for k in ks))
    want = R.g1_affine(s.p, R.g1_msm(s.p, pts, ks)) if pts else None
    assert got == le(s, want)


@pytest.mark.parametrize("suite", [0, 1])
@pytest.mark.parametrize("n", [0, 1, 2, 33, 513, 1500])
def test_random(env, suite, n):
    s, srs, ctx = env[suite]
    rng = random.Random(100 * suite + n)
    check(s, ctx, srs.g1[:n], [rng.randrange(s.r) for _ in range(n)])


@pytest.mark.parametrize("suite", [0, 1])
def test_exceptional_cases(env, suite):
    s, srs, ctx = env[suite]
    rng = random.Random(9 + suite)
    P, Q = srs.g1[3], srs.g1[7]
    negP = (P[0], (-P[1]) % s.p)
    k = rng.randrange(s.r)
    check(s, ctx, [P, negP], [k, k])                                   # cancels to infinity
    check(s, ctx, [P] * 40, [k] * 40)                                  # same point, same bucket: doubling path
    check(s, ctx, [P, None, Q, None], [k, 5, 7, 0])                    # infinity among the bases
    check(s, ctx, [P, Q] * 100, [0] * 200)                             # all-zero scalars
    check(s, ctx, [P, Q] * 100, [1] * 200)
    check(s, ctx, [P, Q] * 100, [s.r - 1] * 200)
    ks = [rng.getrandbits(128) for _ in range(300)]
    check(s, ctx, srs.g1[:300], ks)
    ks = [(1 << 200) * (i % 3 + 1) % s.r for i in range(300)]          # one hot digit
    check(s, ctx, srs.g1[:300], ks)


def test_rejects_out_of_range(env):
    from ark_vrf_amd import _native as nat
    s, srs, ctx = env[0]
    with pytest.raises(nat.AvrfError):
        ctx.g1_msm(le(s, srs.g1[0]), (s.r).to_bytes(32, "little"))
    with pytest.raises(nat.AvrfError):
        ctx.g1_msm(b"\xff" * 96, (1).to_bytes(32, "little"))
