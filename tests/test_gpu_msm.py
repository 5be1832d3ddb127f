"""GPU parity: avrf_msm_te (HIP Pippenger) == oracle MSM, bit-exact on the normalised result.
Reference call sites: src/thin.rs:319, src/pedersen.rs:420, src/utils/common.rs:410-411."""
import random

import pytest

import oracle as orc
from helpers import IDENTITY_XY, R_ORDER, rand_points_xy, rand_scalar

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctxs():
    from ark_vrf_amd import _native as nat
    return {s: nat.Context(s) for s in (0, 1)}


@pytest.mark.parametrize("suite", [0, 1])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 29, 64, 65, 257, 1000, 4097])
def test_msm_random(ctxs, suite, n):
    rng = random.Random(1000 * suite + n)
    pts = rand_points_xy(rng, suite, min(n, 300))
    bases = b"".join(pts[i % len(pts)] for i in range(n)) if n else b""
    sc = b"".join(rand_scalar(rng, suite) for _ in range(n))
    assert ctxs[suite].msm(bases, sc) == (orc.msm(suite, bases, sc) if n else IDENTITY_XY)


@pytest.mark.parametrize("suite", [0, 1])
def test_msm_adversarial(ctxs, suite):
    rng = random.Random(7 + suite)
    r = R_ORDER[suite]
    pts = rand_points_xy(rng, suite, 40)
    n = 600
    bases = [pts[i % 40] for i in range(n)]
    cases = {
        "zeros": [bytes(32)] * n,
        "ones": [(1).to_bytes(32, "little")] * n,
        "r_minus_1": [(r - 1).to_bytes(32, "little")] * n,
        "same_scalar": [rand_scalar(rng, suite)] * n,
        "128bit": [rand_scalar(rng, suite, 128) for _ in range(n)],
        "single_hot_digit": [((1 << 200) * (i % 3 + 1) % r).to_bytes(32, "little") for i in range(n)],
        "mixed": [rand_scalar(rng, suite, 128 if i % 4 == 0 else None) for i in range(n)],
    }
    for name, sc in cases.items():
        got = ctxs[suite].msm(b"".join(bases), b"".join(sc))
        assert got == orc.msm(suite, b"".join(bases), b"".join(sc)), name
    # identity points and repeated points among the bases
    bases2 = [IDENTITY_XY if i % 5 == 0 else pts[0] for i in range(n)]
    sc = [rand_scalar(rng, suite) for _ in range(n)]
    assert ctxs[suite].msm(b"".join(bases2), b"".join(sc)) == orc.msm(suite, b"".join(bases2), b"".join(sc))
    # P and -P cancel
    x = int.from_bytes(pts[1][:32], "little")
    q = {0: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
         1: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}[suite]
    neg = ((q - x) % q).to_bytes(32, "little") + pts[1][32:]
    k = rand_scalar(rng, suite)
    assert ctxs[suite].msm(pts[1] + neg, k + k) == IDENTITY_XY


def test_msm_rejects_noncanonical(ctxs):
    from ark_vrf_amd import _native as nat
    bad_scalar = (R_ORDER[0]).to_bytes(32, "little")
    with pytest.raises(nat.AvrfError):
        ctxs[0].msm(IDENTITY_XY, bad_scalar)
    bad_pt = b"\xff" * 32 + (1).to_bytes(32, "little")
    with pytest.raises(nat.AvrfError):
        ctxs[0].msm(bad_pt, (1).to_bytes(32, "little"))


@pytest.mark.parametrize("suite", [0, 1])
def test_msm_montgomery_limbs_flavour(suite):
    """avrf_msm_te_mont (SURVEY.md 8b zero-copy option): bases and scalars as arkworks holds them in memory -- Montgomery limbs,
    R = 2^256 -- and the result in the same form; must equal the canonical flavour / the oracle after conversion."""
    import ctypes as C
    import random
    from ark_vrf_amd import _native as nat
    from helpers import R_ORDER, rand_points_xy, rand_scalar
    q = {0: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
         1: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}[suite]
    r = R_ORDER[suite]
    R = 1 << 256
    to_m = lambda b, m: (int.from_bytes(b, "little") * R % m).to_bytes(32, "little")
    from_m = lambda b, m: (int.from_bytes(b, "little") * pow(R, -1, m) % m).to_bytes(32, "little")
    rng = random.Random(5 + suite)
    c = nat.Context(suite)
    pts = rand_points_xy(rng, suite, 50)
    for n in (1, 37, 3000):
        bases = [pts[i % 50] for i in range(n)]
        sc = [rand_scalar(rng, suite) for _ in range(n)]
        want = orc.msm(suite, b"".join(bases), b"".join(sc))
        bm = b"".join(to_m(p[:32], q) + to_m(p[32:], q) for p in bases)
        sm = b"".join(to_m(k, r) for k in sc)
        out = (C.c_uint8 * 64)()
        assert nat.lib().avrf_msm_te_mont(c._h, C.c_size_t(n), nat._u8(bm), nat._u8(sm), out) == 0
        got = bytes(out)
        assert from_m(got[:32], q) + from_m(got[32:], q) == want == c.msm(b"".join(bases), b"".join(sc))
    # a limb value >= the modulus is refused
    bad = (q).to_bytes(32, "little") + bytes(32)
    assert nat.lib().avrf_msm_te_mont(c._h, C.c_size_t(1), nat._u8(bad), nat._u8(to_m(rand_scalar(rng, suite), r)), out) == nat.INVALID_DATA
    c.close()
