"""avrf_secret_from_seed / avrf_output_hash: Secret::from_seed (src/lib.rs:346-369) and Output::hash = Suite::point_to_hash
(src/lib.rs:605-609, src/utils/common.rs:290-305) on the device, for every suite: the reference's own vectors (sk and pk from the
seeds [1], [2], .. of src/testing.rs:262-266; beta from gamma, :278) and the oracle on a few hundred random seeds / points."""
import json
import os
import random

import pytest

import oracle as orc
from helpers import R_ORDER

pytestmark = pytest.mark.gpu
SUITES = {
    "bandersnatch_sha-512_ell2": orc.BANDERSNATCH, "baby-jubjub_sha-512_tai": orc.BABYJUBJUB, "jubjub_sha-512_tai": orc.JUBJUB,
    "bandersnatch_sw_sha-512_tai": orc.BANDERSNATCH_SW, "ed25519_sha-512_tai": orc.ED25519, "testing_sha-256_tai": orc.TESTING_SHA256,
    "bandersnatch_shake128_ell2": orc.BANDERSNATCH_SHAKE128, "secp256r1_sha-256_tai": orc.SECP256R1,
}
SEEDS = [1, 2, 3, 4, 5, 5, 6]


def to_xy(suite, comp):
    if suite == orc.BANDERSNATCH_SW:
        comp = orc.sw_decode(suite, comp) if len(comp) == 33 else comp
    st, xy = orc.point_decompress(suite, comp)
    assert st == 0
    return xy


@pytest.mark.parametrize("name", list(SUITES))
def test_reference_vectors(golden_dir, name):
    from ark_vrf_amd import _native as nat
    s = SUITES[name]
    vs = json.load(open(os.path.join(golden_dir, f"{name}_thin.json")))
    ctx = nat.Context(s)
    try:
        seeds = b"".join(bytes([SEEDS[i]]) + bytes(31) for i in range(len(vs)))
        sks, pks = ctx.secret_from_seed(seeds)
        assert sks == b"".join(bytes.fromhex(v["sk"]) for v in vs)
        assert ctx.points_compress(pks) == b"".join(bytes.fromhex(v["pk"]) for v in vs)          # the suite's own wire form
        gam, st = ctx.points_decompress(b"".join(bytes.fromhex(v["gamma"]) for v in vs))
        assert st == [0] * len(vs)
        assert [h.hex() for h in ctx.output_hash(gam)] == [v["beta"] for v in vs]
        assert ctx.secret_from_seed(seeds, with_public=False) == (sks, None)
    finally:
        ctx.close()


@pytest.mark.parametrize("name", list(SUITES))
def test_against_oracle(name):
    from ark_vrf_amd import _native as nat
    s = SUITES[name]
    rng = random.Random(31 + s)
    n = 300
    seeds = [rng.randbytes(32) for _ in range(n - 3)] + [bytes(32), b"\xff" * 32, R_ORDER[s].to_bytes(32, "little")]
    ctx = nat.Context(s)
    try:
        sks, pks = ctx.secret_from_seed(b"".join(seeds))
        want = [orc.from_seed(s, sd) for sd in seeds]
        assert sks == b"".join(w[0] for w in want)
        assert pks == b"".join(to_xy(s, w[1]) for w in want)
        comp = [w[1] for w in want]                                        # the public keys as sample points
        for nb in (32, 64, 16, 1):
            got = ctx.output_hash(pks, nb)
            assert got == [orc.point_to_hash(s, c, nb) for c in comp]
        assert ctx.output_hash(b"") == []
    finally:
        ctx.close()


def test_bad_arguments():
    import ctypes as C
    from ark_vrf_amd import _native as nat
    ctx = nat.Context(0)
    try:
        L = nat.lib()
        out = (C.c_uint8 * 128)()
        pt = (C.c_uint8 * 64)()
        assert L.avrf_output_hash(ctx._h, C.c_size_t(1), pt, C.c_size_t(65), out) == -2
        assert L.avrf_output_hash(ctx._h, C.c_size_t(1), pt, C.c_size_t(0), out) == -2
        assert L.avrf_output_hash(ctx._h, C.c_size_t(1), None, C.c_size_t(32), out) == -2
        assert L.avrf_secret_from_seed(ctx._h, C.c_size_t(1), None, out, None) == -2
    finally:
        ctx.close()
