"""GPU parity at the sizes BASELINE.json's configs are benchmarked at: the MSM plan of a 65 536-item
batch (c = 13, 20 windows, SEG = 32, 33 tiles) is a different branch of the sort / lane-placement / window-sum kernels
than the small batches of the other test files, so the same bit-exact comparisons are repeated here at full size:

  configs[1]  thin::BatchVerifier, 65 536 items   (src/thin.rs:257-325)     262 145 terms
  configs[2]  pedersen::BatchVerifier, 65 536     (src/pedersen.rs:341-426) 327 682 terms
  msm_unchecked at n = 2^18 on random and adversarial scalar distributions (src/thin.rs:319)
  configs[4]  Baby-JubJub / BN254 ring 4096 -> N = 8192 over a generated 24 577-power SRS: URS bytes, ring commitment and
              one proof equal the pure-Python oracle's (fixture tests/golden/ring_large_oracle.json, made by
              tests/golden/gen_ring_large_oracle.py)
"""
import hashlib
import json
import os
import random

import pytest

import oracle as orc
from helpers import IDENTITY_XY, R_ORDER, nat_batch, rand_points_xy, rand_scalar

pytestmark = pytest.mark.gpu
N_FULL = 65536
THREADS = min(32, os.cpu_count() or 8)


@pytest.fixture(scope="module")
def ctx():
    from ark_vrf_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


def test_thin_batch_65536_terms_msm_verdict(ctx):
    b = orc.gen_batch(0, 0, N_FULL, threads=THREADS)
    assert ctx.thin_batch_stage(nat_batch(b)) == 0
    assert ctx.thin_batch_run() == 0
    st, bases, sc = orc.thin_batch_terms_xy(0, b)
    assert st == 0 and len(sc) == 32 * (4 * N_FULL + 1)
    gb, gs = ctx.last_terms()
    assert gs == sc, "weights / scalars differ from src/thin.rs:287-317"
    assert gb == bases
    # the MSM point itself (not only is_zero): valid batch -> identity; and on a perturbed scalar vector -> the oracle's point
    assert ctx.msm(bases, sc) == orc.msm(0, bases, sc) == IDENTITY_XY
    sc2 = bytearray(sc); sc2[32 * 12345] ^= 1; sc2[32 * 200000 + 7] ^= 0x10; sc2 = bytes(sc2)
    want = orc.msm(0, bases, sc2)
    assert want != IDENTITY_XY and ctx.msm(bases, sc2) == want
    # tampered response scalar -> VerificationFailure; identity public key -> InvalidData (checked before any equation)
    pr = bytearray(b["proofs"]); pr[96 * 40000 + 64] ^= 1
    b2 = dict(b); b2["proofs"] = bytes(pr)
    assert ctx.thin_batch_stage(nat_batch(b2)) == 0 and ctx.thin_batch_run() == 1 == orc.thin_batch_verify_xy(0, b2)
    pk = bytearray(b["pks_xy"]); pk[64 * 65535: 64 * 65536] = IDENTITY_XY
    b3 = dict(b); b3["pks_xy"] = bytes(pk)
    assert ctx.thin_batch_stage(nat_batch(b3)) == 0 and ctx.thin_batch_run() == 2 == orc.thin_batch_verify_xy(0, b3)


def test_pedersen_batch_65536_terms_msm_verdict(ctx):
    b = orc.gen_batch(0, 1, N_FULL, threads=THREADS)
    b["pks_xy"] = b""
    assert ctx.pedersen_batch_stage(nat_batch(b)) == 0
    assert ctx.pedersen_batch_run() == 0
    st, bases, sc = orc.pedersen_batch_terms_xy(0, b)
    assert st == 0 and len(sc) == 32 * (5 * N_FULL + 2)
    gb, gs = ctx.last_terms()
    assert gs == sc, "weights / scalars differ from src/pedersen.rs:373-418"
    assert gb == bases
    assert ctx.msm(bases, sc) == orc.msm(0, bases, sc) == IDENTITY_XY
    sc2 = bytearray(sc); sc2[32 * 300000 + 3] ^= 4; sc2 = bytes(sc2)
    assert ctx.msm(bases, sc2) == orc.msm(0, bases, sc2)
    pr = bytearray(b["proofs"]); pr[256 * 1234 + 224] ^= 1                   # sb of item 1234
    b2 = dict(b); b2["proofs"] = bytes(pr)
    assert ctx.pedersen_batch_stage(nat_batch(b2)) == 0 and ctx.pedersen_batch_run() == 1


@pytest.mark.parametrize("kind", [0, 1])
def test_independent_prove_verify_65536_vs_oracle(ctx, kind):
    """configs[2] / the per-item kernels at the benchmarked size: 65 536 INDEPENDENT Thin (kind 0) / Pedersen (kind 1) proofs
    equal the oracle's byte for byte (src/thin.rs:111-129, src/pedersen.rs:136-186), every one verifies independently
    (src/thin.rs:131-165, src/pedersen.rs:188-249), and tampered items -- first, last and a few inside, across workgroup and
    workspace-slot boundaries -- are the only ones reported."""
    b = orc.gen_batch(0, kind, N_FULL, threads=THREADS)
    psz = 96 if kind == 0 else 256
    if kind == 0:
        assert ctx.thin_prove(nat_batch(b, with_sks=True, with_proofs=False)) == b["proofs"]
        assert ctx.thin_verify(nat_batch(b)) == [0] * N_FULL
    else:
        pr, _ = ctx.pedersen_prove(nat_batch(b, with_sks=True, with_proofs=False))
        assert pr == b["proofs"]
        b["pks_xy"] = b""
        assert ctx.pedersen_verify(nat_batch(b)) == [0] * N_FULL
    bad_items = [0, 1, 63, 64, 127, 128, 4095, 40000, N_FULL - 1]
    p2 = bytearray(b["proofs"])
    for j in bad_items:
        p2[psz * j + (64 if kind == 0 else 200)] ^= 1                        # a bit of s
    b2 = dict(b); b2["proofs"] = bytes(p2)
    st = ctx.thin_verify(nat_batch(b2)) if kind == 0 else ctx.pedersen_verify(nat_batch(b2))
    assert [j for j, v in enumerate(st) if v != 0] == bad_items and all(st[j] == 1 for j in bad_items)


@pytest.mark.parametrize("suite", [0, 1])
def test_msm_2pow18_adversarial(suite):
    """n = 2^18 + 3 -> c = 13, 33 tiles: random scalars and the skewed distributions of test_gpu_msm.py at full size."""
    from ark_vrf_amd import _native as nat
    c = nat.Context(suite)
    rng = random.Random(4242 + suite)
    r = R_ORDER[suite]
    n = (1 << 18) + 3
    pts = rand_points_xy(rng, suite, 257)
    bases = b"".join(pts[(i * 7 + i // 257) % 257] for i in range(n))
    rs = [rand_scalar(rng, suite) for _ in range(1024)]
    cases = {
        "random": [rs[(i * 31 + i // 1024) % 1024] for i in range(n)],
        "mixed_128bit": [rs[i % 1024] if i % 4 else rs[i % 1024][:16] + bytes(16) for i in range(n)],
        "single_hot_digit": [((1 << 200) * (i % 3 + 1) % r).to_bytes(32, "little") for i in range(n)],
        "same_scalar": [rs[0]] * n,
        "r_minus_1_and_zero": [(r - 1).to_bytes(32, "little") if i % 2 else bytes(32) for i in range(n)],
    }
    for name, sc in cases.items():
        sc = b"".join(sc)
        assert c.msm(bases, sc) == orc.msm(suite, bases, sc), name
    # identity points among the bases, one bucket taking every entry of a window
    bases2 = b"".join(IDENTITY_XY if i % 5 == 0 else pts[0] for i in range(n))
    sc = b"".join(rs[i % 7] for i in range(n))
    assert c.msm(bases2, sc) == orc.msm(suite, bases2, sc)
    c.close()


def test_bn254_ring_4096_matches_oracle_fixture(golden_dir):
    """BASELINE configs[4] shape (Baby-JubJub / BN254, ring 4096 -> N = 8192, NTT 2^15, generated 24 577-power SRS)."""
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import RingSetup, ring_batch_verify, srs_generate
    fx = json.load(open(os.path.join(golden_dir, "ring_large_oracle.json")))["bn254_ring4096"]
    suite, ring = fx["suite"], fx["ring_size"]
    r_bn = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    srs_file = open(os.path.join(golden_dir, "bn254-testing-2-9-uncompressed.bin"), "rb").read()
    cnt = int.from_bytes(srs_file[:8], "little")
    g1 = srs_file[8: 8 + 64]
    g2 = srs_file[8 + cnt * 64 + 8: 8 + cnt * 64 + 8 + 128]
    ctx = nat.Context(suite)
    urs = srs_generate(ctx, suite, int(fx["tau"], 16), g1, g2, ring)
    assert int.from_bytes(urs[:8], "little") == fx["n_g1"]
    assert hashlib.sha256(urs).hexdigest() == fx["urs_sha256"]               # Kzg::setup == oracle, all 24 577 powers + tau g2
    setup = RingSetup(ctx, urs, ring)
    assert setup.domain_size == fx["domain_size"] == 8192
    ks = b"".join((int.from_bytes(hashlib.sha512(b"k%d" % i).digest(), "little") % (r_bn >> 3) + 1).to_bytes(32, "little") for i in range(ring))
    pks_xy = ctx.scalar_mul_base(ks)
    assert hashlib.sha256(pks_xy).hexdigest() == fx["pks_sha256"]
    pkl = [pks_xy[64 * i: 64 * i + 64] for i in range(ring)]
    key = setup.index(pkl)
    assert key.commitment.hex() == fx["commitment"]                          # ring_proof::index at N = 8192
    proof = key.prove([fx["key_index"]], [bytes.fromhex(fx["blinding"])])[0]
    assert proof.hex() == fx["proof"]                                        # RingProver::prove at N = 8192
    # and it verifies as the statement "Yb = pk_777 + b * B" (instance computed by the oracle's group law)
    b = int.from_bytes(bytes.fromhex(fx["blinding"]), "little")
    bb = orc.smul(suite, b.to_bytes(32, "little"), orc.suite_point(suite, 1))
    st, bbxy = orc.point_decompress(suite, bb)
    assert st == 0
    yb = ctx.msm(pkl[fx["key_index"]] + bbxy, (1).to_bytes(32, "little") * 2)
    assert ring_batch_verify(setup, [key.commitment], None, [yb], [proof]) == 0
    assert ring_batch_verify(setup, [key.commitment], None, [pkl[0]], [proof]) == 1
    key.close(); setup.close(); ctx.close()
