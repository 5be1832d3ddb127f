"""GPU parity for the crate's own test suite, Testing-SHA256-TAI (src/suites/testing.rs): edwards25519 with
HashTranscript<Sha256> -- the third transcript type the kernels are generic over (sha256_dev.h: 32-byte digests, 32-byte
counter-mode blocks).  Tiny / Thin / Pedersen against the reference's `testing_sha-256_tai_{thin,tiny,pedersen}.json` vectors and
the oracle (suite id 6); no RingSuite."""
import hashlib
import json
import os
import random

import pytest

import oracle as orc
from helpers import IDENTITY_XY, compressed_items, nat_batch, proof_comp, proof_xy, rand_points_xy, rand_scalar, xy

pytestmark = pytest.mark.gpu
S = 6
NAME = "testing_sha-256_tai"


@pytest.fixture(scope="module")
def ctx():
    from ark_vrf_amd import _native as nat
    c = nat.Context(nat.TESTING_SHA256_TAI)
    yield c
    c.close()


def load(golden_dir, k):
    return json.load(open(os.path.join(golden_dir, f"{NAME}_{k}.json")))


def test_msm_and_hash_to_curve(ctx, golden_dir):
    rng = random.Random(7)
    pts = rand_points_xy(rng, S, 60)
    for n in (1, 29, 700, 4097):
        bases = b"".join(pts[i % 60] for i in range(n))
        sc = b"".join(rand_scalar(rng, S) for _ in range(n))
        assert ctx.msm(bases, sc) == orc.msm(S, bases, sc)
    vs = load(golden_dir, "thin")
    msgs = [bytes.fromhex(v["alpha"]) for v in vs] + [hashlib.sha512(b"ts%d" % i).digest()[: i % 65] for i in range(100)]
    xy_, st = ctx.hash_to_curve(msgs)
    got = ctx.points_compress(xy_)
    assert all(s == 0 for s in st)
    assert [got[32 * i: 32 * i + 32].hex() for i in range(7)] == [v["h"] for v in vs]                  # try-and-increment, alpha -> h
    assert all(got[32 * i: 32 * i + 32] == orc.hash_to_curve(S, msgs[i]) for i in range(7, len(msgs)))
    # point codec with Validate::Yes: a point of order 2 is on the curve but not in the prime-order subgroup (cofactor 8)
    q = 2 ** 255 - 19
    _, st = ctx.points_decompress((q - 1).to_bytes(32, "little") + bytes.fromhex(vs[0]["pk"]), validate=True)
    assert st == [2, 0]


def test_thin_tiny_pedersen_vectors(ctx, golden_dir):
    from ark_vrf_amd._native import Batch
    th, ti, pe = load(golden_dir, "thin"), load(golden_dir, "tiny"), load(golden_dir, "pedersen")
    sks = [bytes.fromhex(v["sk"]) for v in th]
    pks = [xy(S, bytes.fromhex(v["pk"])) for v in th]
    ios = [[(xy(S, bytes.fromhex(v["h"])), xy(S, bytes.fromhex(v["gamma"])))] for v in th]
    ads = [bytes.fromhex(v["ad"]) for v in th]
    assert ctx.scalar_mul_base(b"".join(sks)) == b"".join(pks)                                            # sk -> pk
    assert ctx.scalar_mul(b"".join(sks), b"".join(i[0][0] for i in ios)) == b"".join(i[0][1] for i in ios)   # gamma = sk * h
    got = ctx.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    tp = [got[96 * j: 96 * j + 96] for j in range(7)]
    assert [proof_comp(S, p, 0).hex() for p in tp] == [v["proof_r"] + v["proof_s"] for v in th]
    assert ctx.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=tp)) == [0] * 7
    assert ctx.thin_batch_verify(pks, ios, ads, tp) == 0
    st, bases, sc = orc.thin_batch_terms(S, [bytes.fromhex(v["pk"]) for v in th], [[(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))] for v in th], ads,
                                         [bytes.fromhex(v["proof_r"] + v["proof_s"]) for v in th])
    gb, gs = ctx.last_terms()
    assert st == 0 and gs == sc and gb == bases
    bad = tp[:2] + [tp[2][:70] + bytes([tp[2][70] ^ 1]) + tp[2][71:]] + tp[3:]
    assert ctx.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=bad)) == [0, 0, 1, 0, 0, 0, 0]
    assert ctx.thin_batch_verify(pks, ios, ads, bad) == 1
    assert ctx.thin_batch_verify([IDENTITY_XY] + pks[1:], ios, ads, tp) == 2
    got = ctx.tiny_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    yp = [got[48 * j: 48 * j + 48] for j in range(7)]
    assert [p.hex() for p in yp] == [v["proof_c"] + v["proof_s"] for v in ti]
    assert ctx.tiny_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=yp)) == [0] * 7
    pr, bl = ctx.pedersen_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    pp = [pr[256 * j: 256 * j + 256] for j in range(7)]
    assert [proof_comp(S, p, 1).hex() for p in pp] == [v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] for v in pe]
    assert [bl[32 * j: 32 * j + 32].hex() for j in range(7)] == [v["blinding"] for v in pe]
    assert ctx.pedersen_verify(Batch.from_items(ios, ads, proofs=pp)) == [0] * 7
    assert ctx.pedersen_batch_verify(ios, ads, pp) == 0


@pytest.mark.parametrize("kind,n", [(0, 900), (1, 400)])
def test_synthetic_batches_vs_oracle(ctx, kind, n):
    b = orc.gen_batch(S, kind, n)
    if kind == 0:
        assert ctx.thin_prove(nat_batch(b, with_sks=True, with_proofs=False)) == b["proofs"]
        assert ctx.thin_verify(nat_batch(b)) == [0] * n
        assert ctx.thin_batch_stage(nat_batch(b)) == 0 and ctx.thin_batch_run() == 0
        pks, ios, ads, proofs = compressed_items(S, b, 0)
        st, bases, sc = orc.thin_batch_terms(S, pks, ios, ads, proofs)
    else:
        b["pks_xy"] = b""
        assert ctx.pedersen_batch_stage(nat_batch(b)) == 0 and ctx.pedersen_batch_run() == 0
        _, ios, ads, pr = compressed_items(S, b, 1)
        st, bases, sc = orc.pedersen_batch_terms(S, ios, ads, pr)
    gb, gs = ctx.last_terms()
    assert st == 0 and gs == sc and gb == bases
    psz = 96 if kind == 0 else 256
    p2 = bytearray(b["proofs"]); p2[psz * (n // 3) + (64 if kind == 0 else 200)] ^= 1
    b2 = dict(b); b2["proofs"] = bytes(p2)
    if kind == 0:
        assert ctx.thin_batch_stage(nat_batch(b2)) == 0 and ctx.thin_batch_run() == 1
    else:
        assert ctx.pedersen_batch_stage(nat_batch(b2)) == 0 and ctx.pedersen_batch_run() == 1


def test_not_a_ring_suite(ctx, golden_dir):
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import RingSetup
    srs = open(os.path.join(golden_dir, "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()
    with pytest.raises(nat.AvrfError, match="-> -2"):
        RingSetup(ctx, srs, 8)
