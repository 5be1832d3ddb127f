"""GPU parity for the sixth suite, Bandersnatch-SHAKE128-ELL2 (src/suites/bandersnatch_shake128.rs): suite 0's curve with the
SHAKE128 sponge as the Fiat-Shamir transcript (XofTranscript<Shake128>, src/utils/transcript.rs:292-293) and
expand_message_xof in hash-to-curve.  Every scheme of the C ABI against the reference's
`bandersnatch_shake128_ell2_{thin,tiny,pedersen,ring}.json` vectors and the oracle (suite id 5).  The batch verifiers' weights are
the sponge's sequential output stream: the host squeezes it, the terms kernel reads it (capi.hip)."""
import hashlib
import json
import os
import random

import pytest

import oracle as orc
from helpers import IDENTITY_XY, compressed_items, nat_batch, proof_comp, proof_xy, rand_points_xy, rand_scalar, xy

pytestmark = pytest.mark.gpu
S = 5
NAME = "bandersnatch_shake128_ell2"


@pytest.fixture(scope="module")
def ctx():
    from ark_vrf_amd import _native as nat
    c = nat.Context(nat.BANDERSNATCH_SHAKE128_ELL2)
    yield c
    c.close()


def load(golden_dir, k):
    return json.load(open(os.path.join(golden_dir, f"{NAME}_{k}.json")))


def test_msm_and_hash_to_curve(ctx, golden_dir):
    rng = random.Random(6)
    pts = rand_points_xy(rng, S, 60)
    for n in (1, 29, 700, 4097):
        bases = b"".join(pts[i % 60] for i in range(n))
        sc = b"".join(rand_scalar(rng, S) for _ in range(n))
        assert ctx.msm(bases, sc) == orc.msm(S, bases, sc)
    vs = load(golden_dir, "thin")
    msgs = [bytes.fromhex(v["alpha"]) for v in vs] + [hashlib.sha512(b"sh%d" % i).digest()[: i % 65] for i in range(100)]
    xy_, st = ctx.hash_to_curve(msgs)
    got = ctx.points_compress(xy_)
    assert all(s == 0 for s in st)
    assert [got[32 * i: 32 * i + 32].hex() for i in range(7)] == [v["h"] for v in vs]                  # Elligator2 over expand_message_xof, alpha -> h
    assert all(got[32 * i: 32 * i + 32] == orc.hash_to_curve(S, msgs[i]) for i in range(7, len(msgs)))
    # point codec with Validate::Yes: a point of order 2 is on the curve but not in the prime-order subgroup (cofactor 4)
    q = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
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


def test_ring_vectors(ctx, golden_dir):
    """ring_proof::index, RingProver::prove and the verifiers on the reference's SHAKE128-suite ring vectors (BLS12-381 SRS file)."""
    import ctypes as C
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import RingSetup, ring_batch_verify, ring_verify_each
    vs = load(golden_dir, "ring")
    srs = open(os.path.join(golden_dir, "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()
    setup = RingSetup(ctx, srs, 8)
    assert setup.domain_size == 512 and setup.max_ring_size == 512 - 4 - 253 and (setup.proof_len, setup.commitment_len) == (592, 144)
    coms, insts, proofs = [], [], []
    for v in vs:
        raw = bytes.fromhex(v["ring_pks"])
        pks = [xy(S, raw[32 * i: 32 * i + 32]) for i in range(len(raw) // 32)]
        key = setup.index(pks)
        assert key.commitment.hex() == v["ring_pks_com"]
        idx = [raw[32 * i: 32 * i + 32].hex() for i in range(len(pks))].index(v["pk"])
        proof = key.prove([idx], [bytes.fromhex(v["blinding"])])[0]
        assert proof.hex() == v["ring_proof"]
        out = (C.c_uint8 * (160 + 592))()
        io_xy = xy(S, bytes.fromhex(v["h"])) + xy(S, bytes.fromhex(v["gamma"]))
        ad = bytes.fromhex(v["ad"])
        assert nat.lib().avrf_ring_vrf_prove(ctx._h, key._h, C.c_size_t(592), C.c_size_t(1), nat._u8(bytes.fromhex(v["sk"])), nat._u32([idx]),
                                             nat._u8(io_xy), nat._u32([1]), nat._u8(ad), nat._u32([len(ad)]), 0, out) == 0
        assert bytes(out).hex() == v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] + v["ring_proof"]
        coms.append(key.commitment); insts.append(xy(S, bytes.fromhex(v["proof_pk_com"]))); proofs.append(proof)
        key.close()
    assert ring_batch_verify(setup, coms, list(range(7)), insts, proofs) == 0
    assert ring_verify_each(setup, coms, list(range(7)), insts, proofs) == [0] * 7
    bad = list(proofs); bad[1] = bad[1][:300] + bytes([bad[1][300] ^ 1]) + bad[1][301:]
    assert ring_batch_verify(setup, coms, list(range(7)), insts, bad) in (1, 2)
    st = ring_verify_each(setup, coms, list(range(7)), insts, bad)
    assert st[1] in (1, 2) and st[:1] + st[2:] == [0] * 6
    setup.close()
