"""Pins the CPU oracle to the reference's own known-answer vectors.

Vectors: tests/golden/*.json are verbatim copies of /root/reference/data/vectors/
{bandersnatch_sha-512_ell2,baby-jubjub_sha-512_tai}_{thin,pedersen}.json -- the files the
reference asserts in src/testing.rs:263-280, src/thin.rs:635-648, src/pedersen.rs:793-809.
Seeds per src/testing.rs:291-299.
"""
import hashlib
import json
import os

import pytest

import oracle as orc

SUITES = {
    "bandersnatch_sha-512_ell2": orc.BANDERSNATCH,
    "baby-jubjub_sha-512_tai": orc.BABYJUBJUB,
    "jubjub_sha-512_tai": orc.JUBJUB,                 # src/suites/jubjub.rs
    "ed25519_sha-512_tai": orc.ED25519,               # src/suites/ed25519.rs (Tiny / Thin / Pedersen; no ring suite)
    "testing_sha-256_tai": orc.TESTING_SHA256,        # src/suites/testing.rs: edwards25519, HashTranscript<Sha256>
    "bandersnatch_shake128_ell2": orc.BANDERSNATCH_SHAKE128,   # src/suites/bandersnatch_shake128.rs: SHAKE128 sponge transcript, expand_message_xof
}
SEEDS = [1, 2, 3, 4, 5, 5, 6]


def load(golden_dir, name, scheme):
    with open(os.path.join(golden_dir, f"{name}_{scheme}.json")) as f:
        return json.load(f)


def test_sha512_matches_hashlib():
    for n in [0, 1, 55, 111, 112, 113, 127, 128, 129, 255, 256, 1000]:
        d = bytes((i * 7 + n) & 0xFF for i in range(n))
        assert orc.sha512(d) == hashlib.sha512(d).digest()


@pytest.mark.parametrize("name", list(SUITES))
@pytest.mark.parametrize("scheme", ["thin", "pedersen"])
def test_base_fields(golden_dir, name, scheme):
    s = SUITES[name]
    for i, v in enumerate(load(golden_dir, name, scheme)):
        sk, pk = bytes.fromhex(v["sk"]), bytes.fromhex(v["pk"])
        seed = bytes([SEEDS[i]]) + bytes(31)
        assert orc.from_seed(s, seed) == (sk, pk)                      # src/lib.rs:346-369
        assert orc.sk_to_pk(s, sk) == pk
        h = orc.hash_to_curve(s, bytes.fromhex(v["alpha"]))           # src/testing.rs:271
        assert h.hex() == v["h"]
        gamma = orc.vrf_output(s, sk, h)                               # src/testing.rs:275
        assert gamma.hex() == v["gamma"]
        assert orc.point_to_hash(s, gamma).hex() == v["beta"]          # src/testing.rs:278


@pytest.mark.parametrize("name", list(SUITES))
def test_thin_vectors(golden_dir, name):
    s = SUITES[name]
    vs = load(golden_dir, name, "thin")
    pks, ios, ads, proofs = [], [], [], []
    for v in vs:
        sk, pk = bytes.fromhex(v["sk"]), bytes.fromhex(v["pk"])
        io = [(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))]
        ad = bytes.fromhex(v["ad"])
        proof = orc.thin_prove(s, sk, io, ad)                          # src/thin.rs:635-648
        assert proof.hex() == v["proof_r"] + v["proof_s"]
        assert orc.thin_verify(s, pk, io, ad, proof) == orc.OK
        bad = bytearray(proof); bad[40] ^= 1
        assert orc.thin_verify(s, pk, io, ad, bytes(bad)) == orc.VERIFICATION_FAILURE
        assert orc.thin_verify(s, pk, io, ad + b"x", proof) == orc.VERIFICATION_FAILURE
        pks.append(pk); ios.append(io); ads.append(ad); proofs.append(proof)
    # batch equation on the reference's 7 proofs: 29-term MSM == identity
    st, bases, sc = orc.thin_batch_terms(s, pks, ios, ads, proofs)
    assert st == orc.OK and len(sc) == 32 * 29
    ident = bytes(32) + (1).to_bytes(32, "little")
    assert orc.msm(s, bases, sc, algo=1) == ident
    assert orc.msm(s, bases, sc, algo=0) == ident
    assert orc.thin_batch_verify(s, pks, ios, ads, proofs) == orc.OK
    bad = bytearray(proofs[3]); bad[33] ^= 4
    assert orc.thin_batch_verify(s, pks, ios, ads, proofs[:3] + [bytes(bad)] + proofs[4:]) == orc.VERIFICATION_FAILURE


@pytest.mark.parametrize("name", list(SUITES))
def test_pedersen_vectors(golden_dir, name):
    s = SUITES[name]
    vs = load(golden_dir, name, "pedersen")
    ios, ads, proofs = [], [], []
    for v in vs:
        sk = bytes.fromhex(v["sk"])
        io = [(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))]
        ad = bytes.fromhex(v["ad"])
        proof, blinding = orc.pedersen_prove(s, sk, io, ad)            # src/pedersen.rs:793-809
        assert blinding.hex() == v["blinding"]
        assert proof.hex() == v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]
        assert orc.pedersen_verify(s, io, ad, proof) == orc.OK
        bad = bytearray(proof); bad[100] ^= 1
        assert orc.pedersen_verify(s, io, ad, bytes(bad)) == orc.VERIFICATION_FAILURE
        ios.append(io); ads.append(ad); proofs.append(proof)
    st, bases, sc = orc.pedersen_batch_terms(s, ios, ads, proofs)
    assert st == orc.OK and len(sc) == 32 * 37
    ident = bytes(32) + (1).to_bytes(32, "little")
    assert orc.msm(s, bases, sc, algo=1) == ident
    assert orc.pedersen_batch_verify(s, ios, ads, proofs) == orc.OK
    bad = bytearray(proofs[2]); bad[130] ^= 4
    assert orc.pedersen_batch_verify(s, ios, ads, proofs[:2] + [bytes(bad)] + proofs[3:]) == orc.VERIFICATION_FAILURE


@pytest.mark.parametrize("name", list(SUITES))
def test_suite_constants(golden_dir, name):
    """BLINDING_BASE / ACCUMULATOR_BASE / PADDING are hash-to-curve outputs
    (src/pedersen.rs:39,568-579 `blinding_base_check`; src/ring.rs:66-69 `padding_check`, `accumulator_base_check`)."""
    s = SUITES[name]
    assert orc.hash_to_curve(s, b"pedersen-blinding") == orc.suite_point(s, 1)
    if s in (orc.ED25519, orc.TESTING_SHA256):             # not a RingSuite: no accumulator base / padding point
        return
    assert orc.hash_to_curve(s, b"ring-accumulator") == orc.suite_point(s, 2)
    assert orc.hash_to_curve(s, b"ring-padding") == orc.suite_point(s, 3)


@pytest.mark.parametrize("name", list(SUITES))
def test_tiny_vectors(golden_dir, name):
    """Tiny VRF (src/tiny.rs:163-214): proof_c / proof_s of the reference's `*_tiny.json`, prove and verify."""
    import json, os
    s = SUITES[name]
    vs = json.load(open(os.path.join(golden_dir, name + "_tiny.json")))
    assert len(vs) == 7
    for v in vs:
        sk, pk = bytes.fromhex(v["sk"]), bytes.fromhex(v["pk"])
        io = [(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))]
        ad = bytes.fromhex(v["ad"])
        proof = orc.tiny_prove(s, sk, io, ad)
        assert proof.hex() == v["proof_c"] + v["proof_s"]
        assert orc.tiny_verify(s, pk, io, ad, proof) == orc.OK
        bad = bytearray(proof); bad[17] ^= 1
        assert orc.tiny_verify(s, pk, io, ad, bytes(bad)) == orc.VERIFICATION_FAILURE
        assert orc.tiny_verify(s, pk, io, ad + b"x", proof) == orc.VERIFICATION_FAILURE


@pytest.mark.parametrize("scheme", ["thin", "tiny", "pedersen"])
def test_bandersnatch_sw_vectors(golden_dir, scheme):
    """Bandersnatch-SW-SHA512-TAI (src/suites/bandersnatch_sw.rs): the same curve in its short-Weierstrass presentation -- every
    serialised point is the 33-byte SW form (orc.sw_encode / sw_decode, src/utils/te_sw_map.rs), try-and-increment runs on SW
    x-coordinates, transcripts absorb the 33-byte forms.  All 21 vectors: sk -> pk, alpha -> h, gamma, beta, proofs."""
    s = orc.BANDERSNATCH_SW
    te = lambda h: (lambda r: (r[1] if r[0] == 0 else None))(orc.sw_decode(s, bytes.fromhex(h)))
    for i, v in enumerate(load(golden_dir, "bandersnatch_sw_sha-512_tai", scheme)):
        sk, pk = bytes.fromhex(v["sk"]), te(v["pk"])
        assert orc.from_seed(s, bytes([SEEDS[i]]) + bytes(31)) == (sk, pk)
        assert orc.sw_encode(s, pk).hex() == v["pk"]
        h = orc.hash_to_curve(s, bytes.fromhex(v["alpha"]))
        assert orc.sw_encode(s, h).hex() == v["h"]
        gamma = orc.vrf_output(s, sk, h)
        assert orc.sw_encode(s, gamma).hex() == v["gamma"] and orc.point_to_hash(s, gamma).hex() == v["beta"]
        io, ad = [(h, gamma)], bytes.fromhex(v["ad"])
        if scheme == "thin":
            pr = orc.thin_prove(s, sk, io, ad)
            assert orc.sw_encode(s, pr[:32]).hex() + pr[32:].hex() == v["proof_r"] + v["proof_s"]
            assert orc.thin_verify(s, pk, io, ad, pr) == orc.OK
            assert orc.thin_verify(s, pk, io, ad + b"x", pr) == orc.VERIFICATION_FAILURE
        elif scheme == "tiny":
            pr = orc.tiny_prove(s, sk, io, ad)
            assert pr.hex() == v["proof_c"] + v["proof_s"] and orc.tiny_verify(s, pk, io, ad, pr) == orc.OK
        else:
            pr, bl = orc.pedersen_prove(s, sk, io, ad)
            assert bl.hex() == v["blinding"]
            assert "".join(orc.sw_encode(s, pr[32 * k: 32 * k + 32]).hex() for k in range(3)) + pr[96:].hex() == \
                v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]
            assert orc.pedersen_verify(s, io, ad, pr) == orc.OK
    # BLINDING_BASE and PADDING are hash-to-curve outputs; ACCUMULATOR_BASE is NOT for an SW suite (the reference adds a point
    # outside the prime-order subgroup, src/ring.rs:896-910) -- it is the reference's constant, mapped
    assert orc.hash_to_curve(s, b"pedersen-blinding") == orc.suite_point(s, 1)
    assert orc.hash_to_curve(s, b"ring-padding") == orc.suite_point(s, 3)
    assert orc.hash_to_curve(s, b"ring-accumulator") != orc.suite_point(s, 2)
    # infinity and unused flag bits do not decode; the round trip keeps the flag byte
    assert orc.sw_decode(s, bytes(32) + b"\x40")[0] != 0 and orc.sw_decode(s, bytes.fromhex(v["pk"])[:32] + b"\x01")[0] != 0


@pytest.mark.parametrize("scheme", ["thin", "tiny", "pedersen"])
def test_secp256r1_vectors(golden_dir, scheme):
    """Secp256r1-SHA256-TAI (src/suites/secp256r1.rs:49-70): NIST P-256, a genuinely short-Weierstrass suite -- a = -3 Jacobian
    group law, 256-bit base / scalar fields with the top bit set, HashTranscript<Sha256>, try-and-increment on SW x-coordinates,
    33-byte points (LE32(x) || flags).  All 21 vectors: sk -> pk, alpha -> h, gamma, beta, proofs; then the batch verifiers'
    MSMs on the reference's proofs (identity of a short-Weierstrass group in the xy flavour: all zero bytes)."""
    s = orc.SECP256R1
    vs = load(golden_dir, "secp256r1_sha-256_tai", scheme)
    assert len(vs) == 7
    pks, ios, ads, proofs = [], [], [], []
    for i, v in enumerate(vs):
        sk, pk = bytes.fromhex(v["sk"]), bytes.fromhex(v["pk"])
        assert len(pk) == 33
        assert orc.from_seed(s, bytes([SEEDS[i]]) + bytes(31)) == (sk, pk)
        assert orc.sk_to_pk(s, sk) == pk
        h = orc.hash_to_curve(s, bytes.fromhex(v["alpha"]))
        assert h.hex() == v["h"]
        gamma = orc.vrf_output(s, sk, h)
        assert gamma.hex() == v["gamma"] and orc.point_to_hash(s, gamma).hex() == v["beta"]
        io, ad = [(h, gamma)], bytes.fromhex(v["ad"])
        if scheme == "thin":
            pr = orc.thin_prove(s, sk, io, ad)
            assert pr.hex() == v["proof_r"] + v["proof_s"] and len(pr) == 65
            assert orc.thin_verify(s, pk, io, ad, pr) == orc.OK
            bad = bytearray(pr); bad[40] ^= 1
            assert orc.thin_verify(s, pk, io, ad, bytes(bad)) == orc.VERIFICATION_FAILURE
            assert orc.thin_verify(s, pk, io, ad + b"x", pr) == orc.VERIFICATION_FAILURE
        elif scheme == "tiny":
            pr = orc.tiny_prove(s, sk, io, ad)
            assert pr.hex() == v["proof_c"] + v["proof_s"] and orc.tiny_verify(s, pk, io, ad, pr) == orc.OK
            bad = bytearray(pr); bad[17] ^= 1
            assert orc.tiny_verify(s, pk, io, ad, bytes(bad)) == orc.VERIFICATION_FAILURE
        else:
            pr, bl = orc.pedersen_prove(s, sk, io, ad)
            assert bl.hex() == v["blinding"] and len(pr) == 163
            assert pr.hex() == v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]
            assert orc.pedersen_verify(s, io, ad, pr) == orc.OK
            bad = bytearray(pr); bad[110] ^= 1
            assert orc.pedersen_verify(s, io, ad, bytes(bad)) == orc.VERIFICATION_FAILURE
        pks.append(pk); ios.append(io); ads.append(ad); proofs.append(pr)
    ident = bytes(64)
    if scheme == "thin":
        st, bases, sc = orc.thin_batch_terms(s, pks, ios, ads, proofs)
        assert st == orc.OK and len(sc) == 32 * 29
        assert orc.msm(s, bases, sc, algo=1) == ident and orc.msm(s, bases, sc, algo=0) == ident
        assert orc.thin_batch_verify(s, pks, ios, ads, proofs) == orc.OK
        bad = bytearray(proofs[3]); bad[34] ^= 4
        assert orc.thin_batch_verify(s, pks, ios, ads, proofs[:3] + [bytes(bad)] + proofs[4:]) == orc.VERIFICATION_FAILURE
    elif scheme == "pedersen":
        st, bases, sc = orc.pedersen_batch_terms(s, ios, ads, proofs)
        assert st == orc.OK and len(sc) == 32 * 37
        assert orc.msm(s, bases, sc, algo=1) == ident
        assert orc.pedersen_batch_verify(s, ios, ads, proofs) == orc.OK
        bad = bytearray(proofs[2]); bad[135] ^= 4
        assert orc.pedersen_batch_verify(s, ios, ads, proofs[:2] + [bytes(bad)] + proofs[3:]) == orc.VERIFICATION_FAILURE
    # BLINDING_BASE is a hash-to-curve output (src/pedersen.rs:568-579 `blinding_base_check`); the codec: infinity, unused flag bits
    assert orc.hash_to_curve(s, b"pedersen-blinding") == orc.suite_point(s, 1)
    assert orc.point_decompress(s, bytes(32) + b"\x40")[0] == 0 and orc.point_decompress(s, bytes(32) + b"\x40")[1] == bytes(64)
    assert orc.point_decompress(s, pks[0][:32] + b"\x01")[0] != 0
    st, pxy = orc.point_decompress(s, pks[0], validate=True)
    assert st == 0 and orc.point_compress(s, pxy) == pks[0]
