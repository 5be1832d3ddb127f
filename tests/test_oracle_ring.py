"""Pins the ring-SNARK oracle (oracle/ring_py.py) to the reference's ring vectors:
ring commitment (`ring_pks_com`, 3 x G1) and the complete deterministic ring proof (`ring_proof`,
592 B BLS12-381 / 480 B BN254) of tests/golden/*_ring.json, with the SRS files the reference's
tests load (src/ring.rs:1412-1421, 1529-1571)."""
import json
import os

import pytest

from oracle import ring_py as R

FILES = {0: ("bandersnatch_sha-512_ell2_ring.json", "bls12-381-srs-2-11-uncompressed-zcash.bin"),
         1: ("baby-jubjub_sha-512_tai_ring.json", "bn254-testing-2-9-uncompressed.bin"),
         2: ("jubjub_sha-512_tai_ring.json", "bls12-381-srs-2-11-uncompressed-zcash.bin")}


@pytest.fixture(scope="module")
def setups(golden_dir):
    out = {}
    for i, (vec, srsf) in FILES.items():
        s = R.SUITES[i]
        srs = R.Srs(s, open(os.path.join(golden_dir, srsf), "rb").read())
        out[i] = (s, srs, json.load(open(os.path.join(golden_dir, vec))), R.Params(s, ring_size=8))
    return out


def test_domain_sizes(setups):
    # src/ring.rs:810-843: piop = next_pow2(ring + 4 + L); pcs = 3*piop + 1
    for i, L in ((0, 253), (1, 251), (2, 252)):
        s = R.SUITES[i]
        assert R.Params(s, ring_size=8).N == 512
        assert R.Params(s, ring_size=1024).N == 2048
        assert R.Params(s, ring_size=4096).N == 8192
        p = R.Params(s, ring_size=8)
        assert p.keyset_part_size == 512 - 4 - L and p.capacity == 509


@pytest.mark.parametrize("suite,vec", [(s, v) for s in (0, 1) for v in range(7)] + [(2, 0), (2, 4)])
def test_ring_commitment_and_proof(setups, suite, vec):
    s, srs, vs, prm = setups[suite]
    v = vs[vec]
    pks = bytes.fromhex(v["ring_pks"])
    keys = [R.te_decode(s, pks[32 * i: 32 * i + 32]) for i in range(len(pks) // 32)]
    cols = R.index(prm, srs, keys)
    assert R.commitment_bytes(s, cols).hex() == v["ring_pks_com"]
    idx = keys.index(R.te_decode(s, bytes.fromhex(v["pk"])))
    assert idx == 3                                                   # src/ring.rs:1477
    b = int.from_bytes(bytes.fromhex(v["blinding"]), "little")
    proof, instance = R.prove(prm, srs, cols, idx, b)
    assert proof.hex() == v["ring_proof"]
    # the instance is the Pedersen key commitment Yb of the same vector (src/ring.rs:237-242)
    assert instance == R.te_decode(s, bytes.fromhex(v["proof_pk_com"]))


def test_srs_pairing_consistency(setups):
    """e(tau*g1, g2) == e(g1, tau*g2) on the reference's BLS12-381 SRS file (pins the pairing and the
    zcash G2 byte order)."""
    from oracle import pairing_py as PP
    s, srs, _, _ = setups[0]
    PP.use_curve("bls12_381")
    g2, tg2 = PP.g2_decode_zcash_uncompressed(srs.g2_raw[0]), PP.g2_decode_zcash_uncompressed(srs.g2_raw[1])
    g1, tg1 = srs.g1[0], srs.g1[1]
    neg = lambda P: (P[0], (-P[1]) % s.p)
    assert PP.pairing_product_is_one([(tg1, g2), (neg(g1), tg2)])
    assert not PP.pairing_product_is_one([(tg1, g2), (neg(tg1), tg2)])
    # powers of tau are consistent further up the SRS too
    assert PP.pairing_product_is_one([(srs.g1[5], g2), (neg(srs.g1[4]), tg2)])


def test_bn254_srs_pairing_consistency(setups):
    """Same for BN254 (arkworks little-endian G2: x.c0 || x.c1 || y.c0 || y.c1) -- the survey left the BN254
    pairing unprobed; the reference's SRS file pins it here."""
    from oracle import pairing_py as PP
    s, srs, _, _ = setups[1]
    PP.use_curve("bn254")
    g2, tg2 = PP.g2_decode_arkworks_uncompressed(srs.g2_raw[0]), PP.g2_decode_arkworks_uncompressed(srs.g2_raw[1])
    neg = lambda P: (P[0], (-P[1]) % s.p)
    assert PP.pairing_product_is_one([(srs.g1[1], g2), (neg(srs.g1[0]), tg2)])
    assert PP.pairing_product_is_one([(srs.g1[9], g2), (neg(srs.g1[8]), tg2)])
    assert not PP.pairing_product_is_one([(srs.g1[1], g2), (neg(srs.g1[1]), tg2)])
    PP.use_curve("bls12_381")


@pytest.mark.parametrize("suite,vec", [(0, 0), (0, 3), (0, 6), (1, 0), (1, 5)])
def test_reference_ring_proofs_verify(setups, suite, vec):
    """RingVerifier::verify (src/ring.rs:242) restated per SURVEY.md A.8 accepts the reference's own
    proofs and rejects perturbed ones (both pairing curves)."""
    s, srs, vs, prm = setups[suite]
    v = vs[vec]
    com = bytes.fromhex(v["ring_pks_com"])
    n48 = s.fp_bytes
    fixed = [R.g1_decode_compressed(s, com[n48 * i: n48 * i + n48]) for i in range(3)]
    inst = R.te_decode(s, bytes.fromhex(v["proof_pk_com"]))
    proof = bytes.fromhex(v["ring_proof"])
    assert R.verify(prm, srs, fixed, proof, inst)
    bad = bytearray(proof); bad[4 * n48 + 3 * 32 + 1] ^= 1              # evaluation of `bits`
    assert not R.verify(prm, srs, fixed, bytes(bad), inst)
    assert not R.verify(prm, srs, fixed, proof, R.te_add(s, inst, s.blinding_base))   # other key commitment
    other = bytes.fromhex(vs[(vec + 1) % 7]["ring_pks_com"])
    if other != com:                                                   # other ring
        fixed2 = [R.g1_decode_compressed(s, other[n48 * i: n48 * i + n48]) for i in range(3)]
        assert not R.verify(prm, srs, fixed2, proof, inst)
