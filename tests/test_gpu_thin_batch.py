"""GPU parity: thin::BatchVerifier (src/thin.rs:188-326) through the C ABI vs the oracle.

Bit-exact checks: the MSM terms (bases, weights, scalars) the device builds equal the oracle's
restatement of src/thin.rs:282-317 on the same inputs; accept / reject / InvalidData agree.
Golden: the reference's own 7 thin vectors (tests/golden) must verify as one batch."""
import json
import os

import pytest

import oracle as orc
from helpers import IDENTITY_XY, compressed_items, nat_batch, xy

pytestmark = pytest.mark.gpu

NAMES = {0: "bandersnatch_sha-512_ell2", 1: "baby-jubjub_sha-512_tai"}


@pytest.fixture(scope="module")
def ctxs():
    from ark_vrf_amd import _native as nat
    return {s: nat.Context(s) for s in (0, 1)}


def golden_items(golden_dir, suite):
    with open(os.path.join(golden_dir, NAMES[suite] + "_thin.json")) as f:
        vs = json.load(f)
    pks = [bytes.fromhex(v["pk"]) for v in vs]
    ios = [[(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))] for v in vs]
    ads = [bytes.fromhex(v["ad"]) for v in vs]
    proofs = [bytes.fromhex(v["proof_r"] + v["proof_s"]) for v in vs]
    return pks, ios, ads, proofs


def to_abi(suite, pks, ios, ads, proofs):
    return ([xy(suite, p) for p in pks], [[(xy(suite, i), xy(suite, o)) for i, o in it] for it in ios], ads,
            [xy(suite, p[:32]) + p[32:] for p in proofs])


@pytest.mark.parametrize("suite", [0, 1])
def test_reference_vectors_batch(ctxs, golden_dir, suite):
    pks, ios, ads, proofs = golden_items(golden_dir, suite)
    a = to_abi(suite, pks, ios, ads, proofs)
    assert ctxs[suite].thin_batch_verify(*a) == 0
    st, bases, sc = orc.thin_batch_terms(suite, pks, ios, ads, proofs)
    assert st == 0
    gb, gs = ctxs[suite].last_terms()
    assert gs == sc and gb == bases          # 29 terms, bit-exact weights and scalars
    # tamper s of one proof -> VerificationFailure (src/thin.rs:320-322)
    bad = bytearray(a[3][4]); bad[64 + 3] ^= 1
    assert ctxs[suite].thin_batch_verify(a[0], a[1], a[2], a[3][:4] + [bytes(bad)] + a[3][5:]) == 1
    # wrong ad
    assert ctxs[suite].thin_batch_verify(a[0], a[1], [b"x"] + a[2][1:], a[3]) == 1


@pytest.mark.parametrize("suite", [0, 1])
def test_empty_and_identity(ctxs, golden_dir, suite):
    assert ctxs[suite].thin_batch_verify([], [], [], []) == 0                # src/thin.rs:262-264
    pks, ios, ads, proofs = golden_items(golden_dir, suite)
    a = to_abi(suite, pks, ios, ads, proofs)
    assert ctxs[suite].thin_batch_verify([IDENTITY_XY] + a[0][1:], a[1], a[2], a[3]) == 2   # identity pk
    ios_bad = [list(x) for x in a[1]]
    ios_bad[2] = [(IDENTITY_XY, ios_bad[2][0][1])]
    assert ctxs[suite].thin_batch_verify(a[0], ios_bad, a[2], a[3]) == 2     # identity input
    ios_bad[2] = [(a[1][2][0][0], IDENTITY_XY)]
    assert ctxs[suite].thin_batch_verify(a[0], ios_bad, a[2], a[3]) == 2     # identity output


@pytest.mark.parametrize("suite", [0, 1])
def test_multi_io_items(ctxs, suite):
    """Items with 0, 1, 2, 3, 5 and -- around MSM_THRESHOLD of merge_ios (src/utils/common.rs:397-412) -- 16 and 17 I/O
    pairs (prove_verify_multi / _multi_empty, src/thin.rs:390-470)."""
    pks, ios, ads, proofs = [], [], [], []
    for j, m in enumerate([0, 1, 2, 3, 5, 1, 0, 2, 16, 17]):
        sk, pk = orc.from_seed(suite, bytes([j + 1]) + bytes(31))
        io = []
        for i in range(m):
            h = orc.hash_to_curve(suite, b"in-%d-%d" % (j, i))
            io.append((h, orc.vrf_output(suite, sk, h)))
        ad = b"ad%d" % j * (j % 3)
        pr = orc.thin_prove(suite, sk, io, ad)
        assert orc.thin_verify(suite, pk, io, ad, pr) == 0
        pks.append(pk); ios.append(io); ads.append(ad); proofs.append(pr)
    a = to_abi(suite, pks, ios, ads, proofs)
    assert ctxs[suite].thin_batch_verify(*a) == 0
    st, bases, sc = orc.thin_batch_terms(suite, pks, ios, ads, proofs)
    gb, gs = ctxs[suite].last_terms()
    assert st == 0 and gs == sc and gb == bases
    a[3][3] = a[3][3][:70] + bytes([a[3][3][70] ^ 2]) + a[3][3][71:]
    assert ctxs[suite].thin_batch_verify(*a) == 1


@pytest.mark.parametrize("suite,n", [(0, 1), (0, 2), (0, 130), (0, 3000), (1, 500)])
def test_synthetic_batch_terms_and_verdict(ctxs, suite, n):
    b = orc.gen_batch(suite, 0, n)
    c = ctxs[suite]
    assert c.thin_batch_stage(nat_batch(b)) == 0
    assert c.thin_batch_run() == 0
    pks, ios, ads, proofs = compressed_items(suite, b, 0)
    st, bases, sc = orc.thin_batch_terms(suite, pks, ios, ads, proofs)
    gb, gs = c.last_terms()
    assert st == 0 and gs == sc and gb == bases
    # one tampered response scalar anywhere in the batch flips the verdict
    j = n // 2
    pr = bytearray(b["proofs"]); pr[96 * j + 64] ^= 1
    b2 = dict(b); b2["proofs"] = bytes(pr)
    assert c.thin_batch_stage(nat_batch(b2)) == 0
    assert c.thin_batch_run() == 1
