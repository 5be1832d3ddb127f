"""GPU parity for the Pedersen VRF rows of SURVEY.md §8a (a5, a6, a7) through the C ABI:
prove bytes == oracle bytes (and == the reference's golden vectors), single verify status,
batch verifier terms/weights == oracle (src/pedersen.rs:136-249, 276-293, 341-426)."""
import json
import os

import pytest

import oracle as orc
from helpers import IDENTITY_XY, compressed_items, nat_batch, proof_comp, proof_xy, xy

pytestmark = pytest.mark.gpu
NAMES = {0: "bandersnatch_sha-512_ell2", 1: "baby-jubjub_sha-512_tai"}


@pytest.fixture(scope="module")
def ctxs():
    from ark_vrf_amd import _native as nat
    return {s: nat.Context(s) for s in (0, 1)}


def golden(golden_dir, suite):
    with open(os.path.join(golden_dir, NAMES[suite] + "_pedersen.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("suite", [0, 1])
def test_reference_vectors_prove_verify_batch(ctxs, golden_dir, suite):
    from ark_vrf_amd._native import Batch
    vs = golden(golden_dir, suite)
    c = ctxs[suite]
    ios = [[(xy(suite, bytes.fromhex(v["h"])), xy(suite, bytes.fromhex(v["gamma"])))] for v in vs]
    ads = [bytes.fromhex(v["ad"]) for v in vs]
    sks = [bytes.fromhex(v["sk"]) for v in vs]
    want = [bytes.fromhex(v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]) for v in vs]
    # prove on the GPU: byte-exact with the reference's vectors (src/pedersen.rs:793-809)
    proofs, blind = c.pedersen_prove(Batch.from_items(ios, ads, sks=sks))
    for j, v in enumerate(vs):
        assert proof_comp(suite, proofs[256 * j: 256 * j + 256], 1) == want[j]
        assert blind[32 * j: 32 * j + 32].hex() == v["blinding"]
    # same with the cached public key passed in
    pks = [xy(suite, bytes.fromhex(v["pk"])) for v in vs]
    proofs2, _ = c.pedersen_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    assert proofs2 == proofs
    pl = [proofs[256 * j: 256 * j + 256] for j in range(len(vs))]
    assert c.pedersen_verify(Batch.from_items(ios, ads, proofs=pl)) == [0] * len(vs)
    assert c.pedersen_batch_verify(ios, ads, pl) == 0
    st, bases, sc = orc.pedersen_batch_terms(suite, [[(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))] for v in vs], ads, want)
    gb, gs = c.last_terms()
    assert st == 0 and gs == sc and gb == bases            # 37 terms bit-exact
    # tamper: s, sb, Ok, ad
    bad = [bytearray(p) for p in pl]
    bad[1][192] ^= 1; bad[2][224] ^= 1; bad[3][128:192] = pl[4][128:192]
    stt = c.pedersen_verify(Batch.from_items(ios, ads[:5] + [b"zz"] + ads[6:], proofs=[bytes(x) for x in bad]))
    assert stt == [0, 1, 1, 1, 0, 1, 0]
    assert c.pedersen_batch_verify(ios, ads, [bytes(bad[1])] + pl[1:]) == 1   # item 0 carries item 1's (tampered) proof
    assert c.pedersen_batch_verify(ios, ads, pl[:2] + [bytes(bad[2])] + pl[3:]) == 1
    # ONE item per call: its two equations as two scalar vectors through the single-launch MSM (capi.hip avrf_pedersen_verify) --
    # each equation must fail on its own: s and Ok only enter the first, sb and R only the second, the challenge both
    ads_t = ads[:5] + [b"zz"] + ads[6:]
    for j in range(len(vs)):
        assert c.pedersen_verify(Batch.from_items([ios[j]], [ads[j]], proofs=[pl[j]])) == [0]
        assert c.pedersen_verify(Batch.from_items([ios[j]], [ads_t[j]], proofs=[bytes(bad[j])])) == [[0, 1, 1, 1, 0, 1, 0][j]]
    one_bad = bytearray(pl[0]); one_bad[64:128] = pl[1][64:128]            # R of another proof: second equation only
    assert c.pedersen_verify(Batch.from_items([ios[0]], [ads[0]], proofs=[bytes(one_bad)])) == [1]
    one_bad = bytearray(pl[0]); one_bad[192:224] = b"\xff" * 32             # s >= r
    assert c.pedersen_verify(Batch.from_items([ios[0]], [ads[0]], proofs=[bytes(one_bad)])) == [2]
    # identity key commitment / identity io -> InvalidData (src/pedersen.rs:204-213,348-353)
    idp = IDENTITY_XY + pl[0][64:]
    assert c.pedersen_verify(Batch.from_items(ios[:1], ads[:1], proofs=[idp])) == [2]
    assert c.pedersen_batch_verify(ios, ads, [idp] + pl[1:]) == 2
    ios_bad = [list(x) for x in ios]; ios_bad[3] = [(IDENTITY_XY, ios[3][0][1])]
    assert c.pedersen_batch_verify(ios_bad, ads, pl) == 2
    assert c.pedersen_batch_verify([], [], []) == 0


@pytest.mark.parametrize("suite", [0, 1])
def test_multi_io(ctxs, suite):
    from ark_vrf_amd._native import Batch
    c = ctxs[suite]
    sks, ios_c, ads = [], [], []
    for j, m in enumerate([0, 1, 2, 3, 1, 4, 16, 17]):       # 16, 17: the MSM branch of merge_ios (src/utils/common.rs:405-412)
        sk, _ = orc.from_seed(suite, bytes([j + 9]) + bytes(31))
        io = []
        for i in range(m):
            h = orc.hash_to_curve(suite, b"p-%d-%d" % (j, i))
            io.append((h, orc.vrf_output(suite, sk, h)))
        sks.append(sk); ios_c.append(io); ads.append(b"a" * j)
    want = [orc.pedersen_prove(suite, sk, io, ad) for sk, io, ad in zip(sks, ios_c, ads)]
    ios = [[(xy(suite, i), xy(suite, o)) for i, o in io] for io in ios_c]
    proofs, blind = c.pedersen_prove(Batch.from_items(ios, ads, sks=sks))
    pl = [proofs[256 * j: 256 * j + 256] for j in range(len(sks))]
    for j, (p, b) in enumerate(want):
        assert proof_comp(suite, pl[j], 1) == p and blind[32 * j: 32 * j + 32] == b
    assert c.pedersen_verify(Batch.from_items(ios, ads, proofs=pl)) == [0] * len(sks)
    # ONE item per call: prover and verifier through the MSM engine (capi.hip prove_ped_one_as_msm / avrf_pedersen_verify), every
    # pair count, with and without a given public key -- same bytes, same blinding, same verdicts
    for j in range(len(sks)):
        pk_j = xy(suite, orc.from_seed(suite, bytes([j + 9]) + bytes(31))[1])
        p1, b1 = c.pedersen_prove(Batch.from_items([ios[j]], [ads[j]], sks=[sks[j]]))
        p2, b2 = c.pedersen_prove(Batch.from_items([ios[j]], [ads[j]], sks=[sks[j]], pks_xy=[pk_j]))
        assert p1 == pl[j] and p2 == pl[j] and b1 == blind[32 * j: 32 * j + 32] and b2 == b1
        assert c.pedersen_verify(Batch.from_items([ios[j]], [ads[j]], proofs=[pl[j]])) == [0]
        assert c.pedersen_verify(Batch.from_items([ios[j]], [ads[j] + b"x"], proofs=[pl[j]])) == [1]
    assert c.pedersen_batch_verify(ios, ads, pl) == 0
    st, bases, sc = orc.pedersen_batch_terms(suite, ios_c, ads, [w[0] for w in want])
    gb, gs = c.last_terms()
    assert st == 0 and gs == sc and gb == bases


@pytest.mark.parametrize("suite,n", [(0, 1), (0, 700), (1, 300)])
def test_synthetic_batch(ctxs, suite, n):
    b = orc.gen_batch(suite, 1, n)
    c = ctxs[suite]
    # GPU prover reproduces the oracle's proofs byte for byte
    proofs, _ = c.pedersen_prove(nat_batch(b, with_sks=True, with_proofs=False))
    assert proofs == b["proofs"]
    assert c.pedersen_verify(nat_batch(b)) == [0] * n
    assert c.pedersen_batch_stage(nat_batch(b)) == 0 and c.pedersen_batch_run() == 0
    _, ios, ads, pr = compressed_items(suite, b, 1)
    st, bases, sc = orc.pedersen_batch_terms(suite, ios, ads, pr)
    gb, gs = c.last_terms()
    assert st == 0 and gs == sc and gb == bases
    p2 = bytearray(b["proofs"]); p2[256 * (n // 2) + 200] ^= 8
    b2 = dict(b); b2["proofs"] = bytes(p2)
    assert c.pedersen_batch_stage(nat_batch(b2)) == 0 and c.pedersen_batch_run() == 1
    stl = c.pedersen_verify(nat_batch(b2))
    assert stl[n // 2] == 1 and sum(stl) == 1
