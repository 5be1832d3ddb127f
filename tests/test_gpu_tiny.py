"""GPU parity: Tiny VRF (src/tiny.rs:163-214) through avrf_tiny_prove / avrf_tiny_verify against the reference's
`*_tiny.json` vectors (proof_c, proof_s; asserted by the reference in src/tiny.rs tests via src/testing.rs:263-280) and the
oracle (oracle/orc_vrf.c orc_tiny_*), byte for byte."""
import json
import os

import pytest

import oracle as orc
from helpers import IDENTITY_XY, xy

pytestmark = pytest.mark.gpu
NAMES = {0: "bandersnatch_sha-512_ell2", 1: "baby-jubjub_sha-512_tai"}


@pytest.fixture(scope="module")
def ctxs():
    from ark_vrf_amd import _native as nat
    return {s: nat.Context(s) for s in (0, 1)}


@pytest.mark.parametrize("suite", [0, 1])
def test_reference_vectors(ctxs, golden_dir, suite):
    from ark_vrf_amd._native import Batch
    c = ctxs[suite]
    vs = json.load(open(os.path.join(golden_dir, NAMES[suite] + "_tiny.json")))
    sks = [bytes.fromhex(v["sk"]) for v in vs]
    pks = [xy(suite, bytes.fromhex(v["pk"])) for v in vs]
    ios = [[(xy(suite, bytes.fromhex(v["h"])), xy(suite, bytes.fromhex(v["gamma"])))] for v in vs]
    ads = [bytes.fromhex(v["ad"]) for v in vs]
    want = [bytes.fromhex(v["proof_c"] + v["proof_s"]) for v in vs]
    got = c.tiny_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    assert [got[48 * j: 48 * j + 48] for j in range(7)] == want
    got2 = c.tiny_prove(Batch.from_items(ios, ads, sks=sks))                       # public key derived on the device
    assert got2 == got
    assert c.tiny_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=want)) == [0] * 7
    # tampered c / s / ad / output; identity public key and identity input (tiny.rs:186-198)
    bad = list(want); bad[1] = bytes([bad[1][0] ^ 1]) + bad[1][1:]; bad[2] = bad[2][:20] + bytes([bad[2][20] ^ 4]) + bad[2][21:]
    assert c.tiny_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=bad)) == [0, 1, 1, 0, 0, 0, 0]
    assert c.tiny_verify(Batch.from_items(ios, [b"x"] + ads[1:], pks_xy=pks, proofs=want)) == [1, 0, 0, 0, 0, 0, 0]
    ios_bad = [list(x) for x in ios]; ios_bad[4] = [(ios[4][0][0], ios[3][0][1])]; ios_bad[5] = [(IDENTITY_XY, ios[5][0][1])]
    assert c.tiny_verify(Batch.from_items(ios_bad, ads, pks_xy=pks[:6] + [IDENTITY_XY], proofs=want)) == [0, 0, 0, 0, 1, 2, 2]
    r = {0: 0x1cfb69d4ca675f520cce760202687600ff8f87007419047174fd06b52876e7e1,
         1: 2736030358979909402780800718157159386076813972158567259200215660948447373041}[suite]
    big = want[0][:16] + r.to_bytes(32, "little")                                  # s >= r
    assert c.tiny_verify(Batch.from_items(ios[:1], ads[:1], pks_xy=pks[:1], proofs=[big])) == [2]


@pytest.mark.parametrize("suite", [0, 1])
def test_multi_io_and_synthetic_vs_oracle(ctxs, suite):
    """prove_verify_multi / _multi_empty (src/tiny.rs tests): 0 .. 17 I/O pairs (both sides of MSM_THRESHOLD)."""
    from ark_vrf_amd._native import Batch
    c = ctxs[suite]
    sks, pks, ios_c, ads = [], [], [], []
    for j, m in enumerate([0, 1, 2, 3, 7, 16, 17] + [1] * 60):
        sk, pk = orc.from_seed(suite, bytes([j + 1, 9]) + bytes(30))
        io = []
        for i in range(m):
            h = orc.hash_to_curve(suite, b"tiny-%d-%d" % (j, i))
            io.append((h, orc.vrf_output(suite, sk, h)))
        sks.append(sk); pks.append(pk); ios_c.append(io); ads.append(b"ad" * (j % 4))
    want = [orc.tiny_prove(suite, sk, io, ad) for sk, io, ad in zip(sks, ios_c, ads)]
    ios = [[(xy(suite, i), xy(suite, o)) for i, o in io] for io in ios_c]
    pkl = [xy(suite, p) for p in pks]
    got = c.tiny_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pkl))
    pl = [got[48 * j: 48 * j + 48] for j in range(len(sks))]
    assert pl == want
    assert c.tiny_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=pl)) == [0] * len(sks)
    assert all(orc.tiny_verify(suite, pk, io, ad, p) == 0 for pk, io, ad, p in zip(pks[:8], ios_c[:8], ads[:8], pl[:8]))
    ios[5][15] = (ios[5][15][0], ios[5][14][1])                                   # last pair of the 16-pair item
    st = c.tiny_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=pl))
    assert st[5] == 1 and sum(st) == 1
    assert c.tiny_prove(Batch.from_items([], [], sks=[])) == b""
