"""GPU parity for Thin VRF prove / single verify and the key / codec helpers (SURVEY.md §8a rows
a3, a4, a14): proofs byte-exact with the reference's vectors and with the oracle
(src/thin.rs:111-165, src/lib.rs:331-334,391-393; ark-serialize point codec)."""
import json
import os
import random

import pytest

import oracle as orc
from helpers import IDENTITY_XY, R_ORDER, nat_batch, proof_comp, rand_scalar, xy, proof_xy

pytestmark = pytest.mark.gpu
NAMES = {0: "bandersnatch_sha-512_ell2", 1: "baby-jubjub_sha-512_tai"}


@pytest.fixture(scope="module")
def ctxs():
    from ark_vrf_amd import _native as nat
    return {s: nat.Context(s) for s in (0, 1)}


@pytest.mark.parametrize("suite", [0, 1])
def test_reference_vectors(ctxs, golden_dir, suite):
    from ark_vrf_amd._native import Batch
    with open(os.path.join(golden_dir, NAMES[suite] + "_thin.json")) as f:
        vs = json.load(f)
    c = ctxs[suite]
    sks = [bytes.fromhex(v["sk"]) for v in vs]
    # pk = sk*G, gamma = sk*h  (src/testing.rs:266-275)
    pks = c.scalar_mul_base(b"".join(sks))
    hs = [xy(suite, bytes.fromhex(v["h"])) for v in vs]
    gam = c.scalar_mul(b"".join(sks), b"".join(hs))
    assert c.points_compress(pks).hex() == "".join(v["pk"] for v in vs)
    assert c.points_compress(gam).hex() == "".join(v["gamma"] for v in vs)
    ios = [[(hs[j], gam[64 * j: 64 * j + 64])] for j in range(len(vs))]
    ads = [bytes.fromhex(v["ad"]) for v in vs]
    proofs = c.thin_prove(Batch.from_items(ios, ads, sks=sks))            # pk derived on the device
    pl = [proofs[96 * j: 96 * j + 96] for j in range(len(vs))]
    for j, v in enumerate(vs):
        assert proof_comp(suite, pl[j], 0).hex() == v["proof_r"] + v["proof_s"]     # src/thin.rs:635-648
    pkl = [pks[64 * j: 64 * j + 64] for j in range(len(vs))]
    assert c.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pkl)) == proofs
    assert c.thin_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=pl)) == [0] * len(vs)
    bad = [bytearray(p) for p in pl]
    bad[0][70] ^= 1; bad[2][:64] = pl[3][:64]
    st = c.thin_verify(Batch.from_items(ios, ads[:4] + [b"q"] + ads[5:], pks_xy=[IDENTITY_XY] + pkl[1:-1] + [pkl[0]],
                                        proofs=[bytes(x) for x in bad]))
    assert st == [2, 0, 1, 0, 1, 0, 1]     # identity pk -> InvalidData (src/thin.rs:140-142); wrong R / ad / pk -> failure


@pytest.mark.parametrize("suite", [0, 1])
def test_multi_io(ctxs, suite):
    """prove_verify_multi / _multi_empty / _multi_single (src/thin.rs:390-470)."""
    from ark_vrf_amd._native import Batch
    c = ctxs[suite]
    sks, pks, ios_c, ads = [], [], [], []
    # 15 / 16 / 17 / 33 pairs: both sides of MSM_THRESHOLD = 16 of merge_ios (src/utils/common.rs:397-412); the oracle takes the
    # two-MSM branch there, the device folds -- same merged point, hence byte-identical proofs
    for j, m in enumerate([0, 1, 2, 3, 5, 15, 16, 17, 33]):
        sk, pk = orc.from_seed(suite, bytes([j + 40]) + bytes(31))
        io = []
        for i in range(m):
            h = orc.hash_to_curve(suite, b"t-%d-%d" % (j, i))
            io.append((h, orc.vrf_output(suite, sk, h)))
        sks.append(sk); pks.append(pk); ios_c.append(io); ads.append(b"ad" * j)
    want = [orc.thin_prove(suite, sk, io, ad) for sk, io, ad in zip(sks, ios_c, ads)]
    ios = [[(xy(suite, i), xy(suite, o)) for i, o in io] for io in ios_c]
    proofs = c.thin_prove(Batch.from_items(ios, ads, sks=sks))
    pl = [proofs[96 * j: 96 * j + 96] for j in range(len(sks))]
    assert [proof_comp(suite, p, 0) for p in pl] == want
    pkl = [xy(suite, p) for p in pks]
    assert c.thin_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=pl)) == [0] * len(sks)
    # tamper an output of the 3-pair item, and an input of the 5-pair item
    ios[3][1] = (ios[3][1][0], ios[3][0][1]); ios[4][4] = (ios[4][3][0], ios[4][4][1])
    bad = c.thin_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=pl))
    assert bad == [0, 0, 0, 1, 1, 0, 0, 0, 0]
    ios[6][15] = (ios[6][15][0], ios[6][14][1]); ios[7][16] = (ios[7][0][0], ios[7][16][1])   # last pair of the 16- and 17-pair items
    assert c.thin_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=pl)) == [0, 0, 0, 1, 1, 0, 1, 1, 0]
    # ONE item per call takes the MSM engine (capi.hip prove_one_as_msm / the unit-weight equation of avrf_thin_verify): the
    # same bytes and verdicts for every pair count, with and without a given public key; 33 pairs = 67 terms: beyond the
    # single-launch MSM of <= 64 terms, the call falls back to the general paths
    ios_ok = [[(xy(suite, i), xy(suite, o)) for i, o in io] for io in ios_c]
    for j in range(len(sks)):
        one = lambda **kw: Batch.from_items([ios_ok[j]], [ads[j]], **kw)
        assert c.thin_prove(one(sks=[sks[j]])) == pl[j] and c.thin_prove(one(sks=[sks[j]], pks_xy=[pkl[j]])) == pl[j]
        assert c.thin_verify(one(pks_xy=[pkl[j]], proofs=[pl[j]])) == [0]
        assert c.thin_verify(Batch.from_items([ios[j]], [ads[j]], pks_xy=[pkl[j]], proofs=[pl[j]])) == [[0, 0, 0, 1, 1, 0, 1, 1, 0][j]]
        assert c.thin_verify(one(pks_xy=[pkl[(j + 1) % len(sks)]], proofs=[pl[j]])) == [1]


def test_points_without_endomorphism_image_take_the_plain_path(ctxs):
    """Bandersnatch runs its variable-base multiplications through the endomorphism (glv.h); psi has no finite image in these
    coordinates for points with x y = 0 -- the identity and the points of order 2 and 4 -- and the kernels must fall back to
    the plain window form for exactly those lanes, inside a batch whose other lanes take the GLV path.
    * k_smul computes k P literally: also for the order-2 point (0, -1) it must equal the oracle.
    * The provers (one-pair form R = k G + (k z) I, scalars reduced mod r) are specified on the prime-order subgroup, like the
      reference's own batch verifier (src/thin.rs:78-94: subgroup membership is the caller's contract); inside the subgroup the
      only point without an image is the identity, which must give the oracle's proof bytes."""
    from ark_vrf_amd._native import Batch
    suite = 0
    c = ctxs[suite]
    q = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    p2 = (q - 1).to_bytes(32, "little")                                   # compressed (0, -1): order 2
    ident = (1).to_bytes(32, "little")
    st, p2_xy = orc.point_decompress(suite, p2)
    assert st == 0 and p2_xy == bytes(32) + p2
    sks, pks, ios_c, ads = [], [], [], []
    for j in range(6):
        sk, pk = orc.from_seed(suite, bytes([j + 90]) + bytes(31))
        h = ident if j in (1, 4) else orc.hash_to_curve(suite, b"so-%d" % j)
        sks.append(sk); pks.append(pk); ios_c.append([(h, orc.vrf_output(suite, sk, h))]); ads.append(b"so%d" % j)
    pts = [p2, ident] + [io[0][0] for io in ios_c]
    ks = [sks[0], sks[1]] + sks
    assert c.scalar_mul(b"".join(ks), b"".join(xy(suite, p) for p in pts)) == b"".join(xy(suite, orc.smul(suite, k, p)) for k, p in zip(ks, pts))
    ios = [[(xy(suite, i), xy(suite, o)) for i, o in io] for io in ios_c]
    pkl = [xy(suite, p) for p in pks]
    want = [orc.thin_prove(suite, sk, io, ad) for sk, io, ad in zip(sks, ios_c, ads)]
    proofs = c.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pkl))
    pl = [proofs[96 * j: 96 * j + 96] for j in range(6)]
    assert [proof_comp(suite, p, 0) for p in pl] == want
    exp = [orc.thin_verify(suite, pk, io, ad, w) for pk, io, ad, w in zip(pks, ios_c, ads, want)]
    assert exp == [0, 2, 0, 0, 2, 0]                                       # identity io: InvalidData before any arithmetic
    assert c.thin_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=pl)) == exp
    pw = [orc.pedersen_prove(suite, sk, io, ad) for sk, io, ad in zip(sks, ios_c, ads)]
    pr, bl = c.pedersen_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pkl))
    pp = [pr[256 * j: 256 * j + 256] for j in range(6)]
    assert [proof_comp(suite, p, 1) for p in pp] == [w[0] for w in pw]
    assert c.pedersen_verify(Batch.from_items(ios, ads, proofs=pp)) == [orc.pedersen_verify(suite, io, ad, w[0]) for io, ad, w in zip(ios_c, ads, pw)]


@pytest.mark.parametrize("suite,n", [(0, 600), (1, 200)])
def test_synthetic(ctxs, suite, n):
    b = orc.gen_batch(suite, 0, n)
    c = ctxs[suite]
    assert c.thin_prove(nat_batch(b, with_sks=True, with_proofs=False)) == b["proofs"]
    assert c.scalar_mul_base(b["sks"]) == b["pks_xy"]
    assert c.thin_verify(nat_batch(b)) == [0] * n
    p2 = bytearray(b["proofs"]); p2[96 * 7 + 65] ^= 1
    b2 = dict(b); b2["proofs"] = bytes(p2)
    st = c.thin_verify(nat_batch(b2))
    assert st[7] == 1 and sum(st) == 1


@pytest.mark.parametrize("suite", [0, 1])
def test_point_codec(ctxs, suite):
    c = ctxs[suite]
    rng = random.Random(5 + suite)
    g = orc.suite_point(suite, 0)
    comps = [orc.smul(suite, rand_scalar(rng, suite), g) for _ in range(50)] + [bytes([1]) + bytes(31)]
    xys, st = c.points_decompress(b"".join(comps))
    assert st == [0] * len(comps)
    for j, cp in enumerate(comps):
        assert xys[64 * j: 64 * j + 64] == orc.point_decompress(suite, cp)[1]
    assert c.points_compress(xys) == b"".join(comps)
    # validate: identity rejected, subgroup points accepted
    _, st = c.points_decompress(b"".join(comps), validate=True)
    assert st == [0] * 50 + [2]
    # garbage: agree with the oracle's decoder on accept / reject, and on the subgroup check
    junk = [bytes(rng.getrandbits(8) for _ in range(32)) for _ in range(64)]
    xs, st = c.points_decompress(b"".join(junk))
    xs2, st2 = c.points_decompress(b"".join(junk), validate=True)
    for j, cp in enumerate(junk):
        o_st, o_xy = orc.point_decompress(suite, cp)
        assert st[j] == o_st
        if o_st == 0:
            assert xs[64 * j: 64 * j + 64] == o_xy
        assert st2[j] == orc.point_decompress(suite, cp, validate=True)[0]


@pytest.mark.parametrize("suite", [0, 1])
def test_validation_levels(ctxs, golden_dir, suite):
    """avrf_ctx_set_validation (SURVEY.md 8b "Input validation"; the reference validates in CanonicalDeserialize / the checked
    constructors, src/lib.rs:410-433,471-494): level 1 rejects off-curve coordinates, level 2 also points of small order;
    per item for the *_verify calls, for the whole batch in BatchVerifier."""
    import json, os
    from ark_vrf_amd._native import Batch
    c = ctxs[suite]
    name = {0: "bandersnatch_sha-512_ell2", 1: "baby-jubjub_sha-512_tai"}[suite]
    vs = json.load(open(os.path.join(golden_dir, name + "_thin.json")))
    pks = [xy(suite, bytes.fromhex(v["pk"])) for v in vs]
    ios = [[(xy(suite, bytes.fromhex(v["h"])), xy(suite, bytes.fromhex(v["gamma"])))] for v in vs]
    ads = [bytes.fromhex(v["ad"]) for v in vs]
    proofs = [proof_xy(suite, bytes.fromhex(v["proof_r"] + v["proof_s"]), 0) for v in vs]
    q = {0: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
         1: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}[suite]
    off_curve = (5).to_bytes(32, "little") + (7).to_bytes(32, "little")
    order2 = bytes(32) + (q - 1).to_bytes(32, "little")                     # (0, -1): on the curve, order 2
    try:
        for level, bad_pt, want in ((1, off_curve, 2), (2, off_curve, 2), (2, order2, 2), (1, order2, None)):
            c.set_validation(level)
            assert c.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=proofs)) == [0] * 7
            assert c.thin_batch_verify(pks, ios, ads, proofs) == 0
            ios_bad = [list(x) for x in ios]; ios_bad[3] = [(ios[3][0][0], bad_pt)]
            st = c.thin_verify(Batch.from_items(ios_bad, ads, pks_xy=pks, proofs=proofs))
            if want is not None:
                assert st[3] == want and st[:3] + st[4:] == [0] * 6
                assert c.thin_batch_verify(pks, ios_bad, ads, proofs) == want
                pks_bad = pks[:5] + [bad_pt] + pks[6:]
                st = c.thin_verify(Batch.from_items(ios, ads, pks_xy=pks_bad, proofs=proofs))
                assert st[5] == want and sum(st) == want
                pr_bad = proofs[:1] + [bad_pt + proofs[1][64:]] + proofs[2:]
                assert c.thin_batch_verify(pks, ios, ads, pr_bad) == want
            else:
                assert st[3] in (1, 2)                                       # level 1 does not look at the subgroup: some failure
    finally:
        c.set_validation(0)


def test_validation_with_mixed_pair_counts(ctxs):
    """avrf_ctx_set_validation with per-item statuses on a batch whose items carry DIFFERENT numbers of I/O pairs (0, 1, 2, 3): the
    pairs are validated in one launch, each lane finding its item in the staged offsets (capi.hip validate_staged) -- a bad
    point must mark exactly its own item, whichever pair it sits in."""
    from ark_vrf_amd._native import Batch
    suite = 0
    c = ctxs[suite]
    counts = (1, 3, 0, 2, 1, 2)
    sks, pks, ios_c, ads = [], [], [], []
    for j, m in enumerate(counts):
        sk, pk = orc.from_seed(suite, bytes([j + 120]) + bytes(31))
        io = []
        for i in range(m):
            h = orc.hash_to_curve(suite, b"vm-%d-%d" % (j, i))
            io.append((h, orc.vrf_output(suite, sk, h)))
        sks.append(sk); pks.append(pk); ios_c.append(io); ads.append(b"vm%d" % j)
    ios = [[(xy(suite, i), xy(suite, o)) for i, o in io] for io in ios_c]
    pkl = [xy(suite, p) for p in pks]
    proofs = [proof_xy(suite, orc.thin_prove(suite, sk, io, ad), 0) for sk, io, ad in zip(sks, ios_c, ads)]
    off_curve = (5).to_bytes(32, "little") + (7).to_bytes(32, "little")
    try:
        c.set_validation(1)
        assert c.thin_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=proofs)) == [0] * 6
        for item, pair, which in ((1, 2, 1), (3, 0, 0), (5, 1, 1), (0, 0, 0)):
            bad = [list(x) for x in ios]
            pr = list(bad[item][pair]); pr[which] = off_curve; bad[item][pair] = tuple(pr)
            st = c.thin_verify(Batch.from_items(bad, ads, pks_xy=pkl, proofs=proofs))
            assert st[item] == 2 and [s for k, s in enumerate(st) if k != item] == [0] * 5, (item, pair, st)
    finally:
        c.set_validation(0)
