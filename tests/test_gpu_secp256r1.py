"""GPU parity for Secp256r1-SHA256-TAI (src/suites/secp256r1.rs:49-70), suite id 7: NIST P-256 -- a genuinely
short-Weierstrass curve (a = -3, cofactor 1) over 256-bit fields whose top bit is set, with HashTranscript<Sha256> and 33-byte
points.  The kernels are the other suites' kernels (te.h gives the point types their XYZZ meaning for this suite); everything
is held against the reference's `secp256r1_sha-256_tai_{tiny,thin,pedersen}.json` vectors and the oracle.  In the xy flavour
of the C ABI the identity of this group is the all-zero 64 bytes.  Tiny / Thin / Pedersen only: the reference has no RingSuite
for the curve."""
import ctypes as C
import hashlib
import json
import os
import random

import pytest

import oracle as orc
from helpers import nat_batch, rand_points_xy, rand_scalar, xy

pytestmark = pytest.mark.gpu
S = 7
PL = 33
NAME = "secp256r1_sha-256_tai"
IDENT = bytes(64)


@pytest.fixture(scope="module")
def ctx():
    from ark_vrf_amd import _native as nat
    c = nat.Context(nat.SECP256R1_SHA256_TAI)
    assert c.point_len == PL
    yield c
    c.close()


def load(golden_dir, k):
    return json.load(open(os.path.join(golden_dir, f"{NAME}_{k}.json")))


def comp(pxy):
    return orc.point_compress(S, pxy)


def thin_comp(p):          # ABI proof (R_xy || s) -> the reference's bytes (R 33 || s 32)
    return comp(p[:64]) + p[64:]


def ped_comp(p):
    return b"".join(comp(p[64 * k: 64 * k + 64]) for k in range(3)) + p[192:]


def test_field_and_group_arithmetic(ctx):
    """the 256-bit fields with the top bit set (sums and Montgomery products carry into bit 256) and the XYZZ group law with
    all its exceptional cases, through avrf_scalar_mul / avrf_msm_te against the oracle: random scalars, 0, 1, r - 1, scalars
    with the top bit set, P and -P in one MSM, repeated points, the point at infinity as a base"""
    rng = random.Random(70)
    r = 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551
    pts = rand_points_xy(rng, S, 40)
    ks = [rand_scalar(rng, S) for _ in range(36)] + [(0).to_bytes(32, "little"), (1).to_bytes(32, "little"), (r - 1).to_bytes(32, "little"),
                                                      ((1 << 255) + 12345).to_bytes(32, "little")]
    want = b"".join(orc.point_decompress(S, orc.smul(S, k, comp(p)))[1] for k, p in zip(ks, pts))
    assert ctx.scalar_mul(b"".join(ks), b"".join(pts)) == want
    g = xy(S, orc.suite_point(S, 0))
    assert ctx.scalar_mul_base(b"".join(ks)) == b"".join(orc.point_decompress(S, orc.smul(S, k, comp(g)))[1] for k in ks)
    for n in (1, 29, 700, 4097):
        bases = b"".join(pts[i % 40] for i in range(n))
        sc = b"".join(rand_scalar(rng, S) for _ in range(n))
        assert ctx.msm(bases, sc) == orc.msm(S, bases, sc)
    q = 0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff
    neg = lambda p: p[:32] + ((q - int.from_bytes(p[32:], "little")) % q).to_bytes(32, "little")
    k = rand_scalar(rng, S)
    assert ctx.msm(pts[0] + neg(pts[0]), k + k) == IDENT                                        # P and -P: the sum is the identity
    assert ctx.msm(pts[1] * 3 + IDENT, ks[0] + ks[1] + ks[2] + ks[3]) == orc.msm(S, pts[1] * 3 + IDENT, ks[0] + ks[1] + ks[2] + ks[3])
    one = (1).to_bytes(32, "little")
    assert ctx.msm(pts[2] * 2, one + one) == orc.point_decompress(S, orc.smul(S, (2).to_bytes(32, "little"), comp(pts[2])))[1]   # P + P


def test_codec_and_hash_to_curve(ctx, golden_dir):
    vs = load(golden_dir, "thin")
    msgs = [bytes.fromhex(v["alpha"]) for v in vs] + [hashlib.sha512(b"p256-%d" % i).digest()[: i % 65] for i in range(100)]
    xy_, st = ctx.hash_to_curve(msgs)
    got = ctx.points_compress(xy_)
    assert all(s == 0 for s in st) and len(got) == PL * len(msgs)
    assert [got[PL * i: PL * i + PL].hex() for i in range(7)] == [v["h"] for v in vs]                    # try-and-increment, alpha -> h
    assert all(got[PL * i: PL * i + PL] == orc.hash_to_curve(S, msgs[i]) for i in range(7, len(msgs)))
    raw = b"".join(bytes.fromhex(v["pk"]) + bytes.fromhex(v["gamma"]) for v in vs)
    pxy, st = ctx.points_decompress(raw, validate=True)
    assert st == [0] * 14 and ctx.points_compress(pxy) == raw
    assert pxy == b"".join(orc.point_decompress(S, raw[PL * i: PL * i + PL])[1] for i in range(14))
    bad = bytes.fromhex(vs[0]["pk"])
    _, st = ctx.points_decompress(bad[:32] + b"\x01" + (5).to_bytes(32, "little") + b"\x00" + b"\xff" * 32 + b"\x00", validate=False)
    assert st[0] == 2 and st[2] == 2 and st[1] == orc.point_decompress(S, (5).to_bytes(32, "little") + b"\x00")[0]     # unused flag bit; x >= p


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
    assert [thin_comp(p).hex() for p in tp] == [v["proof_r"] + v["proof_s"] for v in th]
    assert ctx.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=tp)) == [0] * 7
    assert ctx.thin_batch_verify(pks, ios, ads, tp) == 0
    st, bases, sc = orc.thin_batch_terms(S, [bytes.fromhex(v["pk"]) for v in th], [[(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))] for v in th], ads,
                                         [bytes.fromhex(v["proof_r"] + v["proof_s"]) for v in th])
    gb, gs = ctx.last_terms()
    assert st == 0 and gs == sc and gb == bases
    bad = tp[:2] + [tp[2][:70] + bytes([tp[2][70] ^ 1]) + tp[2][71:]] + tp[3:]
    assert ctx.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=bad)) == [0, 0, 1, 0, 0, 0, 0]
    assert ctx.thin_batch_verify(pks, ios, ads, bad) == 1
    assert ctx.thin_batch_verify([IDENT] + pks[1:], ios, ads, tp) == 2                                   # identity key: InvalidData
    assert ctx.thin_verify(Batch.from_items(ios, ads, pks_xy=[IDENT] + pks[1:], proofs=tp))[0] == 2
    got = ctx.tiny_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    yp = [got[48 * j: 48 * j + 48] for j in range(7)]
    assert [p.hex() for p in yp] == [v["proof_c"] + v["proof_s"] for v in ti]
    assert ctx.tiny_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=yp)) == [0] * 7
    yb = yp[:4] + [yp[4][:20] + bytes([yp[4][20] ^ 1]) + yp[4][21:]] + yp[5:]
    assert ctx.tiny_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=yb)) == [0, 0, 0, 0, 1, 0, 0]
    pr, bl = ctx.pedersen_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    pp = [pr[256 * j: 256 * j + 256] for j in range(7)]
    assert [ped_comp(p).hex() for p in pp] == [v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] for v in pe]
    assert [bl[32 * j: 32 * j + 32].hex() for j in range(7)] == [v["blinding"] for v in pe]
    assert ctx.pedersen_verify(Batch.from_items(ios, ads, proofs=pp)) == [0] * 7
    assert ctx.pedersen_batch_verify(ios, ads, pp) == 0
    st, bases, sc = orc.pedersen_batch_terms(S, [[(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))] for v in pe], ads, [ped_comp(p) for p in pp])
    gb, gs = ctx.last_terms()
    assert st == 0 and gs == sc and gb == bases
    pb = pp[:5] + [pp[5][:200] + bytes([pp[5][200] ^ 1]) + pp[5][201:]] + pp[6:]
    assert ctx.pedersen_verify(Batch.from_items(ios, ads, proofs=pb)) == [0, 0, 0, 0, 0, 1, 0]
    assert ctx.pedersen_batch_verify(ios, ads, pb) == 1


def test_wire_flavour(ctx, golden_dir):
    """the reference's own byte strings: 33-byte points, 65-byte thin and 163-byte Pedersen proofs, Validate::Yes / No"""
    from ark_vrf_amd import _native as nat
    L = nat.lib()
    th, pe = load(golden_dir, "thin"), load(golden_dir, "pedersen")
    pks = b"".join(bytes.fromhex(v["pk"]) for v in th)
    ios = b"".join(bytes.fromhex(v["h"]) + bytes.fromhex(v["gamma"]) for v in th)
    ads = [bytes.fromhex(v["ad"]) for v in th]
    tp = b"".join(bytes.fromhex(v["proof_r"] + v["proof_s"]) for v in th)
    pp = b"".join(bytes.fromhex(v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]) for v in pe)
    assert len(tp) == 7 * 65 and len(pp) == 7 * 163
    one, adl = nat._u32([1] * 7), nat._u32([len(a) for a in ads])
    for validate in (0, 1):
        out = (C.c_int32 * 7)()
        assert L.avrf_thin_verify_wire(ctx._h, C.c_size_t(7), nat._u8(pks), nat._u8(ios), one, nat._u8(b"".join(ads)), adl, nat._u8(tp), validate, out) == 0
        assert list(out) == [0] * 7
        assert L.avrf_thin_batch_verify_wire(ctx._h, C.c_size_t(7), nat._u8(pks), nat._u8(ios), one, nat._u8(b"".join(ads)), adl, nat._u8(tp), validate) == 0
        assert L.avrf_pedersen_verify_wire(ctx._h, C.c_size_t(7), nat._u8(ios), one, nat._u8(b"".join(ads)), adl, nat._u8(pp), validate, out) == 0
        assert list(out) == [0] * 7
        assert L.avrf_pedersen_batch_verify_wire(ctx._h, C.c_size_t(7), nat._u8(ios), one, nat._u8(b"".join(ads)), adl, nat._u8(pp), validate) == 0
    bad = bytearray(tp); bad[65 * 3 + 40] ^= 1
    out = (C.c_int32 * 7)()
    assert L.avrf_thin_verify_wire(ctx._h, C.c_size_t(7), nat._u8(pks), nat._u8(ios), one, nat._u8(b"".join(ads)), adl, nat._u8(bytes(bad)), 1, out) == 0
    assert list(out) == [0, 0, 0, 1, 0, 0, 0]


@pytest.mark.parametrize("kind,n", [(0, 900), (1, 400)])
def test_synthetic_batches_vs_oracle(ctx, kind, n):
    b = orc.gen_batch(S, kind, n)
    if kind == 0:
        assert ctx.thin_prove(nat_batch(b, with_sks=True, with_proofs=False)) == b["proofs"]
        assert ctx.thin_verify(nat_batch(b)) == [0] * n
        assert ctx.thin_batch_stage(nat_batch(b)) == 0 and ctx.thin_batch_run() == 0
        st, bases, sc = orc.thin_batch_terms_xy(S, b)
    else:
        pr, _ = ctx.pedersen_prove(nat_batch(b, with_sks=True, with_proofs=False))
        assert pr == b["proofs"]
        b["pks_xy"] = b""
        assert ctx.pedersen_verify(nat_batch(b)) == [0] * n
        assert ctx.pedersen_batch_stage(nat_batch(b)) == 0 and ctx.pedersen_batch_run() == 0
        st, bases, sc = orc.pedersen_batch_terms_xy(S, b)
    gb, gs = ctx.last_terms()
    assert st == 0 and gs == sc and gb == bases
    psz = 96 if kind == 0 else 256
    p2 = bytearray(b["proofs"]); p2[psz * (n // 3) + (64 if kind == 0 else 200)] ^= 1
    b2 = dict(b); b2["proofs"] = bytes(p2)
    if kind == 0:
        assert ctx.thin_batch_stage(nat_batch(b2)) == 0 and ctx.thin_batch_run() == 1
    else:
        assert ctx.pedersen_batch_stage(nat_batch(b2)) == 0 and ctx.pedersen_batch_run() == 1


def test_items_with_several_pairs(ctx):
    """merge_ios (src/utils/common.rs:389-419) on this group: 0, 2, 5 and 17 pairs per item (the >= 16-pair MSM branch in the
    oracle), provers byte-exact, verifiers accept, batch terms bit-exact"""
    from ark_vrf_amd._native import Batch
    rng = random.Random(71)
    sks, pks, ios_c, ads = [], [], [], []
    for j, m in enumerate((0, 2, 5, 17, 1)):
        sk, pk = orc.from_seed(S, bytes([j + 40]) + bytes(31))
        io = []
        for i in range(m):
            h = orc.hash_to_curve(S, b"mp-%d-%d" % (j, i))
            io.append((h, orc.vrf_output(S, sk, h)))
        sks.append(sk); pks.append(pk); ios_c.append(io); ads.append(b"mp%d" % j)
    ios = [[(xy(S, i), xy(S, o)) for i, o in io] for io in ios_c]
    pkl = [xy(S, p) for p in pks]
    want = [orc.thin_prove(S, sk, io, ad) for sk, io, ad in zip(sks, ios_c, ads)]
    got = ctx.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pkl))
    tp = [got[96 * j: 96 * j + 96] for j in range(5)]
    assert [thin_comp(p) for p in tp] == want
    assert ctx.thin_verify(Batch.from_items(ios, ads, pks_xy=pkl, proofs=tp)) == [0] * 5
    assert ctx.thin_batch_verify(pkl, ios, ads, tp) == 0
    st, bases, sc = orc.thin_batch_terms(S, pks, ios_c, ads, want)
    gb, gs = ctx.last_terms()
    assert st == 0 and gs == sc and gb == bases
    wantp = [orc.pedersen_prove(S, sk, io, ad) for sk, io, ad in zip(sks, ios_c, ads)]
    pr, bl = ctx.pedersen_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pkl))
    pp = [pr[256 * j: 256 * j + 256] for j in range(5)]
    assert [ped_comp(p) for p in pp] == [w[0] for w in wantp] and [bl[32 * j: 32 * j + 32] for j in range(5)] == [w[1] for w in wantp]
    assert ctx.pedersen_verify(Batch.from_items(ios, ads, proofs=pp)) == [0] * 5
    assert ctx.pedersen_batch_verify(ios, ads, pp) == 0
    st, bases, sc = orc.pedersen_batch_terms(S, ios_c, ads, [w[0] for w in wantp])
    gb, gs = ctx.last_terms()
    assert st == 0 and gs == sc and gb == bases


def test_not_a_ring_suite(ctx, golden_dir):
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import RingSetup
    srs = open(os.path.join(golden_dir, "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()
    with pytest.raises(nat.AvrfError, match="-> -2"):
        RingSetup(ctx, srs, 8)
