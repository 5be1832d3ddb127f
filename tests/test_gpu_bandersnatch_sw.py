"""GPU parity for the fifth suite, Bandersnatch-SW-SHA512-TAI (src/suites/bandersnatch_sw.rs): the Bandersnatch
curve in its short-Weierstrass PRESENTATION.  Every serialised point is ark-serialize's 33-byte SW form; the kernels compute in
the twisted-Edwards model through the maps of src/utils/te_sw_map.rs (sw_map.h), so the xy side of the ABI is the TEMapping of
the suite's SWAffine and the compressed side (avrf_points_*, *_wire, proofs) is exactly what the reference writes.
Pinned against the reference's `bandersnatch_sw_sha-512_tai_{thin,tiny,pedersen,ring}.json` and the oracle (suite id 4)."""
import ctypes as C
import hashlib
import json
import os
import random

import pytest

import oracle as orc
from helpers import rand_scalar

pytestmark = pytest.mark.gpu
S = 4
NAME = "bandersnatch_sw_sha-512_tai"


@pytest.fixture(scope="module")
def ctx():
    from ark_vrf_amd import _native as nat
    c = nat.Context(nat.BANDERSNATCH_SW_SHA512_TAI)
    assert c.point_len == 33
    yield c
    c.close()


def load(golden_dir, k):
    return json.load(open(os.path.join(golden_dir, f"{NAME}_{k}.json")))


def te32(sw33):                     # the oracle's entry points take the 32-byte twisted-Edwards form of a point
    st, t = orc.sw_decode(S, sw33)
    assert st == 0
    return t


def oxy(sw33):                      # xy (TEMapping) of a 33-byte SW point, through the oracle
    st, xy = orc.point_decompress(S, te32(sw33))
    assert st == 0
    return xy


def test_codec_and_hash_to_curve(ctx, golden_dir):
    vs = load(golden_dir, "thin")
    pts = [bytes.fromhex(v[k]) for v in vs for k in ("pk", "h", "gamma", "proof_r")]
    xy, st = ctx.points_decompress(b"".join(pts), validate=True)
    assert st == [0] * len(pts)
    assert [xy[64 * i: 64 * i + 64] for i in range(len(pts))] == [oxy(p) for p in pts]
    assert ctx.points_compress(xy) == b"".join(pts)                        # te_to_sw o sw_to_te = id, flags included
    # undecodable inputs: an x with no point, x >= q, the infinity flag (no twisted-Edwards image), unused flag bits
    rng = random.Random(4)
    q = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    junk = [rng.randrange(q).to_bytes(32, "little") + bytes([rng.choice([0, 0x80])]) for _ in range(64)]
    junk += [(q + 5).to_bytes(32, "little") + b"\x00", bytes(32) + b"\x40", pts[0][:32] + bytes([pts[0][32] | 0x01]), pts[0][:32] + b"\xc0"]
    _, st = ctx.points_decompress(b"".join(junk))
    assert st == [0 if orc.sw_decode(S, j)[0] == 0 else 2 for j in junk] and st[-4:] == [2, 2, 2, 2] and 0 in st and 2 in st[:64]
    # try-and-increment on SW x-coordinates: alpha -> h of the vectors, and the oracle on other messages
    msgs = [bytes.fromhex(v["alpha"]) for v in vs] + [hashlib.sha512(b"sw%d" % i).digest()[: i % 65] for i in range(80)]
    hxy, st = ctx.hash_to_curve(msgs)
    got = ctx.points_compress(hxy)
    assert all(s == 0 for s in st)
    assert [got[33 * i: 33 * i + 33].hex() for i in range(7)] == [v["h"] for v in vs]
    assert all(got[33 * i: 33 * i + 33] == orc.sw_encode(S, orc.hash_to_curve(S, msgs[i])) for i in range(7, len(msgs)))
    # the group law is the twisted-Edwards one of suite 0 (the maps are isomorphisms): MSM against the oracle
    g = orc.suite_point(S, 0)
    base = [orc.point_decompress(S, orc.smul(S, rand_scalar(rng, 0), g))[1] for _ in range(40)]
    for n in (1, 33, 900):
        bases = b"".join(base[i % 40] for i in range(n)); sc = b"".join(rand_scalar(rng, 0) for _ in range(n))
        assert ctx.msm(bases, sc) == orc.msm(S, bases, sc)


def test_thin_tiny_pedersen_vectors(ctx, golden_dir):
    from ark_vrf_amd._native import Batch
    th, ti, pe = load(golden_dir, "thin"), load(golden_dir, "tiny"), load(golden_dir, "pedersen")
    sks = [bytes.fromhex(v["sk"]) for v in th]
    pks = [oxy(bytes.fromhex(v["pk"])) for v in th]
    ios = [[(oxy(bytes.fromhex(v["h"])), oxy(bytes.fromhex(v["gamma"])))] for v in th]
    ads = [bytes.fromhex(v["ad"]) for v in th]
    assert ctx.scalar_mul_base(b"".join(sks)) == b"".join(pks)
    assert ctx.scalar_mul(b"".join(sks), b"".join(i[0][0] for i in ios)) == b"".join(i[0][1] for i in ios)
    got = ctx.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    tp = [got[96 * j: 96 * j + 96] for j in range(7)]
    assert [(ctx.points_compress(p[:64]) + p[64:]).hex() for p in tp] == [v["proof_r"] + v["proof_s"] for v in th]
    assert ctx.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=tp)) == [0] * 7
    assert ctx.thin_batch_verify(pks, ios, ads, tp) == 0
    # the batch verifier's MSM terms against the oracle's (whose entry points take the twisted-Edwards 32-byte form)
    st, bases, sc = orc.thin_batch_terms(S, [te32(bytes.fromhex(v["pk"])) for v in th],
                                         [[(te32(bytes.fromhex(v["h"])), te32(bytes.fromhex(v["gamma"])))] for v in th], ads,
                                         [te32(bytes.fromhex(v["proof_r"])) + bytes.fromhex(v["proof_s"]) for v in th])
    gb, gs = ctx.last_terms()
    assert st == 0 and gs == sc and gb == bases
    bad = tp[:2] + [tp[2][:70] + bytes([tp[2][70] ^ 1]) + tp[2][71:]] + tp[3:]
    assert ctx.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=bad)) == [0, 0, 1, 0, 0, 0, 0]
    assert ctx.thin_batch_verify(pks, ios, ads, bad) == 1
    got = ctx.tiny_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    yp = [got[48 * j: 48 * j + 48] for j in range(7)]
    assert [p.hex() for p in yp] == [v["proof_c"] + v["proof_s"] for v in ti]
    assert ctx.tiny_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=yp)) == [0] * 7
    pr, bl = ctx.pedersen_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
    pp = [pr[256 * j: 256 * j + 256] for j in range(7)]
    assert [(ctx.points_compress(p[:192]) + p[192:]).hex() for p in pp] == [v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] for v in pe]
    assert [bl[32 * j: 32 * j + 32].hex() for j in range(7)] == [v["blinding"] for v in pe]
    assert ctx.pedersen_verify(Batch.from_items(ios, ads, proofs=pp)) == [0] * 7
    assert ctx.pedersen_batch_verify(ios, ads, pp) == 0
    # several I/O pairs per item (merge path) against the oracle's provers
    sks2, pks2, ios_c, ads2 = [], [], [], []
    for j, m in enumerate([0, 2, 3, 17]):
        sk, pk = orc.from_seed(S, bytes([j + 60]) + bytes(31))
        io = []
        for i in range(m):
            h = orc.hash_to_curve(S, b"swm-%d-%d" % (j, i)); io.append((h, orc.vrf_output(S, sk, h)))
        sks2.append(sk); pks2.append(pk); ios_c.append(io); ads2.append(b"m" * j)
    dxy = lambda t: orc.point_decompress(S, t)[1]
    ios2 = [[(dxy(i), dxy(o)) for i, o in io] for io in ios_c]
    pk2 = [dxy(p) for p in pks2]
    got = ctx.thin_prove(Batch.from_items(ios2, ads2, sks=sks2, pks_xy=pk2))
    want = [orc.thin_prove(S, sk, io, ad) for sk, io, ad in zip(sks2, ios_c, ads2)]
    assert [ctx.points_compress(got[96 * j: 96 * j + 64]) + got[96 * j + 64: 96 * j + 96] for j in range(4)] == [orc.sw_encode(S, w[:32]) + w[32:] for w in want]
    assert ctx.thin_verify(Batch.from_items(ios2, ads2, pks_xy=pk2, proofs=[got[96 * j: 96 * j + 96] for j in range(4)])) == [0] * 4


@pytest.mark.parametrize("validate", [0, 1])
def test_wire_flavour(ctx, golden_dir, validate):
    """the reference's byte strings as they are: 33-byte points, 65 / 48 / 163-byte proofs"""
    from test_gpu_wire import _call
    th, ti, pe = load(golden_dir, "thin"), load(golden_dir, "tiny"), load(golden_dir, "pedersen")
    pks = [bytes.fromhex(v["pk"]) for v in th]
    ios = [[(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))] for v in th]
    ads = [bytes.fromhex(v["ad"]) for v in th]
    one = [1] * 7
    tp = [bytes.fromhex(v["proof_r"] + v["proof_s"]) for v in th]
    assert all(len(p) == 65 for p in tp)
    assert _call("avrf_thin_batch_verify_wire", ctx, 7, pks, ios, one, ads, tp, validate, False)[0] == 0
    assert _call("avrf_thin_verify_wire", ctx, 7, pks, ios, one, ads, tp, validate, True) == (0, [0] * 7)
    yp = [bytes.fromhex(v["proof_c"] + v["proof_s"]) for v in ti]
    assert _call("avrf_tiny_verify_wire", ctx, 7, pks, ios, one, ads, yp, validate, True) == (0, [0] * 7)
    pp = [bytes.fromhex(v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]) for v in pe]
    assert all(len(p) == 163 for p in pp)
    assert _call("avrf_pedersen_batch_verify_wire", ctx, 7, None, ios, one, ads, pp, validate, False)[0] == 0
    assert _call("avrf_pedersen_verify_wire", ctx, 7, None, ios, one, ads, pp, validate, True) == (0, [0] * 7)
    bad_x = next(k.to_bytes(32, "little") + b"\x00" for k in range(2, 300) if orc.sw_decode(S, k.to_bytes(32, "little") + b"\x00")[0] != 0)
    assert _call("avrf_thin_verify_wire", ctx, 7, pks[:2] + [bad_x] + pks[3:], ios, one, ads, tp, validate, True) == (0, [0, 0, 2, 0, 0, 0, 0])
    tp2 = tp[:4] + [tp[4][:40] + bytes([tp[4][40] ^ 1]) + tp[4][41:]] + tp[5:]
    assert _call("avrf_thin_verify_wire", ctx, 7, pks, ios, one, ads, tp2, validate, True) == (0, [0, 0, 0, 0, 1, 0, 0])
    pp2 = pp[:1] + [pp[1][:33] + pp[2][33:66] + pp[1][66:]] + pp[2:]          # another (valid) point as R
    assert _call("avrf_pedersen_verify_wire", ctx, 7, None, ios, one, ads, pp2, validate, True) == (0, [0, 1, 0, 0, 0, 0, 0])


def test_ring_vectors(ctx, golden_dir):
    """ring_proof::index, RingProver::prove, ring::Proof bytes (755 = 163 + 592) and the verifiers on the reference's
    Bandersnatch-SW ring vectors: the ring runs on the TEMapping of the 33-byte keys (src/ring.rs:75-81)."""
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import RingSetup, ring_batch_verify, ring_verify_each
    vs = load(golden_dir, "ring")
    srs = open(os.path.join(golden_dir, "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()
    setup = RingSetup(ctx, srs, 8)
    assert setup.domain_size == 512 and (setup.proof_len, setup.commitment_len) == (592, 144)
    L = nat.lib()
    coms, insts, proofs, full, ios_w, ads = [], [], [], [], [], []
    for v in vs:
        raw = bytes.fromhex(v["ring_pks"])
        keys = [raw[33 * i: 33 * i + 33] for i in range(len(raw) // 33)]
        xy, st = ctx.points_decompress(raw)
        assert st == [0] * len(keys)
        key = setup.index([xy[64 * i: 64 * i + 64] for i in range(len(keys))])
        assert key.commitment.hex() == v["ring_pks_com"]
        idx = [k.hex() for k in keys].index(v["pk"])
        proof = key.prove([idx], [bytes.fromhex(v["blinding"])])[0]
        assert proof.hex() == v["ring_proof"]
        out = (C.c_uint8 * (163 + 592))()
        io_xy = oxy(bytes.fromhex(v["h"])) + oxy(bytes.fromhex(v["gamma"]))
        ad = bytes.fromhex(v["ad"])
        assert L.avrf_ring_vrf_prove(ctx._h, key._h, C.c_size_t(592), C.c_size_t(1), nat._u8(bytes.fromhex(v["sk"])), nat._u32([idx]),
                                     nat._u8(io_xy), nat._u32([1]), nat._u8(ad), nat._u32([len(ad)]), 0, out) == 0
        want = bytes.fromhex(v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] + v["ring_proof"])
        assert bytes(out) == want
        coms.append(key.commitment); insts.append(oxy(bytes.fromhex(v["proof_pk_com"]))); proofs.append(proof); full.append(want)
        ios_w.append(bytes.fromhex(v["h"]) + bytes.fromhex(v["gamma"])); ads.append(ad)
        key.close()
    n = len(vs)
    assert ring_batch_verify(setup, coms, list(range(n)), insts, proofs) == 0
    assert ring_verify_each(setup, coms, list(range(n)), insts, proofs) == [0] * n
    bad = list(proofs); bad[1] = bad[1][:300] + bytes([bad[1][300] ^ 1]) + bad[1][301:]
    st = ring_verify_each(setup, coms, list(range(n)), insts, bad)
    assert st[1] in (1, 2) and st[:1] + st[2:] == [0] * (n - 1)

    def verify(prs, each):
        out = (C.c_int32 * n)()
        rc = L.avrf_ring_vrf_verify(ctx._h, setup._h, C.c_size_t(n), nat._u8(b"".join(coms)), C.c_size_t(n), nat._u32(list(range(n))), nat._u8(b"".join(ios_w)),
                                    nat._u32([1] * n), nat._u8(b"".join(ads)), nat._u32([len(a) for a in ads]), nat._u8(b"".join(prs)), 1, int(each), out)
        return rc, list(out)
    assert verify(full, True) == (0, [0] * n)
    assert verify(full, False)[0] == 0
    tam = list(full); tam[3] = tam[3][:110] + bytes([tam[3][110] ^ 1]) + tam[3][111:]               # Pedersen response
    rc, st = verify(tam, True)
    assert rc == 0 and st[3] == 1 and st[:3] + st[4:] == [0] * (n - 1)
    setup.close()
