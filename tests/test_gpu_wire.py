"""GPU: the wire-format flavour of the C ABI (include/avrf.h "Wire-format flavour") -- the reference's vectors fed
in exactly the bytes `CanonicalSerialize` produces (32-byte points, 64 / 48 / 160 / 752-byte proofs), with the validate flag."""
import ctypes as C
import json
import os

import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
NAMES = {0: "bandersnatch_sha-512_ell2", 1: "baby-jubjub_sha-512_tai"}


@pytest.fixture(scope="module")
def ctxs():
    from ark_vrf_amd import _native as nat
    return {s: nat.Context(s) for s in (0, 1)}


def _call(fn, ctx, n, pks, ios, counts, ads, proofs, validate, per_item):
    from ark_vrf_amd import _native as nat
    args = [ctx._h, C.c_size_t(n)]
    if pks is not None:
        args.append(nat._u8(b"".join(pks)))
    args += [nat._u8(b"".join(i + o for it in ios for i, o in it)), nat._u32(counts), nat._u8(b"".join(ads)), nat._u32([len(a) for a in ads]),
             nat._u8(b"".join(proofs)), int(validate)]
    if per_item:
        out = (C.c_int32 * max(1, n))()
        rc = getattr(nat.lib(), fn)(*args, out)
        return rc, list(out)[:n]
    return getattr(nat.lib(), fn)(*args), None


@pytest.mark.parametrize("suite", [0, 1])
@pytest.mark.parametrize("validate", [0, 1])
def test_thin_tiny_pedersen_vectors_wire(ctxs, golden_dir, suite, validate):
    c = ctxs[suite]
    load = lambda k: json.load(open(os.path.join(golden_dir, f"{NAMES[suite]}_{k}.json")))
    th, ti, pe = load("thin"), load("tiny"), load("pedersen")
    pks = [bytes.fromhex(v["pk"]) for v in th]
    ios = [[(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))] for v in th]
    ads = [bytes.fromhex(v["ad"]) for v in th]
    one = [1] * 7
    tp = [bytes.fromhex(v["proof_r"] + v["proof_s"]) for v in th]
    assert _call("avrf_thin_batch_verify_wire", c, 7, pks, ios, one, ads, tp, validate, False)[0] == 0
    assert _call("avrf_thin_verify_wire", c, 7, pks, ios, one, ads, tp, validate, True) == (0, [0] * 7)
    yp = [bytes.fromhex(v["proof_c"] + v["proof_s"]) for v in ti]
    assert _call("avrf_tiny_verify_wire", c, 7, pks, ios, one, ads, yp, validate, True) == (0, [0] * 7)
    pp = [bytes.fromhex(v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]) for v in pe]
    assert all(len(p) == 160 for p in pp)
    assert _call("avrf_pedersen_batch_verify_wire", c, 7, None, ios, one, ads, pp, validate, False)[0] == 0
    assert _call("avrf_pedersen_verify_wire", c, 7, None, ios, one, ads, pp, validate, True) == (0, [0] * 7)
    # a point that does not decode: InvalidData for its item / for the batch
    bad_y = None
    for k in range(2, 300):                                                     # a y with no x on the curve
        cand = k.to_bytes(32, "little")
        if orc.point_decompress(suite, cand)[0] != 0:
            bad_y = cand; break
    pks_bad = pks[:2] + [bad_y] + pks[3:]
    assert _call("avrf_thin_verify_wire", c, 7, pks_bad, ios, one, ads, tp, validate, True) == (0, [0, 0, 2, 0, 0, 0, 0])
    assert _call("avrf_thin_batch_verify_wire", c, 7, pks_bad, ios, one, ads, tp, validate, False)[0] == 2
    ios_bad = [list(x) for x in ios]; ios_bad[5] = [(ios[5][0][0], bad_y)]
    assert _call("avrf_pedersen_verify_wire", c, 7, None, ios_bad, one, ads, pp, validate, True) == (0, [0, 0, 0, 0, 0, 2, 0])
    assert _call("avrf_tiny_verify_wire", c, 7, pks, ios_bad, one, ads, yp, validate, True) == (0, [0, 0, 0, 0, 0, 2, 0])
    pp_bad = [bad_y + pp[0][32:]] + pp[1:]
    assert _call("avrf_pedersen_batch_verify_wire", c, 7, None, ios, one, ads, pp_bad, validate, False)[0] == 2
    # tampered scalar: VerificationFailure
    tp2 = tp[:4] + [tp[4][:40] + bytes([tp[4][40] ^ 1]) + tp[4][41:]] + tp[5:]
    assert _call("avrf_thin_verify_wire", c, 7, pks, ios, one, ads, tp2, validate, True) == (0, [0, 0, 0, 0, 1, 0, 0])
    # Validate::Yes rejects a point of small order (on the curve, outside the prime-order subgroup); Validate::No lets it decode
    q = {0: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
         1: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}[suite]
    order2 = (q - 1).to_bytes(32, "little")                                     # (0, -1)
    st = _call("avrf_thin_verify_wire", c, 7, pks[:6] + [order2], ios, one, ads, tp, validate, True)[1]
    assert st[:6] == [0] * 6 and st[6] == (2 if validate else 1)


@pytest.mark.parametrize("suite", [0, 1])
def test_torsion_points(ctxs, golden_dir, suite):
    """Points with a torsion component, P' = P + (0, -1) = (-x, -y): on the curve, outside the prime-order subgroup
    (include/avrf.h "PRIME-ORDER-SUBGROUP MEMBERSHIP IS A HARD PRECONDITION").
    * avrf_scalar_mul is the literal product k P' for any curve point (Bandersnatch: NOT the GLV split, which is k P' only
      inside the subgroup) -- equal to the oracle's double-and-add;
    * Validate::Yes answers InvalidData for exactly the item that carries such a point, before any equation;
    * Validate::No: that item's verdict is unspecified (it is some status), every other item is unaffected."""
    c = ctxs[suite]
    q = {0: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
         1: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}[suite]
    th = json.load(open(os.path.join(golden_dir, f"{NAMES[suite]}_thin.json")))
    pks = [bytes.fromhex(v["pk"]) for v in th]
    ios = [[(bytes.fromhex(v["h"]), bytes.fromhex(v["gamma"]))] for v in th]
    ads = [bytes.fromhex(v["ad"]) for v in th]
    tp = [bytes.fromhex(v["proof_r"] + v["proof_s"]) for v in th]

    def tweak(comp):
        st, pxy = orc.point_decompress(suite, comp)
        assert st == 0
        x, y = int.from_bytes(pxy[:32], "little"), int.from_bytes(pxy[32:], "little")
        txy = ((q - x) % q).to_bytes(32, "little") + ((q - y) % q).to_bytes(32, "little")
        return orc.point_compress(suite, txy), txy
    tw = [tweak(p) for p in pks]
    ks = [bytes.fromhex(v["sk"]) for v in th]
    got = c.scalar_mul(b"".join(ks), b"".join(t[1] for t in tw))
    want = b"".join(orc.point_decompress(suite, orc.smul(suite, k, t[0]))[1] for k, t in zip(ks, tw))
    assert got == want
    one = [1] * 7
    pks_t = pks[:4] + [tw[4][0]] + pks[5:]
    assert _call("avrf_thin_verify_wire", c, 7, pks_t, ios, one, ads, tp, 1, True) == (0, [0, 0, 0, 0, 2, 0, 0])
    assert _call("avrf_thin_batch_verify_wire", c, 7, pks_t, ios, one, ads, tp, 1, False)[0] == 2
    rc, st = _call("avrf_thin_verify_wire", c, 7, pks_t, ios, one, ads, tp, 0, True)
    assert rc == 0 and st[:4] + st[5:] == [0] * 6 and st[4] in (0, 1, 2)
    ios_t = [list(x) for x in ios]; ios_t[2] = [(ios[2][0][0], tweak(ios[2][0][1])[0])]
    assert _call("avrf_thin_verify_wire", c, 7, pks, ios_t, one, ads, tp, 1, True) == (0, [0, 0, 2, 0, 0, 0, 0])


def test_subgroup_check_all_cosets(ctxs):
    """Validate::Yes on random AFFINE Bandersnatch points: a curve point is S + T with S of order r and T one of the four 2-torsion
    points of this model -- the identity, (0, -1), and two points at infinity (a = -5 is not a square) -- so only a quarter of
    the curve's affine points pass.  The few-points kernel tests  a2 P + b2 psi(P) == 0  with the endomorphism (two 127-bit halves
    instead of r P); the lane-per-item kernels compute r P.  Both must agree with plain affine arithmetic on every coset (the OTHER
    basis vector of the lattice would accept the coset of (0, -1): vrf_single.hip k_decompress_wave)."""
    import random
    c = ctxs[0]
    q = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    r = 0x1cfb69d4ca675f520cce760202687600ff8f87007419047174fd06b52876e7e1
    d = 0x6389c12633c267cbc66e3bf86be3b6d8cb66677177e54f92b369f2f5188d58e7
    a = q - 5

    def add(p1, p2):                                                    # affine law; ValueError = the sum is a point at infinity
        x1, y1 = p1; x2, y2 = p2
        t = d * x1 * x2 * y1 * y2 % q
        return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, q) % q, (y1 * y2 - a * x1 * x2) * pow(1 - t, -1, q) % q)

    def coset(p):                                                       # r P: (0, 1) in the subgroup, (0, -1), or at infinity
        acc = (0, 1)
        try:
            for bit in bin(r)[2:]:
                acc = add(acc, acc)
                if bit == "1":
                    acc = add(acc, p)
        except ValueError:
            return "inf"
        return {(0, 1): "sub", (0, q - 1): "neg"}[acc]
    rng = random.Random(11)
    comp, kinds = [], []
    while len(comp) < 40:
        y = rng.randrange(q)
        st, pxy = orc.point_decompress(0, y.to_bytes(32, "little"))
        if st != 0:
            continue
        x = int.from_bytes(pxy[:32], "little")
        assert (a * x * x + y * y - 1 - d * x * x * y * y) % q == 0
        comp.append(y.to_bytes(32, "little")); kinds.append(coset((x, y)))
    assert {"sub", "neg", "inf"} <= set(kinds)                          # all three kinds occur among 40 random points
    want = [0 if k == "sub" else 2 for k in kinds]
    xy_out, st = c.points_decompress(b"".join(comp), validate=True)     # the few-points kernel (<= 4096 points)
    assert st == want
    xy_all, st_all = c.points_decompress(b"".join(comp) * 110, validate=True)   # 4400 points: the lane-per-item kernel
    assert st_all == want * 110 and xy_all[: 64 * 40] == xy_out
    assert c.points_decompress(b"".join(comp), validate=False)[1] == [0] * 40


@pytest.mark.parametrize("suite", [0, 1])
def test_ring_vrf_one_call(ctxs, golden_dir, suite):
    """ring::Prover::prove / ring::Verifier::verify / ring::BatchVerifier as single calls on the reference's ring vectors:
    with blinding disabled the 752 / 640-byte proof equals `proof_pk_com || proof_r || proof_ok || proof_s || proof_sb || ring_proof`."""
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import RingSetup
    from helpers import xy
    c = ctxs[suite]
    vs = json.load(open(os.path.join(golden_dir, NAMES[suite] + "_ring.json")))
    srs = open(os.path.join(golden_dir, ["bls12-381-srs-2-11-uncompressed-zcash.bin", "bn254-testing-2-9-uncompressed.bin"][suite]), "rb").read()
    setup = RingSetup(c, srs, 8)
    rlen = setup.proof_len
    L = nat.lib()
    proofs, coms, ios_w, ads = [], [], [], []
    for v in vs:
        raw = bytes.fromhex(v["ring_pks"])
        pks = [xy(suite, raw[32 * i: 32 * i + 32]) for i in range(len(raw) // 32)]
        key = setup.index(pks)
        idx = [raw[32 * i: 32 * i + 32].hex() for i in range(len(pks))].index(v["pk"])
        io_xy = xy(suite, bytes.fromhex(v["h"])) + xy(suite, bytes.fromhex(v["gamma"]))
        ad = bytes.fromhex(v["ad"])
        out = (C.c_uint8 * (160 + rlen))()
        rc = L.avrf_ring_vrf_prove(c._h, key._h, C.c_size_t(rlen), C.c_size_t(1), nat._u8(bytes.fromhex(v["sk"])), nat._u32([idx]), nat._u8(io_xy),
                                   nat._u32([1]), nat._u8(ad), nat._u32([len(ad)]), 0, out)
        assert rc == 0
        want = bytes.fromhex(v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] + v["ring_proof"])
        assert bytes(out) == want
        # the length argument is checked against the key's setup, a context of another suite is refused (nothing is written)
        for bad_len in (rlen - 1, rlen + 16, 0):
            assert L.avrf_ring_vrf_prove(c._h, key._h, C.c_size_t(bad_len), C.c_size_t(1), nat._u8(bytes.fromhex(v["sk"])), nat._u32([idx]), nat._u8(io_xy),
                                         nat._u32([1]), nat._u8(ad), nat._u32([len(ad)]), 0, out) == nat.ERR_BAD_ARG
        assert L.avrf_ring_vrf_prove(ctxs[1 - suite]._h, key._h, C.c_size_t(rlen), C.c_size_t(1), nat._u8(bytes.fromhex(v["sk"])), nat._u32([idx]), nat._u8(io_xy),
                                     nat._u32([1]), nat._u8(ad), nat._u32([len(ad)]), 0, out) == nat.ERR_BAD_ARG
        assert bytes(out) == want
        proofs.append(want); coms.append(key.commitment); ads.append(ad)
        ios_w.append(bytes.fromhex(v["h"]) + bytes.fromhex(v["gamma"]))
        key.close()
    n = len(vs)

    def verify(prs, each, validate=1, ios=ios_w):
        out = (C.c_int32 * n)()
        rc = L.avrf_ring_vrf_verify(c._h, setup._h, C.c_size_t(n), nat._u8(b"".join(coms)), C.c_size_t(n), nat._u32(list(range(n))), nat._u8(b"".join(ios)),
                                    nat._u32([1] * n), nat._u8(b"".join(ads)), nat._u32([len(a) for a in ads]), nat._u8(b"".join(prs)), validate, int(each), out)
        return rc, list(out)
    assert verify(proofs, True) == (0, [0] * n)
    assert verify(proofs, False)[0] == 0
    assert L.avrf_ring_vrf_verify(ctxs[1 - suite]._h, setup._h, C.c_size_t(n), nat._u8(b"".join(coms)), C.c_size_t(n), nat._u32(list(range(n))), nat._u8(b"".join(ios_w)),
                                  nat._u32([1] * n), nat._u8(b"".join(ads)), nat._u32([len(a) for a in ads]), nat._u8(b"".join(proofs)), 1, 1, (C.c_int32 * n)()) == nat.ERR_BAD_ARG
    bad = list(proofs); bad[3] = bad[3][:100] + bytes([bad[3][100] ^ 1]) + bad[3][101:]           # Pedersen response
    bad[5] = bad[5][:-10] + bytes([bad[5][-10] ^ 1]) + bad[5][-9:]                                   # ring opening proof
    rc, st = verify(bad, True)
    assert rc == 0 and st[3] == 1 and st[5] in (1, 2) and [st[i] for i in (0, 1, 2, 4, 6)] == [0] * 5
    assert verify(bad, False)[0] in (1, 2)
    ios_bad = list(ios_w); ios_bad[1] = ios_w[1][:32] + ios_w[2][32:]                                  # wrong output
    assert verify(proofs, True, ios=ios_bad)[1] == [0, 1, 0, 0, 0, 0, 0]
    setup.close()


def test_empty_and_bad_arguments(ctxs, golden_dir):
    """Empty batches are Ok(()) everywhere (src/thin.rs:262-264, src/pedersen.rs:343-345); out-of-range ring indices and NULL
    arguments are AVRF_ERR_BAD_ARG, never a crash."""
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.ring import RingSetup, pairing_check, ring_batch_verify, ring_verify_each
    c = ctxs[0]
    L = nat.lib()
    z8, z32 = nat._u8(b""), nat._u32([])
    assert L.avrf_thin_batch_verify_wire(c._h, C.c_size_t(0), z8, z8, z32, z8, z32, z8, 1) == 0
    assert L.avrf_pedersen_batch_verify_wire(c._h, C.c_size_t(0), z8, z32, z8, z32, z8, 1) == 0
    out = (C.c_int32 * 1)()
    assert L.avrf_thin_verify_wire(c._h, C.c_size_t(0), z8, z8, z32, z8, z32, z8, 0, out) == 0
    assert L.avrf_tiny_verify_wire(c._h, C.c_size_t(0), z8, z8, z32, z8, z32, z8, 0, out) == 0
    assert L.avrf_thin_verify_wire(c._h, C.c_size_t(1), None, z8, z32, z8, z32, z8, 0, out) == nat.ERR_BAD_ARG
    srs = open(os.path.join(golden_dir, "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()
    setup = RingSetup(c, srs, 8)
    assert pairing_check(setup, [], []) == [] and ring_verify_each(setup, [], None, [], []) == []
    v = json.load(open(os.path.join(golden_dir, NAMES[0] + "_ring.json")))[0]
    from helpers import xy
    com, inst, proof = bytes.fromhex(v["ring_pks_com"]), xy(0, bytes.fromhex(v["proof_pk_com"])), bytes.fromhex(v["ring_proof"])
    assert ring_batch_verify(setup, [com], [5], [inst], [proof]) == nat.ERR_BAD_ARG           # ring index out of range
    with pytest.raises(nat.AvrfError, match="-> -2"):
        ring_verify_each(setup, [com], [1], [inst], [proof])
    assert L.avrf_ring_vrf_verify(c._h, setup._h, C.c_size_t(0), None, C.c_size_t(0), None, None, None, None, None, None, 1, 1, None) == 0
    assert L.avrf_ctx_set_validation(c._h, 7) == nat.ERR_BAD_ARG
    setup.close()


@pytest.mark.parametrize("kind", [0, 1])
def test_batch_stage_wire_full_size(ctxs, kind):
    """BASELINE configs[1] / [2] from `serialize_compressed` bytes: avrf_*_batch_stage_wire decompresses (and validates) all
    4 x 65 536 / 5 x 65 536 points on the device straight into the staged buffers; the staged batch then IS the x || y batch -- same
    terms, same verdicts (one-call run and the three-call run), a tampered scalar fails, a point that does not decode / a torsion
    point under Validate::Yes is InvalidData and leaves nothing staged."""
    from ark_vrf_amd import _native as nat
    from helpers import compressed_items, nat_batch
    c = ctxs[0]
    n = 65536
    b = orc.gen_batch(0, kind, n, threads=16)
    if kind == 1:
        b["pks_xy"] = b""
    pks, ios, ads, proofs = compressed_items(0, b, kind)
    L = nat.lib()
    io_b = nat._u8(b"".join(i + o for it in ios for i, o in it)); cnt = nat._u32([1] * n); ad_b = nat._u8(b"".join(ads)); adl = nat._u32([len(a) for a in ads])

    def stage(pks_l, proofs_l, validate, io_buf=io_b):
        if kind == 0:
            return L.avrf_thin_batch_stage_wire(c._h, C.c_size_t(n), nat._u8(b"".join(pks_l)), io_buf, cnt, ad_b, adl, nat._u8(b"".join(proofs_l)), validate)
        return L.avrf_pedersen_batch_stage_wire(c._h, C.c_size_t(n), io_buf, cnt, ad_b, adl, nat._u8(b"".join(proofs_l)), validate)
    run = c.thin_batch_run if kind == 0 else c.pedersen_batch_run
    # the xy batch's terms, then the wire-staged batch's terms: identical
    assert (c.thin_batch_stage if kind == 0 else c.pedersen_batch_stage)(nat_batch(b)) == 0 and run() == 0
    want_terms = c.last_terms()
    for validate in (0, 1):
        assert stage(pks, proofs, validate) == 0 and run() == 0
        assert c.last_terms() == want_terms
    assert stage(pks, proofs, 1) == 0
    assert c.batch_run_begin() == 0 and c.batch_run_hash() == 0 and c.batch_run_end() == 0       # the three-call run on a wire-staged batch
    # tampered response scalar of item 40 000
    j = 40000
    p2 = list(proofs); p2[j] = proofs[j][:-20] + bytes([proofs[j][-20] ^ 1]) + proofs[j][-19:]
    assert stage(pks, p2, 1) == 0 and run() == 1
    # a y with no x on the curve in the LAST proof; a torsion point (x, y) -> (-x, -y) as an output point
    bad_y = next(k.to_bytes(32, "little") for k in range(2, 300) if orc.point_decompress(0, k.to_bytes(32, "little"))[0] != 0)
    p3 = list(proofs); p3[n - 1] = bad_y + proofs[n - 1][32:]
    assert stage(pks, p3, 0) == 2
    assert run() == -2                                                                            # nothing staged: BAD_ARG
    q = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    st, oxy = orc.point_decompress(0, ios[123][0][1])
    tors = orc.point_compress(0, ((q - int.from_bytes(oxy[:32], "little")) % q).to_bytes(32, "little") + ((q - int.from_bytes(oxy[32:], "little")) % q).to_bytes(32, "little"))
    ios_t = list(ios); ios_t[123] = [(ios[123][0][0], tors)]
    io_t = nat._u8(b"".join(i + o for it in ios_t for i, o in it))
    assert stage(pks, proofs, 1, io_t) == 2                                                        # Validate::Yes: InvalidData before any equation
    assert stage(pks, proofs, 0, io_t) == 0 and run() in (0, 1)                                    # Validate::No: it decodes; the verdict is the equation's
