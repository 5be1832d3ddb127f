"""Shared test helpers: deterministic inputs built with the CPU oracle (test infrastructure)."""
import hashlib
import random

import oracle as orc

R_ORDER = {
    orc.BANDERSNATCH: 0x1cfb69d4ca675f520cce760202687600ff8f87007419047174fd06b52876e7e1,
    orc.BABYJUBJUB: 2736030358979909402780800718157159386076813972158567259200215660948447373041,
    orc.JUBJUB: 6554484396890773809930967563523245729705921265872317281365359162392183254199,
    orc.ED25519: 2 ** 252 + 27742317777372353535851937790883648493,
    orc.TESTING_SHA256: 2 ** 252 + 27742317777372353535851937790883648493,
    orc.BANDERSNATCH_SW: 0x1cfb69d4ca675f520cce760202687600ff8f87007419047174fd06b52876e7e1,
    orc.BANDERSNATCH_SHAKE128: 0x1cfb69d4ca675f520cce760202687600ff8f87007419047174fd06b52876e7e1,
    orc.SECP256R1: 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551,
}
IDENTITY_XY = bytes(32) + (1).to_bytes(32, "little")


def rand_scalar(rng, suite, bits=None):
    r = R_ORDER[suite]
    v = rng.getrandbits(bits) if bits else rng.randrange(r)
    return (v % r).to_bytes(32, "little")


def rand_points_xy(rng, suite, n):
    """n pseudo-random prime-order-subgroup points as xy bytes (k_i * G via the oracle)."""
    g = orc.suite_point(suite, 0)
    out = []
    for _ in range(n):
        k = rand_scalar(rng, suite)
        st, xy = orc.point_decompress(suite, orc.smul(suite, k, g))
        assert st == 0
        out.append(xy)
    return out


def xy(suite, comp):
    st, out = orc.point_decompress(suite, comp)
    assert st == 0
    return out


def nat_batch(b, with_sks=False, with_proofs=True):
    """oracle.gen_batch dict -> ark_vrf_amd Batch."""
    from ark_vrf_amd._native import Batch
    return Batch(b["n"], b["ios_xy"], b["io_counts"], b["ads"], b["ad_lens"], pks_xy=b["pks_xy"] or None,
                 proofs=b["proofs"] if with_proofs else None, sks=b["sks"] if with_sks else None)


def compressed_items(suite, b, kind):
    """oracle.gen_batch dict -> (pks, ios, ads, proofs) in the oracle's compressed encodings."""
    n = b["n"]
    psz = 96 if kind == 0 else 256
    pks = [orc.point_compress(suite, b["pks_xy"][64 * j: 64 * j + 64]) for j in range(n)] if b["pks_xy"] else []
    ios, ads, proofs, off = [], [], [], 0
    for j in range(n):
        i = orc.point_compress(suite, b["ios_xy"][128 * j: 128 * j + 64])
        o = orc.point_compress(suite, b["ios_xy"][128 * j + 64: 128 * j + 128])
        ios.append([(i, o)])
        ads.append(b["ads"][off: off + b["ad_lens"][j]]); off += b["ad_lens"][j]
        pr = b["proofs"][psz * j: psz * (j + 1)]
        if kind == 0:
            proofs.append(orc.point_compress(suite, pr[:64]) + pr[64:])
        else:
            proofs.append(b"".join(orc.point_compress(suite, pr[64 * k: 64 * k + 64]) for k in range(3)) + pr[192:])
    return pks, ios, ads, proofs


def proof_xy(suite, comp, kind):
    """compressed proof (64 / 160 bytes) -> ABI proof (96 / 256 bytes)."""
    if kind == 0:
        return xy(suite, comp[:32]) + comp[32:]
    return b"".join(xy(suite, comp[32 * k: 32 * k + 32]) for k in range(3)) + comp[96:]


def proof_comp(suite, p, kind):
    if kind == 0:
        return orc.point_compress(suite, p[:64]) + p[64:]
    return b"".join(orc.point_compress(suite, p[64 * k: 64 * k + 64]) for k in range(3)) + p[192:]
