"""oracle -- ctypes binding of the CPU parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (ark_vrf_amd) never does.  The C sources under oracle/
restate the reference path (each function cites /root/reference file:line) and are
pinned to the reference's known-answer vectors by tests/test_oracle_golden.py.
"""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "liborc.so")

BANDERSNATCH = 0
BABYJUBJUB = 1
JUBJUB = 2
ED25519 = 3
TESTING_SHA256 = 6            # the crate's test suite: edwards25519 with HashTranscript<Sha256>
BANDERSNATCH_SHAKE128 = 5     # Bandersnatch with the SHAKE128 transcript (XofTranscript<Shake128>)
SECP256R1 = 7          # NIST P-256 with HashTranscript<Sha256>: a short-Weierstrass suite, compressed points are 33 bytes
BANDERSNATCH_SW = 4    # Bandersnatch in its short-Weierstrass presentation: 33-byte serialised points (sw_encode / sw_decode)

OK, VERIFICATION_FAILURE, INVALID_DATA = 0, 1, 2


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _buf(n):
    return (C.c_uint8 * n)()


def _b(x):
    return bytes(x)


def _u8(data):
    data = bytes(data)
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data.ljust(1, b"\0"))


def _u32(vals):
    vals = list(vals)
    return (C.c_uint32 * max(1, len(vals)))(*vals)


def pt_len(suite):
    """bytes of a compressed point at the oracle's entry points (suite_t.pt_len)"""
    return 33 if suite == SECP256R1 else 32


def from_seed(suite, seed):
    sk, pk = _buf(32), _buf(pt_len(suite))
    assert lib().orc_from_seed(suite, _u8(seed), sk, pk) == 0
    return _b(sk), _b(pk)


def sk_to_pk(suite, sk):
    pk = _buf(pt_len(suite))
    assert lib().orc_sk_to_pk(suite, _u8(sk), pk) == 0
    return _b(pk)


def hash_to_curve(suite, data):
    out = _buf(pt_len(suite))
    st = lib().orc_hash_to_curve(suite, _u8(data), C.c_size_t(len(data)), out)
    assert st == 0, st
    return _b(out)


def vrf_output(suite, sk, inp):
    out = _buf(pt_len(suite))
    assert lib().orc_vrf_output(suite, _u8(sk), _u8(inp), out) == 0
    return _b(out)


def point_to_hash(suite, pt, n=32):
    out = _buf(n)
    assert lib().orc_point_to_hash(suite, _u8(pt), out, C.c_size_t(n)) == 0
    return _b(out)


def thin_prove(suite, sk, ios, ad):
    """ios: list of (input32, output32)."""
    iob = b"".join(i + o for i, o in ios)
    proof = _buf(pt_len(suite) + 32)
    st = lib().orc_thin_prove(suite, _u8(sk), _u8(iob), C.c_size_t(len(ios)), _u8(ad), C.c_size_t(len(ad)), proof)
    assert st == 0, st
    return _b(proof)


def thin_verify(suite, pk, ios, ad, proof):
    iob = b"".join(i + o for i, o in ios)
    return lib().orc_thin_verify(suite, _u8(pk), _u8(iob), C.c_size_t(len(ios)), _u8(ad), C.c_size_t(len(ad)), _u8(proof))


def tiny_prove(suite, sk, ios, ad):
    """tiny::Prover::prove (src/tiny.rs:163-176): 48-byte proof LE16(c) || LE32(s)."""
    iob = b"".join(i + o for i, o in ios)
    proof = _buf(48)
    st = lib().orc_tiny_prove(suite, _u8(sk), _u8(iob), C.c_size_t(len(ios)), _u8(ad), C.c_size_t(len(ad)), proof)
    assert st == 0, st
    return _b(proof)


def tiny_verify(suite, pk, ios, ad, proof):
    iob = b"".join(i + o for i, o in ios)
    return lib().orc_tiny_verify(suite, _u8(pk), _u8(iob), C.c_size_t(len(ios)), _u8(ad), C.c_size_t(len(ad)), _u8(proof))


def _batch_args(items_ios, ads):
    iob = b"".join(i + o for ios in items_ios for i, o in ios)
    counts = [len(ios) for ios in items_ios]
    adb = b"".join(ads)
    adl = [len(a) for a in ads]
    return _u8(iob), _u32(counts), _u8(adb), _u32(adl), sum(counts)


def thin_batch_verify(suite, pks, items_ios, ads, proofs):
    n = len(pks)
    iob, cnt, adb, adl, _ = _batch_args(items_ios, ads)
    return lib().orc_thin_batch_verify(suite, C.c_size_t(n), _u8(b"".join(pks)), iob, cnt, adb, adl, _u8(b"".join(proofs)))


def thin_batch_terms(suite, pks, items_ios, ads, proofs):
    """Returns (status, bases_xy bytes, scalars bytes) of the batch MSM (src/thin.rs:282-317)."""
    n = len(pks)
    iob, cnt, adb, adl, tot = _batch_args(items_ios, ads)
    cap = 2 * n + 2 * tot + 1
    bases, sc = _buf(cap * 64), _buf(cap * 32)
    k = C.c_size_t(0)
    st = lib().orc_thin_batch_terms(suite, C.c_size_t(n), _u8(b"".join(pks)), iob, cnt, adb, adl,
                                    _u8(b"".join(proofs)), bases, sc, C.byref(k))
    return st, _b(bases)[: 64 * k.value], _b(sc)[: 32 * k.value]


def pedersen_prove(suite, sk, ios, ad):
    iob = b"".join(i + o for i, o in ios)
    proof, bl = _buf(3 * pt_len(suite) + 64), _buf(32)
    st = lib().orc_pedersen_prove(suite, _u8(sk), _u8(iob), C.c_size_t(len(ios)), _u8(ad), C.c_size_t(len(ad)), proof, bl)
    assert st == 0, st
    return _b(proof), _b(bl)


def pedersen_verify(suite, ios, ad, proof):
    iob = b"".join(i + o for i, o in ios)
    return lib().orc_pedersen_verify(suite, _u8(iob), C.c_size_t(len(ios)), _u8(ad), C.c_size_t(len(ad)), _u8(proof))


def pedersen_batch_verify(suite, items_ios, ads, proofs):
    n = len(proofs)
    iob, cnt, adb, adl, _ = _batch_args(items_ios, ads)
    return lib().orc_pedersen_batch_verify(suite, C.c_size_t(n), iob, cnt, adb, adl, _u8(b"".join(proofs)))


def pedersen_batch_terms(suite, items_ios, ads, proofs):
    n = len(proofs)
    iob, cnt, adb, adl, _ = _batch_args(items_ios, ads)
    cap = 5 * n + 2
    bases, sc = _buf(cap * 64), _buf(cap * 32)
    k = C.c_size_t(0)
    st = lib().orc_pedersen_batch_terms(suite, C.c_size_t(n), iob, cnt, adb, adl, _u8(b"".join(proofs)),
                                        bases, sc, C.byref(k))
    return st, _b(bases)[: 64 * k.value], _b(sc)[: 32 * k.value]


def msm(suite, bases_xy, scalars, algo=1):
    n = len(scalars) // 32
    assert len(bases_xy) == 64 * n
    out = _buf(64)
    st = lib().orc_msm(suite, C.c_size_t(n), _u8(bases_xy), _u8(scalars), out, algo)
    assert st == 0, st
    return _b(out)


def point_decompress(suite, pt, validate=False):
    out = _buf(64)
    st = lib().orc_point_decompress(suite, _u8(pt), out, int(validate))
    return st, _b(out)


def point_compress(suite, xy):
    out = _buf(pt_len(suite))
    assert lib().orc_point_compress(suite, _u8(xy), out) == 0
    return _b(out)


def suite_point(suite, which):
    out = _buf(pt_len(suite))
    lib().orc_suite_point(suite, which, out)
    return _b(out)


def smul(suite, k, pt):
    out = _buf(pt_len(suite))
    assert lib().orc_smul(suite, _u8(k), _u8(pt), out) == 0
    return _b(out)


def sha512(data):
    out = _buf(64)
    lib().orc_sha512(_u8(data), C.c_size_t(len(data)), out)
    return _b(out)


def gen_batch(suite, kind, n, run_seed=bytes(32), start=0, threads=8):
    """Deterministic synthetic batch (SURVEY.md §8d recipe), generated by the oracle's prover on
    `threads` host threads.  Returns a dict of bytes in the C-ABI layouts of include/avrf.h:
    sks, pks_xy, ios_xy, ads, ad_lens (list), proofs; io_counts is all ones."""
    from concurrent.futures import ThreadPoolExecutor
    L = lib()
    psz = 96 if kind == 0 else 256
    chunks = []
    per = max(1, (n + threads - 1) // threads)
    for lo in range(0, n, per):
        chunks.append((lo, min(per, n - lo)))

    def work(ch):
        lo, cnt = ch
        sks, pks, ios = _buf(32 * cnt), _buf(64 * cnt), _buf(128 * cnt)
        ads, adl, prf = _buf(24 * cnt), (C.c_uint32 * cnt)(), _buf(psz * cnt)
        st = L.orc_gen_batch(suite, kind, _u8(run_seed), C.c_uint64(start + lo), C.c_size_t(cnt), sks, pks, ios, ads, adl, prf)
        assert st == 0, st
        lens = list(adl)
        return _b(sks), _b(pks), _b(ios), _b(ads)[: sum(lens)], lens, _b(prf)

    if n == 0:
        return dict(n=0, sks=b"", pks_xy=b"", ios_xy=b"", ads=b"", ad_lens=[], io_counts=[], proofs=b"")
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(work, chunks))
    return dict(n=n, sks=b"".join(p[0] for p in parts), pks_xy=b"".join(p[1] for p in parts),
                ios_xy=b"".join(p[2] for p in parts), ads=b"".join(p[3] for p in parts),
                ad_lens=[x for p in parts for x in p[4]], io_counts=[1] * n, proofs=b"".join(p[5] for p in parts))


def thin_batch_verify_raw(suite, b):
    """orc_thin_batch_verify on a gen_batch dict: converts xy -> compressed first."""
    n = b["n"]
    pks = [point_compress(suite, b["pks_xy"][64 * j: 64 * j + 64]) for j in range(n)]
    ios, ads, proofs, off = [], [], [], 0
    for j in range(n):
        i = point_compress(suite, b["ios_xy"][128 * j: 128 * j + 64]); o = point_compress(suite, b["ios_xy"][128 * j + 64: 128 * j + 128])
        ios.append([(i, o)])
        ads.append(b["ads"][off: off + b["ad_lens"][j]]); off += b["ad_lens"][j]
        pr = b["proofs"][96 * j: 96 * j + 96]
        proofs.append(point_compress(suite, pr[:64]) + pr[64:])
    return pks, ios, ads, proofs


def thin_batch_verify_xy(suite, b):
    """thin::BatchVerifier on a gen_batch-style dict in the C-ABI (xy) layout; returns the status."""
    return lib().orc_thin_batch_verify_xy(suite, C.c_size_t(b["n"]), _u8(b["pks_xy"]), _u8(b["ios_xy"]), _u32(b["io_counts"]),
                                          _u8(b["ads"]), _u32(b["ad_lens"]), _u8(b["proofs"]))


def pedersen_batch_verify_xy(suite, b):
    return lib().orc_pedersen_batch_verify_xy(suite, C.c_size_t(b["n"]), _u8(b["ios_xy"]), _u32(b["io_counts"]),
                                              _u8(b["ads"]), _u32(b["ad_lens"]), _u8(b["proofs"]))


def thin_batch_terms_xy(suite, b):
    """(status, bases_xy, scalars) of the batch MSM (src/thin.rs:282-317) for a gen_batch-style dict (xy layout)."""
    n = b["n"]; cap = 2 * n + 2 * sum(b["io_counts"]) + 1
    bases, sc, k = _buf(cap * 64), _buf(cap * 32), C.c_size_t(0)
    st = lib().orc_thin_batch_terms_xy(suite, C.c_size_t(n), _u8(b["pks_xy"]), _u8(b["ios_xy"]), _u32(b["io_counts"]), _u8(b["ads"]),
                                       _u32(b["ad_lens"]), _u8(b["proofs"]), bases, sc, C.byref(k))
    return st, _b(bases)[: 64 * k.value], _b(sc)[: 32 * k.value]


def pedersen_batch_terms_xy(suite, b):
    n = b["n"]; cap = 5 * n + 2
    bases, sc, k = _buf(cap * 64), _buf(cap * 32), C.c_size_t(0)
    st = lib().orc_pedersen_batch_terms_xy(suite, C.c_size_t(n), _u8(b["ios_xy"]), _u32(b["io_counts"]), _u8(b["ads"]),
                                           _u32(b["ad_lens"]), _u8(b["proofs"]), bases, sc, C.byref(k))
    return st, _b(bases)[: 64 * k.value], _b(sc)[: 32 * k.value]


def sw_encode(suite, te32):
    """twisted-Edwards 32-byte form (what every oracle entry point takes) -> the suite's 33-byte SW wire form"""
    out = _buf(33)
    st = lib().orc_sw_encode(suite, _u8(te32), out)
    assert st == 0, st
    return _b(out)


def sw_decode(suite, sw33):
    out = _buf(32)
    st = lib().orc_sw_decode(suite, _u8(sw33), out)
    return st, _b(out)
