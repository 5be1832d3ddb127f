"""Host-side product logic that needs no GPU: the sequential weight transcript (host_sha512.h via
avrf_batch_weight_seed), the host point arithmetic that finishes an MSM / combines per-GPU partials
(host_te.h via avrf_points_sum), batch packing and sharding helpers."""
import ctypes as C
import hashlib
import random

import pytest

import oracle as orc
from helpers import IDENTITY_XY, rand_points_xy, rand_scalar

SUITE_ID = {0: b"Bandersnatch-SHA512-ELL2-v1", 1: b"BabyJubJub-SHA512-TAI-v1"}


@pytest.mark.parametrize("suite", [0, 1])
@pytest.mark.parametrize("pedersen", [0, 1])
def test_weight_seed_matches_transcript(suite, pedersen):
    """src/thin.rs:274-279 / src/pedersen.rs:361-367: new(SUITE_ID); absorb [0x50]; per item
    LE32(c) || LE32(s) [|| LE32(sb)]; seed = SHA-512 of all of it (src/utils/transcript.rs:227-240)."""
    from ark_vrf_amd import _native as nat
    rng = random.Random(3)
    for n in [0, 1, 2, 7, 300]:
        cs = [rng.getrandbits(128).to_bytes(16, "little") for _ in range(n)]
        rs = [b"".join(rand_scalar(rng, suite) for _ in range(2 if pedersen else 1)) for _ in range(n)]
        h = hashlib.sha512(SUITE_ID[suite] + b"\x50")
        for c, r in zip(cs, rs):
            h.update(c + bytes(16) + r)
        seed = (C.c_uint8 * 64)()
        st = nat.lib().avrf_batch_weight_seed(suite, pedersen, C.c_size_t(n), nat._u8(b"".join(cs)), nat._u8(b"".join(rs)), seed)
        assert st == 0 and bytes(seed) == h.digest()


@pytest.mark.parametrize("suite", [0, 1])
def test_points_sum_matches_oracle(suite):
    from ark_vrf_amd import _native as nat
    rng = random.Random(11 + suite)
    pts = rand_points_xy(rng, suite, 9) + [IDENTITY_XY]
    out = (C.c_uint8 * 64)()
    assert nat.lib().avrf_points_sum(suite, C.c_size_t(len(pts)), nat._u8(b"".join(pts)), out) == 0
    ones = b"".join((1).to_bytes(32, "little") for _ in pts)
    assert bytes(out) == orc.msm(suite, b"".join(pts), ones, algo=0)
    assert nat.lib().avrf_points_sum(suite, C.c_size_t(0), None, out) == 0 and bytes(out) == IDENTITY_XY
    bad = b"\xff" * 64
    assert nat.lib().avrf_points_sum(suite, C.c_size_t(1), nat._u8(bad), out) == nat.INVALID_DATA


def test_shard_helpers():
    from ark_vrf_amd.dist import shard_range, shard_thin_batch
    for n in [0, 1, 5, 8, 65536, 65537]:
        for world in [1, 2, 3, 8]:
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
    b = dict(n=3, pks_xy=bytes(range(192)), ios_xy=bytes(128 * 4), io_counts=[2, 0, 2], ads=b"abcdef", ad_lens=[1, 2, 3],
             proofs=bytes(96 * 3))
    s = shard_thin_batch(b, 1, 3)
    assert s["n"] == 2 and s["ads"] == b"bcdef" and s["io_counts"] == [0, 2] and len(s["ios_xy"]) == 256
    assert s["pks_xy"] == bytes(range(64, 192))


def test_batch_packing():
    from ark_vrf_amd._native import Batch
    b = Batch.from_items([[(b"i" * 64, b"o" * 64)], [], [(b"a" * 64, b"b" * 64), (b"c" * 64, b"d" * 64)]], [b"x", b"", b"yz"],
                         pks_xy=[b"p" * 64] * 3, proofs=[b"q" * 96] * 3)
    assert b.n == 3 and list(b.io_counts)[:3] == [1, 0, 2] and list(b.ad_lens)[:3] == [1, 0, 2]
    assert bytes(b.ads)[:3] == b"xyz" and bytes(b.ios_xy)[:128] == b"i" * 64 + b"o" * 64


@pytest.mark.parametrize("curve,srs", [(0, "bls12-381-srs-2-11-uncompressed-zcash.bin"), (1, "bn254-testing-2-9-uncompressed.bin")])
def test_host_pairing_on_reference_srs(tmp_path, golden_dir, curve, srs):
    """Product host code (host_pairing.h / host_g1.h, plain C++): e(tau g1, g2) == e(g1, tau g2) on the
    reference's SRS files for both pairing curves, and a negative control.  Compiled with g++, no GPU."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "hp")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "host_pairing_check.cpp"), "-o", exe])
    out = subprocess.run([exe, str(curve), os.path.join(golden_dir, srs)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip() == "consistent=1 negative=0 consistent_high=1 g2_codec=1 g2_law=1 cyclo_sqr=1 f12_inv=1 line_tables=1"


def test_multibuffer_weight_hash_matches_scalar():
    """host_sha512_mb.h: eight weight transcripts in the lanes of an AVX-512 register must give the digests of the scalar
    chain (avrf_batch_weight_seed), for equal and unequal batch sizes, empty batches, both record sizes and every lane count."""
    import ctypes as C
    import random
    from ark_vrf_amd import _native as nat
    L = nat.lib()
    rng = random.Random(11)
    for pedersen in (0, 1):
        rsz = 64 if pedersen else 32
        for sizes in ([5], [0, 1], [3, 3, 3, 3, 3, 3, 3, 3], [0, 1, 2, 3, 4, 5, 6, 700], [129, 64, 1000, 2, 0, 17, 333], [2048] * 8):
            cs = [bytes(rng.getrandbits(8) for _ in range(16 * n)) for n in sizes]
            rs = [bytes(rng.getrandbits(8) for _ in range(rsz * n)) for n in sizes]
            want = []
            for n, c, r in zip(sizes, cs, rs):
                out = (C.c_uint8 * 64)()
                assert L.avrf_batch_weight_seed(0, pedersen, C.c_size_t(n), nat._u8(c or b"\0"), nat._u8(r or b"\0"), out) == 0
                want.append(bytes(out))
            k = len(sizes)
            cbuf = [C.create_string_buffer(c or b"\0", max(1, len(c))) for c in cs]
            rbuf = [C.create_string_buffer(r or b"\0", max(1, len(r))) for r in rs]
            cp = (C.c_void_p * k)(*[C.cast(b, C.c_void_p) for b in cbuf])
            rp = (C.c_void_p * k)(*[C.cast(b, C.c_void_p) for b in rbuf])
            ns = (C.c_size_t * k)(*sizes)
            out = (C.c_uint8 * (64 * k))()
            rc = L.avrf_batch_weight_seeds_x8(0, pedersen, k, ns, cp, rp, out)
            if rc == nat.ERR_NO_DEVICE:
                import pytest
                pytest.skip("host CPU without AVX-512")
            assert rc == 0
            assert [bytes(out)[64 * i: 64 * i + 64] for i in range(k)] == want
            # the contiguous form the batch verifiers hand to the hash service, against an independent SHA-512
            import hashlib
            prefix = b"Bandersnatch-SHA512-ELL2-v1" + bytes([0x50])
            msgs = [prefix + b"".join(c[16 * i: 16 * i + 16] + bytes(16) + r[rsz * i: rsz * i + rsz] for i in range(n)) for n, c, r in zip(sizes, cs, rs)]
            mbuf = [C.create_string_buffer(m, len(m)) for m in msgs]
            mp = (C.c_void_p * k)(*[C.cast(b, C.c_void_p) for b in mbuf])
            ls = (C.c_size_t * k)(*[len(m) for m in msgs])
            out2 = (C.c_uint8 * (64 * k))()
            assert L.avrf_sha512_x8(k, mp, ls, out2) == 0
            assert [bytes(out2)[64 * i: 64 * i + 64] for i in range(k)] == [hashlib.sha512(m).digest() for m in msgs] == want


def test_multibuffer_hash_sixteen_lanes_matches_hashlib():
    """host_sha512_mb.h sha512_weights_x16 (what the pool calls; compiled by g++, host_hash.cpp): one group through the transposing
    loader up to eight messages, two interleaved groups above, lanes of unequal length, empty messages, lengths around the
    block and padding boundaries -- against hashlib."""
    import ctypes as C
    import hashlib
    import random
    from ark_vrf_amd import _native as nat
    L = nat.lib()
    rng = random.Random(16)
    edge = [0, 1, 111, 112, 113, 127, 128, 129, 239, 240, 255, 256, 1000, 4096 + 27, 64 * 700 + 28, 3]
    for k in (1, 7, 8, 9, 12, 16):
        for lens in ([edge[(i + k) % 16] for i in range(k)], [rng.randrange(0, 5000) for _ in range(k)], [2048 * 64 + 28] * k):
            msgs = [bytes(rng.getrandbits(8) for _ in range(n)) for n in lens]
            mbuf = [C.create_string_buffer(m or b"\0", max(1, len(m))) for m in msgs]
            mp = (C.c_void_p * k)(*[C.cast(b, C.c_void_p) for b in mbuf])
            ls = (C.c_size_t * k)(*lens)
            out = (C.c_uint8 * (64 * k))()
            rc = L.avrf_sha512_x16(k, mp, ls, out)
            if rc == nat.ERR_NO_DEVICE:
                import pytest
                pytest.skip("host CPU without AVX-512BW")
            assert rc == 0
            assert [bytes(out)[64 * i: 64 * i + 64] for i in range(k)] == [hashlib.sha512(m).digest() for m in msgs], (k, lens)


def test_host_sha512_long_message_path(tmp_path):
    """host_sha512.h: inputs of >= 4096 bytes in one update take the four-blocks-per-step path (message schedule of the next
    four blocks on the vector pipes, AVX-512VL; the scalar rounds read K + W from a buffer).  The digest must not depend on how
    the message is cut into updates, and must be SHA-512 (hashlib) -- lengths around the switch and the 4 MiB weight transcript."""
    import hashlib
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "hs")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-march=x86-64-v3", os.path.join(ROOT, "tests", "cpp", "host_sha512_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "consistent=1" in out.stdout, out.stdout + out.stderr
    n = (4 << 20) + 28
    d = bytes((((i * 2654435761) & 0xffffffff) >> 13) & 0xff for i in range(0, n))
    assert out.stdout.strip().split()[-1] == hashlib.sha512(d).hexdigest()[:8]


def test_host_shake128_and_challenge_reduction(tmp_path):
    """host_shake128.h (the ark-transcript of the ring proof, the SHAKE128 suite's weight transcript): XOF output == hashlib.shake_128 for
    messages around the 168-byte rate, absorbed in one piece and in odd pieces, squeezed past one block.  ring.hip's reduction of a
    challenge's 48 big-endian bytes (hi 2^256 + lo with three Montgomery products) == the integer mod r, on both scalar fields of the
    KZG commitments, including all-ones bytes."""
    import hashlib
    import os
    import random
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "hk")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "host_shake_check.cpp"), "-o", exe])
    rng = random.Random(11)
    for n in (0, 1, 167, 168, 169, 335, 336, 337, 1000):
        m = bytes(rng.randrange(256) for _ in range(n))
        for cut, outn in ((max(n, 1), 48), (7, 200), (168, 400)):
            got = subprocess.run([exe, "shake", m.hex() or "", str(outn), str(cut)], capture_output=True, text=True).stdout.strip()
            assert got == hashlib.shake_128(m).hexdigest(outn), (n, cut, outn)
    R = (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,      # Fr(BLS12-381) = Fq(Bandersnatch)
         0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001)      # Fr(BN254) = Fq(Baby-JubJub)
    cases = [bytes(48), b"\xff" * 48, bytes(16) + b"\xff" * 32, b"\xff" * 16 + bytes(32)] + [bytes(rng.randrange(256) for _ in range(48)) for _ in range(40)]
    for f in (0, 1):
        for b in cases + [(R[f] - 1).to_bytes(48, "big"), R[f].to_bytes(48, "big"), ((R[f] << 128) + 5).to_bytes(48, "big")]:
            got = subprocess.run([exe, "be48", str(f), b.hex()], capture_output=True, text=True).stdout.strip()
            assert int(got, 16) == int.from_bytes(b, "big") % R[f], (f, b.hex())
