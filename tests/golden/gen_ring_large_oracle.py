#!/usr/bin/env python3
"""Generates tests/golden/ring_large_oracle.json: expected outputs of the ring SNARK at the sizes BASELINE configs[3]/[4]
are benchmarked at, computed by the pure-Python oracle (oracle/ring_py.py, itself pinned to the reference's 14 ring vectors
at N = 512).  No reference vector exists at these sizes (SURVEY.md §8c iii) and the oracle needs minutes there, so its
outputs are committed as a fixture and the `-m gpu` tests compare the device path with them byte for byte.

  python tests/golden/gen_ring_large_oracle.py            # ~10 min of CPU

Cases
  bn254_ring4096:  Baby-JubJub / BN254, ring 4096 -> N = 8192; SRS = Kzg::setup with tau fixed below over the generators of
                   data/srs/bn254-testing-2-9-uncompressed.bin (24 577 G1 powers + {g2, tau g2}); sha256 of the URS bytes,
                   ring commitment, one proof (prover index 777, blinding disabled).
Inputs are derived from fixed strings, so the GPU test can rebuild them without this file's help:
  ring keys  pk_i = (sha512("k<i>") mod (r >> 3) + 1) * G      (G = suite generator)
  blinding   b = sha512("blinding") mod (r >> 3)
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as orc                      # noqa: E402
from oracle import pairing_py as PP       # noqa: E402
from oracle import ring_py as R           # noqa: E402

TAU = 0x1234567890abcdef1234567890abcdef
GOLD = os.path.join(ROOT, "tests", "golden")


def fixed_base_powers(s, G, tau, n):
    """[tau^i * G for i < n] (affine) with an 8-bit fixed-base window table of G."""
    p = s.p
    tbl, base = [], (G[0], G[1], 1)
    for w in range(32):
        row, acc = [None], None
        for _ in range(255):
            acc = R.g1_add(p, acc, base)
            row.append(acc)
        tbl.append(row)
        for _ in range(8):
            base = R.g1_dbl(p, base)
    out, k = [], 1
    for _ in range(n):
        acc = None
        for w in range(32):
            d = (k >> (8 * w)) & 255
            if d:
                acc = R.g1_add(p, acc, tbl[w][d])
        out.append(R.g1_affine(p, acc))
        k = k * tau % s.r
    return out


def ring_keys(suite, n):
    s = R.SUITES[suite]
    g = orc.suite_point(suite, 0)
    ks = [(int.from_bytes(hashlib.sha512(b"k%d" % i).digest(), "little") % (s.r >> 3) + 1).to_bytes(32, "little") for i in range(n)]
    xy = []
    for k in ks:
        st, p = orc.point_decompress(suite, orc.smul(suite, k, g))
        assert st == 0
        xy.append(p)
    return b"".join(ks), xy


def case_bn254_ring4096():
    suite, ring = 1, 4096
    s = R.SUITES[suite]
    srs_file = open(os.path.join(GOLD, "bn254-testing-2-9-uncompressed.bin"), "rb").read()
    cnt = int.from_bytes(srs_file[:8], "little")
    g1b = srs_file[8: 8 + 64]
    g2b = srs_file[8 + cnt * 64 + 8: 8 + cnt * 64 + 8 + 128]
    prm = R.Params(s, ring_size=ring)
    assert prm.N == 8192
    n_g1 = 3 * prm.N + 1
    t0 = time.time()
    G = R.g1_decode_uncompressed(s, g1b)
    pw = fixed_base_powers(s, G, TAU % s.r, n_g1)
    PP.use_curve("bn254")
    q = PP.g2_decode_arkworks_uncompressed(g2b)
    tq = PP.g2_mul(q, TAU % s.r)
    neg = lambda P: (P[0], (-P[1]) % s.p)
    assert PP.pairing_product_is_one([(pw[1], q), (neg(pw[0]), tq)])
    urs = n_g1.to_bytes(8, "little") + b"".join(R.g1_encode(s, P, False) for P in pw) + (2).to_bytes(8, "little") + g2b + PP.g2_encode_arkworks_uncompressed(tq)
    PP.use_curve("bls12_381")
    print("srs", time.time() - t0, flush=True)
    srs = R.Srs(s, urs)
    _, pks = ring_keys(suite, ring)
    keys = [(int.from_bytes(p[:32], "little"), int.from_bytes(p[32:], "little")) for p in pks]
    cols = R.index(prm, srs, keys)
    print("index", time.time() - t0, flush=True)
    b = int.from_bytes(hashlib.sha512(b"blinding").digest(), "little") % (s.r >> 3)
    proof, _ = R.prove(prm, srs, cols, 777, b)
    print("prove", time.time() - t0, flush=True)
    return dict(suite=suite, ring_size=ring, domain_size=prm.N, tau=hex(TAU), n_g1=n_g1,
                urs_sha256=hashlib.sha256(urs).hexdigest(), pks_sha256=hashlib.sha256(b"".join(pks)).hexdigest(),
                commitment=R.commitment_bytes(s, cols).hex(), key_index=777, blinding=b.to_bytes(32, "little").hex(), proof=proof.hex())


if __name__ == "__main__":
    out = {"bn254_ring4096": case_bn254_ring4096()}
    with open(os.path.join(GOLD, "ring_large_oracle.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("written")
