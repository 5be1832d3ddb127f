#!/usr/bin/env python3
"""Single-call latencies through the C ABI on one GPU (DESIGN.md section 5 "One item"; references: benches/SUMMARY.md:53-88):
   python tools/latency_report.py
Thin / Pedersen verify and prove with n = 1, 8, 64; the reference's own batch-size sweep (benches/thin.rs:41 BATCH_SIZES = 1 .. 256;
its published table: benches/SUMMARY.md:43-47,57-61,81-88) for the Thin, Pedersen and ring-VRF BatchVerifiers; one ring verification
(avrf_ring_batch_verify / avrf_ring_verify_each, n = 1); one complete ring-VRF verification from wire bytes (avrf_ring_vrf_verify,
with and without Validate::Yes); decompression of five points.  Best of 20-50 calls each."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (make_batch / derive_scalars: synthetic items built on the GPU itself, as the bench line's are)
from ark_vrf_amd import _native as nat  # noqa: E402
from ark_vrf_amd.ring import RingSetup, ring_batch_verify, ring_verify_each  # noqa: E402


def best(fn, reps=30):
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t = min(t, time.perf_counter() - t0)
    return t * 1e3


c = nat.Context(0)
L = nat.lib()


def items(n):
    """n synthetic one-pair items (bench.make_batch): thin verifier / prover batches and the Pedersen ones over the same keys"""
    vb, raw = bench.make_batch(c, nat, n, start=0)
    sks = bench.derive_scalars(b"avrf-bench-sk", 0, n, bench.R_BANDERSNATCH)
    pb = nat.Batch(n, raw["ios_xy"], raw["io_counts"], raw["ads"], raw["ad_lens"], pks_xy=raw["pks_xy"], sks=sks)
    ped, _ = c.pedersen_prove(pb)
    pvb = nat.Batch(n, raw["ios_xy"], raw["io_counts"], raw["ads"], raw["ad_lens"], proofs=ped)
    return vb, pb, pvb, raw


for n in (1, 8, 64):
    vb, pb, pvb, raw = items(n)
    assert c.thin_verify(vb) == [0] * n and c.thin_prove(pb) == raw["proofs"]
    print(f"thin      n={n:<5} verify {best(lambda: c.thin_verify(vb)):.3f} ms   prove {best(lambda: c.thin_prove(pb)):.3f} ms", flush=True)
    assert c.pedersen_verify(pvb) == [0] * n
    print(f"pedersen  n={n:<5} verify {best(lambda: c.pedersen_verify(pvb)):.3f} ms   prove {best(lambda: c.pedersen_prove(pb)):.3f} ms", flush=True)
for n in (1, 8, 32):
    ks = bench.derive_scalars(b"k", 0, n, bench.R_BANDERSNATCH); pts = c.scalar_mul_base(ks)
    ss = bench.derive_scalars(b"s", 0, n, bench.R_BANDERSNATCH)
    c.scalar_mul(ss, pts)
    print(f"common    n={n:<5} avrf_scalar_mul (Secret::output, vrf_output) {best(lambda: c.scalar_mul(ss, pts)):.3f} ms   "
          f"avrf_scalar_mul_base (public key) {best(lambda: c.scalar_mul_base(ks)):.3f} ms", flush=True)
# ---- the reference's batch-size sweep: BatchVerifier::verify (prepare included) for 1 .. 256 items; ring-VRF on a ring of 1023 keys
REF = {"thin": [0.5035, 0.5746, 0.7301, 1.68, 2.00, 3.40, 5.81, 7.77, 14.3],             # benches/SUMMARY.md:61 (batch_verify) + :60 (batch_prepare)
       "pedersen": [0.5294, 0.6061, 0.7747, 1.72, 2.09, 3.52, 6.04, 8.53, 15.6],        # :47 + :46
       "ring": [3.35, 3.91, 5.26, 7.76, 11.09, 18.94, 28.46, 49.66, 84.04]}             # :88 (batch_verify; + :86 batch_prepare_par)
PREP = {"thin": [0.00187, 0.00374, 0.00747, 0.0142, 0.0287, 0.0583, 0.1169, 0.2341, 0.4978],
        "pedersen": [0.00121, 0.00241, 0.00487, 0.00929, 0.0197, 0.0367, 0.0735, 0.1582, 0.3087],
        "ring": [0.0424, 0.0801, 0.1172, 0.1834, 0.2419, 0.2557, 0.2109, 0.4516, 0.8404]}
SIZES = [1, 2, 4, 8, 16, 32, 64, 128, 256]
RING = 1023
from ark_vrf_amd.ring import srs_generate  # noqa: E402
srs_b = open(os.path.join(ROOT, "tests", "golden", "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()
have = int.from_bytes(srs_b[:8], "little")
if have < 3 * (1 << (RING + 4 + 253 - 1).bit_length()) + 1:
    srs_b = srs_generate(c, 0, 0x1234567890abcdef1234567890abcdef, srs_b[8: 8 + 96], srs_b[8 + have * 96 + 8: 8 + have * 96 + 8 + 192], RING)
rsetup = RingSetup(c, srs_b, RING)
rsks = bench.derive_scalars(b"avrf-ring-sk", 0, RING, bench.R_BANDERSNATCH)
rpks = c.scalar_mul_base(rsks)
rkey = rsetup.index([rpks[64 * i: 64 * i + 64] for i in range(RING)])
nmax = SIZES[-1]
idx = [(7 * j + 3) % RING for j in range(nmax)]
inputs = c.scalar_mul_base(bench.derive_scalars(b"avrf-ring-input", 0, nmax, bench.R_BANDERSNATCH))
psk = b"".join(rsks[32 * k: 32 * k + 32] for k in idx); ppk = b"".join(rpks[64 * k: 64 * k + 64] for k in idx)
outs = c.scalar_mul(psk, inputs)
rios = b"".join(inputs[64 * j: 64 * j + 64] + outs[64 * j: 64 * j + 64] for j in range(nmax))
rads = [b"ad-%d" % j for j in range(nmax)]
ped_all, blind = c.pedersen_prove(nat.Batch(nmax, rios, [1] * nmax, b"".join(rads), [len(x) for x in rads], pks_xy=ppk, sks=psk))
rproofs = rkey.prove(idx, [blind[32 * j: 32 * j + 32] for j in range(nmax)])
rows = {"thin": [], "pedersen": [], "ring": []}
for n in SIZES:
    vb, pb, pvb, raw = items(n)
    f = lambda: L.avrf_thin_batch_verify(c._h, C.c_size_t(vb.n), vb.pks_xy, vb.ios_xy, vb.io_counts, vb.ads, vb.ad_lens, vb.proofs)
    g = lambda: L.avrf_pedersen_batch_verify(c._h, C.c_size_t(pvb.n), pvb.ios_xy, pvb.io_counts, pvb.ads, pvb.ad_lens, pvb.proofs)
    rvb = nat.Batch(n, rios[: 128 * n], [1] * n, b"".join(rads[:n]), [len(x) for x in rads[:n]], proofs=ped_all[: 256 * n])
    insts = [ped_all[256 * j: 256 * j + 64] for j in range(n)]

    def h():                                                # ring::BatchVerifier::verify: the Pedersen batch and the ring batch (src/ring.rs:713-735)
        assert L.avrf_pedersen_batch_verify(c._h, C.c_size_t(rvb.n), rvb.ios_xy, rvb.io_counts, rvb.ads, rvb.ad_lens, rvb.proofs) == 0
        assert ring_batch_verify(rsetup, [rkey.commitment], None, insts, rproofs[:n]) == 0
    assert f() == 0 and g() == 0
    h()
    rows["thin"].append(best(f, 15)); rows["pedersen"].append(best(g, 15)); rows["ring"].append(best(h, 8))
print("batch-size sweep, ms per BatchVerifier::verify call (prepare included); reference = benches/SUMMARY.md batch_prepare + batch_verify, one CPU thread")
print("           " + "".join(f"n={n:<8}" for n in SIZES))
for k in ("thin", "pedersen", "ring"):
    print(f"{k:<9}  " + "".join(f"{v:<10.3f}" for v in rows[k]) + " this engine")
    print(f"{'':<9}  " + "".join(f"{a + b:<10.3f}" for a, b in zip(REF[k], PREP[k])) + " reference")
    print(f"{'':<9}  " + "".join(f"{(a + b) / v:<10.1f}" for a, b, v in zip(REF[k], PREP[k], rows[k])) + " x")

gdir = os.path.join(ROOT, "tests", "golden")
vs = json.load(open(os.path.join(gdir, "bandersnatch_sha-512_ell2_ring.json")))
srs = open(os.path.join(gdir, "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()
setup = RingSetup(c, srs, 8)
v = vs[0]
raw = bytes.fromhex(v["ring_pks"])
xy = lambda comp: c.points_decompress(comp)[0]
key = setup.index([xy(raw[32 * i: 32 * i + 32]) for i in range(len(raw) // 32)])
inst = xy(bytes.fromhex(v["proof_pk_com"])); rp = bytes.fromhex(v["ring_proof"])
assert ring_batch_verify(setup, [key.commitment], None, [inst], [rp]) == 0 and ring_verify_each(setup, [key.commitment], None, [inst], [rp]) == [0]
print(f"ring      n=1     avrf_ring_batch_verify {best(lambda: ring_batch_verify(setup, [key.commitment], None, [inst], [rp])):.3f} ms   "
      f"avrf_ring_verify_each {best(lambda: ring_verify_each(setup, [key.commitment], None, [inst], [rp])):.3f} ms", flush=True)
proof = bytes.fromhex(v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] + v["ring_proof"])
io_w = bytes.fromhex(v["h"]) + bytes.fromhex(v["gamma"]); ad = bytes.fromhex(v["ad"])


def wire(each, validate):
    out = (C.c_int32 * 1)()
    rc = L.avrf_ring_vrf_verify(c._h, setup._h, C.c_size_t(1), nat._u8(key.commitment), C.c_size_t(1), nat._u32([0]), nat._u8(io_w),
                                nat._u32([1]), nat._u8(ad), nat._u32([len(ad)]), nat._u8(proof), validate, int(each), out)
    assert rc == 0 and out[0] == 0


for validate in (1, 0):
    wire(1, validate)
    print(f"ring-VRF  n=1     avrf_ring_vrf_verify from wire bytes, validate={validate}: {best(lambda: wire(1, validate)):.3f} ms", flush=True)
pts = bytes.fromhex(v["h"] + v["gamma"] + v["proof_pk_com"] + v["proof_r"] + v["proof_ok"])
for validate in (True, False):
    assert c.points_decompress(pts, validate=validate)[1] == [0] * 5
    print(f"codec     5 pts   avrf_points_decompress validate={int(validate)}: {best(lambda: c.points_decompress(pts, validate=validate)):.3f} ms", flush=True)
