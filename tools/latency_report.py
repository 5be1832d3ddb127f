#!/usr/bin/env python3
"""Single-call latencies through the C ABI on one GPU (DESIGN.md section 5 "One item"; references: benches/SUMMARY.md:53-88):
   python tools/latency_report.py
Thin / Pedersen verify and prove with n = 1, 8, 64; BatchVerifier calls of 1, 8, 64, 1024 items; one ring verification
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
for n in (1, 8, 64, 1024):
    vb, pb, pvb, raw = items(n)
    f = lambda: L.avrf_thin_batch_verify(c._h, C.c_size_t(vb.n), vb.pks_xy, vb.ios_xy, vb.io_counts, vb.ads, vb.ad_lens, vb.proofs)
    g = lambda: L.avrf_pedersen_batch_verify(c._h, C.c_size_t(pvb.n), pvb.ios_xy, pvb.io_counts, pvb.ads, pvb.ad_lens, pvb.proofs)
    assert f() == 0 and g() == 0
    print(f"batch     n={n:<5} thin BatchVerifier {best(f, 20):.3f} ms   pedersen BatchVerifier {best(g, 20):.3f} ms", flush=True)

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
