import json, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from ark_vrf_amd import _native as nat
from ark_vrf_amd.ring import RingSetup, ring_batch_verify, ring_verify_each
import ctypes as C
ctx = nat.Context(0)
g = "tests/golden"
v = json.load(open(f"{g}/bandersnatch_sha-512_ell2_ring.json"))[0]
srs = open(f"{g}/bls12-381-srs-2-11-uncompressed-zcash.bin", "rb").read()
setup = RingSetup(ctx, srs, 8)
yb, st = ctx.points_decompress(bytes.fromhex(v["proof_pk_com"]))
com, proof = bytes.fromhex(v["ring_pks_com"]), bytes.fromhex(v["ring_proof"])
for _ in range(3): assert ring_batch_verify(setup, [com], None, [yb], [proof]) == 0
t = time.perf_counter(); n = 50
for _ in range(n): ring_batch_verify(setup, [com], None, [yb], [proof])
print("ring verify (ring half, n=1): %.3f ms" % ((time.perf_counter() - t) / n * 1e3))
# complete: pedersen verify + ring verify via one call
ios = bytes.fromhex(v["h"] + v["gamma"]); ad = bytes.fromhex(v["ad"])
full = bytes.fromhex(v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] + v["ring_proof"])
out = (C.c_int32 * 1)()
L = nat.lib()
def full_verify(each):
    return L.avrf_ring_vrf_verify(ctx._h, setup._h, C.c_size_t(1), nat._u8(com), C.c_size_t(1), None, nat._u8(ios), nat._u32([1]), nat._u8(ad), nat._u32([len(ad)]), nat._u8(full), 1, each, out)
for _ in range(3): assert full_verify(0) == 0
t = time.perf_counter()
for _ in range(n): full_verify(0)
print("complete ring-VRF verify from wire bytes, n=1: %.3f ms" % ((time.perf_counter() - t) / n * 1e3))
os.environ["AVRF_RING_TRACE"] = "1"
