#!/usr/bin/env python3
"""libavrf_probe.so on a GPU box: the multiplier stream at several occupancies, the plain-VALU stream, the clock probe."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "ark_vrf_amd", "libavrf_probe.so"))
out = (C.c_double * 6)()
for name, fn in (("v_mad_u64_u32", L.avrf_probe_mad_stream), ("v_xor_b32", L.avrf_probe_valu_stream)):
    for w in (4, 8, 16, 32):
        rc = fn(0, w, 4096, 3, out)
        print(f"{name:14s} {w:2d} waves/CU: rc {rc}  {out[0]:.2f} T lane-ops/s  shader clock {out[1]:.0f} MHz (runtime reports {out[4]:.0f})  {out[2]:.3f} ms  CUs {int(out[3])}  {out[5]:.1f} lane-ops/clk/CU")
o2 = (C.c_double * 2)()
print("clock probe idle:", L.avrf_probe_clock(0, C.c_double(2000.0), o2), o2[0], "MHz over", o2[1], "us")
o4 = (C.c_double * 4)()
for which, name in ((0, "teu_madd<Bandersnatch>"), (1, "g1u_madd<Bls12381>"), (2, "g1u_madd<Bn254>")):
    for bpc in (2, 3):
        rc = L.avrf_probe_madd_loop(0, which, bpc, 128 if which == 0 else 64, 2, o4)
        print(f"{name:24s} {bpc} workgroups/CU: rc {rc}  {o4[0]:.2f} G additions/s  {o4[1]:.3f} ms")
