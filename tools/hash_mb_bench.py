#!/usr/bin/env python3
"""Times the 8-way multi-buffer weight hash against the scalar chain on this host (no GPU needed)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AVRF_TRACE_HASH"] = "1"
from ark_vrf_amd import _native as nat
L = nat.lib(); n = 65536
c = os.urandom(16 * n); r = os.urandom(32 * n)
for k in (8, 4, 2):
    cb = [C.create_string_buffer(c, len(c)) for _ in range(k)]; rb = [C.create_string_buffer(r, len(r)) for _ in range(k)]
    cp = (C.c_void_p * k)(*[C.cast(b, C.c_void_p) for b in cb]); rp = (C.c_void_p * k)(*[C.cast(b, C.c_void_p) for b in rb]); ns = (C.c_size_t * k)(*[n] * k)
    o8 = (C.c_uint8 * (64 * k))()
    assert L.avrf_batch_weight_seeds_x8(0, 0, k, ns, cp, rp, o8) == 0
