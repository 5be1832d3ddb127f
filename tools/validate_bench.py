#!/usr/bin/env python3
"""Validate::Yes at BASELINE configs[1] size (src/lib.rs:410-433,471-494: what every typed point of the reference has passed before
`push`): 65 536 Thin items whose 4 x 65 536 points arrive (a) as x || y and are checked on the device by the pool at validation
level 2 (on the curve + prime-order subgroup), (b) as the 32-byte compressed encodings through avrf_thin_batch_verify_wire
(validate = 1: decompression + subgroup + non-identity), one context.   python tools/validate_bench.py [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402
import bench  # noqa: E402
from ark_vrf_amd import _native as nat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
import oracle as orc  # noqa: E402
b = orc.gen_batch(0, 0, n, threads=16)
out = {}
# (a) the pool, resident batches, validation levels 0 / 1 / 2
for level in (0, 1, 2):
    pool = nat.Pool(0, kind=1, slots=48, lanes=10, threads=6, hash_group=8)
    try:
        pool.set_validation(level)
        pg = nat.PinnedBatch(n, b["ios_xy"], b["io_counts"], b["ads"], b["ad_lens"], pks_xy=b["pks_xy"], proofs=b["proofs"])
        tk = [pool.submit(pg) for _ in range(48)]
        assert all(pool.wait(t) == 0 for t in tk)
        done, mism, sec = pool.cycle(steps_block=96, min_seconds=1.5, expect=0)
        assert mism == 0
        out[f"pool_xy_validation_level_{level}_per_sec"] = done * n / sec
        done, mism, sec = pool.cycle(steps_block=96, min_seconds=1.5, from_host=True, expect=0)
        assert mism == 0
        out[f"pool_xy_validation_level_{level}_from_pinned_host_per_sec"] = done * n / sec
    finally:
        pool.close()
# (b) compressed wire encodings, one call at a time on one context
comp = lambda xy_all: b"".join(orc.point_compress(0, xy_all[64 * i: 64 * i + 64]) for i in range(len(xy_all) // 64))
pks_c = comp(b["pks_xy"]); ios_c = comp(b["ios_xy"])
pr = b["proofs"]
proofs_c = b"".join(orc.point_compress(0, pr[96 * j: 96 * j + 64]) + pr[96 * j + 64: 96 * j + 96] for j in range(n))
# (a') the pool fed with the COMPRESSED bytes (avrf_pool_submit_wire): every step staged again from page-locked wire buffers
for v in (0, 1):
    pool = nat.Pool(0, kind=1, slots=48, lanes=10, threads=6, hash_group=8)
    try:
        pw = nat.PinnedBatch(n, ios_c, b["io_counts"], b["ads"], b["ad_lens"], pks_xy=pks_c, proofs=proofs_c)
        tk = [pool.submit_wire(pw, validate=v) for _ in range(48)]
        assert all(pool.wait(t) == 0 for t in tk)
        done, mism, sec = pool.cycle(steps_block=96, min_seconds=1.5, from_host=True, expect=0)
        assert mism == 0
        out[f"pool_wire_validate{v}_from_pinned_host_per_sec"] = done * n / sec
    finally:
        pool.close()
ctx = nat.Context(0)
L = nat.lib()
args = (ctx._h, C.c_size_t(n), nat._u8(pks_c), nat._u8(ios_c), nat._u32(b["io_counts"]), nat._u8(b["ads"]), nat._u32(b["ad_lens"]), nat._u8(proofs_c))
for v in (0, 1):
    assert L.avrf_thin_batch_verify_wire(*args, v) == 0
    t = time.perf_counter(); reps = 5
    for _ in range(reps):
        assert L.avrf_thin_batch_verify_wire(*args, v) == 0
    out[f"thin_batch_verify_wire_validate{v}_per_sec"] = reps * n / (time.perf_counter() - t)
# the same entry point from several caller threads, one context each (the sequential weight transcript of a batch, 3.5 ms of one
# host core, is what a single caller waits for; independent callers overlap theirs)
from concurrent.futures import ThreadPoolExecutor
for nthreads in (4, 8, 12):
    ctxs = [nat.Context(0) for _ in range(nthreads)]
    argl = [(cx._h,) + args[1:] for cx in ctxs]
    for v in (0, 1):
        def work(i, v=v):
            for _ in range(4):
                assert L.avrf_thin_batch_verify_wire(*argl[i], v) == 0
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(work, range(nthreads)))                       # warm-up
            t = time.perf_counter(); list(ex.map(work, range(nthreads))); dt = time.perf_counter() - t
        out[f"thin_batch_verify_wire_validate{v}_{nthreads}_callers_per_sec"] = nthreads * 4 * n / dt
    for cx in ctxs:
        cx.close()
cp, st = None, None
t = time.perf_counter(); xy, st = ctx.points_decompress(pks_c + ios_c, validate=True); dt = time.perf_counter() - t
out["points_decompress_validate1_points_per_sec"] = 3 * n / dt
t = time.perf_counter(); xy, st = ctx.points_decompress(pks_c + ios_c, validate=False); dt = time.perf_counter() - t
out["points_decompress_validate0_points_per_sec"] = 3 * n / dt
import json
print(json.dumps({k: round(v) for k, v in out.items()}, indent=1))
