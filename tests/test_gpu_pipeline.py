"""The run of a staged batch in three calls (avrf_batch_run_begin / _hash / _end, include/avrf.h): same verdicts as the one-call
form and the oracle (thin::BatchVerifier::verify src/thin.rs:257-325, pedersen::BatchVerifier::verify src/pedersen.rs:341-426),
also with several contexts interleaved on one host thread, and a call out of order is refused."""
import pytest

import oracle as orc
from helpers import IDENTITY_XY, nat_batch

pytestmark = pytest.mark.gpu
BAD_ARG = -2


@pytest.fixture(scope="module")
def nat():
    from ark_vrf_amd import _native as nat
    return nat


def three(c):
    st = c.batch_run_begin()
    if st:
        return st
    st = c.batch_run_hash()
    if st:
        return st
    return c.batch_run_end()


@pytest.mark.parametrize("suite", [0, 1, 7])
def test_three_calls_match_one_call_and_oracle(nat, suite):
    c = nat.Context(suite)
    try:
        for kind, n in ((0, 300), (1, 257)):
            b = orc.gen_batch(suite, kind, n)
            if kind == 1:
                b["pks_xy"] = b""
            stage = c.thin_batch_stage if kind == 0 else c.pedersen_batch_stage
            run = c.thin_batch_run if kind == 0 else c.pedersen_batch_run
            want = orc.thin_batch_verify_xy if kind == 0 else orc.pedersen_batch_verify_xy
            psz = len(b["proofs"]) // n
            assert stage(nat_batch(b)) == 0
            assert three(c) == 0 == run() == want(suite, b)
            assert three(c) == 0                                      # the staged batch can be run again
            pr = bytearray(b["proofs"]); pr[psz * (n // 2) + psz - 1 - 31] ^= 1    # low byte of the last response scalar
            b2 = dict(b); b2["proofs"] = bytes(pr)
            assert stage(nat_batch(b2)) == 0
            assert three(c) == 1 == run() == want(suite, b2)
            if kind == 0:
                pk = bytearray(b["pks_xy"]); pk[64 * 7: 64 * 8] = IDENTITY_XY
                b3 = dict(b); b3["pks_xy"] = bytes(pk)
                if want(suite, b3) == 2:
                    assert stage(nat_batch(b3)) == 0
                    assert c.batch_run_begin() == 0 and c.batch_run_hash() == 2      # InvalidData before any equation
                    assert c.batch_run_end() == BAD_ARG                             # the sequence ended with the error
                    assert three(c) == 2 == run()
    finally:
        c.close()


def test_order_is_enforced_and_empty_batch(nat):
    c = nat.Context(0)
    try:
        assert c.batch_run_begin() == BAD_ARG                                      # nothing staged
        b = orc.gen_batch(0, 0, 64)
        assert c.thin_batch_stage(nat_batch(b)) == 0
        assert c.batch_run_hash() == BAD_ARG and c.batch_run_end() == BAD_ARG
        assert c.batch_run_begin() == 0
        assert c.batch_run_begin() == BAD_ARG and c.batch_run_end() == BAD_ARG
        assert c.thin_batch_stage(nat_batch(b)) == BAD_ARG                         # a run in flight owns the staged buffers
        assert c.thin_batch_run() == BAD_ARG
        assert c.batch_run_hash() == 0
        assert c.batch_run_hash() == BAD_ARG and c.batch_run_begin() == BAD_ARG
        assert c.batch_run_end() == 0
        assert c.thin_batch_run() == 0
        e = orc.gen_batch(0, 0, 0)
        assert c.thin_batch_stage(nat_batch(e)) == 0 and three(c) == 0 == c.thin_batch_run()   # src/thin.rs:262-264
    finally:
        c.close()


def test_interleaved_contexts_on_one_thread(nat):
    """what bench.py's pipelined workers do: hash context d while the kernels of the others run"""
    n_ctx, n = 3, 2000
    ctxs = [nat.Context(0) for _ in range(n_ctx)]
    try:
        want = []
        for i, c in enumerate(ctxs):
            b = orc.gen_batch(0, 0, n - 100 * i)
            if i == 1:
                pr = bytearray(b["proofs"]); pr[96 * 5 + 64] ^= 1; b["proofs"] = bytes(pr)
            want.append(orc.thin_batch_verify_xy(0, b))
            assert c.thin_batch_stage(nat_batch(b)) == 0
        assert want == [0, 1, 0]
        got = [[] for _ in ctxs]
        for c in ctxs:
            assert c.batch_run_begin() == 0
        begun, msm = list(range(n_ctx)), []
        for step in range(4 * n_ctx):
            d = begun.pop(0)
            assert ctxs[d].batch_run_hash() == 0
            msm.append(d)
            if len(msm) > 1 or not begun:
                e = msm.pop(0)
                got[e].append(ctxs[e].batch_run_end())
                assert ctxs[e].batch_run_begin() == 0
                begun.append(e)
        for d in begun:                                                            # drain
            assert ctxs[d].batch_run_hash() == 0
            msm.append(d)
        for e in msm:
            got[e].append(ctxs[e].batch_run_end())
        for i in range(n_ctx):
            assert len(got[i]) >= 4 and set(got[i]) == {want[i]}, (i, got[i])
    finally:
        for c in ctxs:
            c.close()


def test_no_other_call_while_a_run_is_open(nat):
    """avrf_thin_batch_partial between _hash and _end ran its own MSM on the context's workspace, consumed the pending
    chain, and _end then folded NOTHING into the identity = AVRF_OK for a tampered batch.  Every entry point that touches the
    context is refused while a run is open, the run itself is not disturbed, and a finish without an enqueued chain is an error."""
    import ctypes as C
    L = nat.lib()
    c = nat.Context(0)
    try:
        n = 500
        b = orc.gen_batch(0, 0, n)
        pr = bytearray(b["proofs"]); pr[96 * 17 + 64] ^= 1; b["proofs"] = bytes(pr)       # tampered: the verdict must be 1
        assert orc.thin_batch_verify_xy(0, b) == 1
        assert c.thin_batch_stage(nat_batch(b)) == 0
        chal = (C.c_uint8 * (16 * n))()
        assert L.avrf_thin_batch_challenges(c._h, chal) == 0                              # (legal before the run: the old hole)
        seed, out = (C.c_uint8 * 64)(), (C.c_uint8 * 64)()
        one_xy, one_sc = (C.c_uint8 * 64)(*IDENTITY_XY), (C.c_uint8 * 32)()
        st4 = (C.c_int32 * 4)()
        for phase in ("begin", "hash"):
            assert getattr(c, "batch_run_" + phase)() == 0
            assert L.avrf_thin_batch_partial(c._h, seed, C.c_uint64(0), out) == BAD_ARG
            assert L.avrf_thin_batch_challenges(c._h, chal) == BAD_ARG
            assert L.avrf_msm_te(c._h, C.c_size_t(1), one_xy, one_sc, out) == BAD_ARG
            assert L.avrf_msm_te_mont(c._h, C.c_size_t(1), one_xy, one_sc, out) == BAD_ARG
            assert L.avrf_scalar_mul(c._h, C.c_size_t(1), one_sc, one_xy, out) == BAD_ARG
            assert L.avrf_scalar_mul_base(c._h, C.c_size_t(1), one_sc, out) == BAD_ARG
            assert L.avrf_points_compress(c._h, C.c_size_t(1), one_xy, out) == BAD_ARG
            assert L.avrf_points_decompress(c._h, C.c_size_t(1), one_sc, out, 0, st4) == BAD_ARG
            assert c.thin_batch_stage(nat_batch(b)) == BAD_ARG
        assert c.batch_run_end() == 1                                                      # the run was not disturbed: still REJECTED
        assert c.batch_run_end() == BAD_ARG                                                # and it is closed
        # the context is usable again
        assert c.thin_batch_run() == 1
        assert L.avrf_thin_batch_partial(c._h, seed, C.c_uint64(0), out) == 0              # legal again once the run is closed
    finally:
        c.close()
