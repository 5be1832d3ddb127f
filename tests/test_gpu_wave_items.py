"""Few items, one I/O pair each: avrf_thin_verify / avrf_thin_prove spread an item over 32 lanes (vrf_single.hip "few items",
k_thin_verify_wave / k_thin_prove_wave) instead of one lane.  Same group elements, so the proofs must be byte-identical to the
oracle's (thin::Prover::prove, src/thin.rs:111-135) and the verdicts those of thin::Verifier::verify (src/thin.rs:137-165), and
both must agree with the lane-per-item kernels (taken above AVRF_WAVE_ITEMS_MAX items or with more than one pair)."""
import pytest

import oracle as orc
from helpers import IDENTITY_XY, nat_batch

pytestmark = pytest.mark.gpu
TE_SUITES = [0, 1, 2, 3, 4, 5, 6]           # every suite but the short-Weierstrass one (7), which keeps the lane-per-item kernel


@pytest.fixture(scope="module")
def nat():
    from ark_vrf_amd import _native as nat
    return nat


@pytest.mark.parametrize("suite", TE_SUITES + [7])
@pytest.mark.parametrize("n", [1, 2, 3, 33])
def test_prove_and_verify_small_n_match_oracle(nat, suite, n):
    b = orc.gen_batch(suite, 0, n, start=7 * n)
    c = nat.Context(suite)
    try:
        assert c.thin_prove(nat_batch(b, with_sks=True, with_proofs=False)) == b["proofs"]        # byte for byte
        assert c.thin_verify(nat_batch(b)) == [0] * n
        # tampered: response scalar / nonce point / additional data / public key -> exactly those items fail
        pr = bytearray(b["proofs"]); pr[96 * (n - 1) + 64] ^= 1
        want = [0] * n; want[n - 1] = 1
        assert c.thin_verify(nat_batch(dict(b, proofs=bytes(pr)))) == want
        pr = bytearray(b["proofs"]); pr[0:64] = b["pks_xy"][0:64]                                   # R := pk of item 0 (a valid point, wrong value)
        want = [0] * n; want[0] = 1
        assert c.thin_verify(nat_batch(dict(b, proofs=bytes(pr)))) == want
        pk = bytearray(b["pks_xy"]); pk[0:64] = bytes(64) if suite == 7 else IDENTITY_XY
        want = [0] * n; want[0] = 2                                                                 # identity public key: InvalidData (thin.rs:140-142)
        assert c.thin_verify(nat_batch(dict(b, pks_xy=bytes(pk)))) == want
        s_bad = bytearray(b["proofs"]); s_bad[96 * 0 + 64: 96 * 0 + 96] = b"\xff" * 32              # s >= r
        assert c.thin_verify(nat_batch(dict(b, proofs=bytes(s_bad))))[0] == 2
    finally:
        c.close()


@pytest.mark.parametrize("suite", [0, 1])
def test_wave_and_lane_kernels_agree(nat, suite):
    """2048 items take the 32-lanes-per-item kernels, 2049 the lane-per-item ones: same proofs, same verdicts"""
    n = 2049
    b = orc.gen_batch(suite, 0, n)
    c = nat.Context(suite)
    try:
        big = c.thin_prove(nat_batch(b, with_sks=True, with_proofs=False))
        assert big == b["proofs"]
        sub = lambda k, d: {**d, "n": k, "sks": d["sks"][: 32 * k], "pks_xy": d["pks_xy"][: 64 * k], "ios_xy": d["ios_xy"][: 128 * k], "io_counts": [1] * k,
                            "ads": d["ads"][: sum(d["ad_lens"][:k])], "ad_lens": d["ad_lens"][:k], "proofs": d["proofs"][: 96 * k]}
        small = sub(2048, b)
        assert c.thin_prove(nat_batch(small, with_sks=True, with_proofs=False)) == big[: 96 * 2048]
        pr = bytearray(b["proofs"])
        for j in (0, 5, 100, 2047, 2048):
            pr[96 * j + 70] ^= 4
        want = [1 if j in (0, 5, 100, 2047, 2048) else 0 for j in range(n)]
        assert c.thin_verify(nat_batch(dict(b, proofs=bytes(pr)))) == want
        assert c.thin_verify(nat_batch(sub(2048, dict(b, proofs=bytes(pr))))) == want[:2048]
    finally:
        c.close()


def test_single_item_latency(nat):
    """one verification / one proof through the C ABI (reference: 188 / 182 us on a CPU core, benches/SUMMARY.md:53-54); the
    lane-per-item kernel took 2.1-2.4 ms"""
    import time
    b = orc.gen_batch(0, 0, 1)
    c = nat.Context(0)
    try:
        vb, pb = nat_batch(b), nat_batch(b, with_sks=True, with_proofs=False)
        for _ in range(3):
            assert c.thin_verify(vb) == [0] and c.thin_prove(pb) == b["proofs"]
        tv = min(_t(lambda: c.thin_verify(vb)) for _ in range(20))
        tp = min(_t(lambda: c.thin_prove(pb)) for _ in range(20))
        print(f"\nthin verify n=1: {tv * 1e3:.3f} ms, thin prove n=1: {tp * 1e3:.3f} ms")
        assert tv < 1.5e-3 and tp < 1.5e-3
    finally:
        c.close()


def _t(fn):
    import time
    t0 = time.perf_counter(); fn(); return time.perf_counter() - t0


@pytest.mark.parametrize("suite", TE_SUITES + [7])
@pytest.mark.parametrize("n", [1, 2, 5, 33])
def test_pedersen_small_n_matches_oracle(nat, suite, n):
    """pedersen::Prover::prove / Verifier::verify (src/pedersen.rs:136-249) on the few-items kernels: proofs and blindings byte for
    byte, verdicts per item"""
    b = orc.gen_batch(suite, 1, n, start=11 * n)
    c = nat.Context(suite)
    try:
        proofs, blind = c.pedersen_prove(nat_batch(b, with_sks=True, with_proofs=False))
        assert proofs == b["proofs"]
        vb = dict(b, pks_xy=b"")
        assert c.pedersen_verify(nat_batch(vb)) == [0] * n
        for off, want_st in ((192, 1), (224, 1), (70, 1), (130, 1)):       # s, sb, R.x, Ok.x of the last item
            pr = bytearray(b["proofs"]); pr[256 * (n - 1) + off] ^= 1
            got = c.pedersen_verify(nat_batch(dict(vb, proofs=bytes(pr))))
            assert got[: n - 1] == [0] * (n - 1) and got[n - 1] in (want_st, 2), (off, got)      # (a flipped coordinate may leave the field: InvalidData)
        pr = bytearray(b["proofs"]); pr[0:64] = bytes(64) if suite == 7 else IDENTITY_XY          # Yb = identity: InvalidData (pedersen.rs:204-206)
        assert c.pedersen_verify(nat_batch(dict(vb, proofs=bytes(pr))))[0] == 2
    finally:
        c.close()
