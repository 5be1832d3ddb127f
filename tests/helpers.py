"""Shared test helpers: deterministic inputs built with the CPU oracle (test infrastructure)."""
import hashlib
import random

import oracle as orc

R_ORDER = {
    orc.BANDERSNATCH: 0x1cfb69d4ca675f520cce760202687600ff8f87007419047174fd06b52876e7e1,
    orc.BABYJUBJUB: 2736030358979909402780800718157159386076813972158567259200215660948447373041,
}
IDENTITY_XY = bytes(32) + (1).to_bytes(32, "little")


def rand_scalar(rng, suite, bits=None):
    r = R_ORDER[suite]
    v = rng.getrandbits(bits) if bits else rng.randrange(r)
    return (v % r).to_bytes(32, "little")


def rand_points_xy(rng, suite, n):
    """n pseudo-random prime-order-subgroup points as xy bytes (k_i * G via the oracle)."""
    g = orc.suite_point(suite, 0)
    out = []
    for _ in range(n):
        k = rand_scalar(rng, suite)
        st, xy = orc.point_decompress(suite, orc.smul(suite, k, g))
        assert st == 0
        out.append(xy)
    return out


def xy(suite, comp):
    st, out = orc.point_decompress(suite, comp)
    assert st == 0
    return out
