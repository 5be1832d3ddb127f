"""oracle/pairing_py.py -- slow, generic pairing for the ring verifier oracle (TEST ORACLE).

Restates from the published definitions what the reference gets from arkworks `Pairing`
(ark-ec / ark-bls12-381, third-party; reached via `verifier.verify`, src/ring.rs:242, and
`ring_batch.verify`, src/ring.rs:731): the ate pairing on BLS12-381 with Fp12 represented directly
as Fp[w]/(w^12 - 2 w^6 + 2) (w^6 = 1 + u, u^2 = -1), the M-twist untwisting (x, y) -> (x/w^2, y/w^3),
a Miller loop over |x| = 0xd201000000010000 with affine line functions on E(Fp12), and the plain
final exponentiation f^((p^12 - 1)/r).  Only product-equals-one tests are needed, so the sign of x
is ignored.  Pinned by: e(tau*g1, g2) == e(g1, tau*g2) on the reference's SRS file
and by the reference's ring proofs verifying (tests/test_oracle_ring.py).
"""

# The module is parameterised by one global curve description; `use_curve("bls12_381" | "bn254")` switches.
# BLS12-381: u^2 = -1, xi = 1 + u, w^6 = xi  =>  w^12 - 2 w^6 + 2 = 0; M-twist, untwist (x/w^2, y/w^3);
#            Miller loop over |x| = 0xd201000000010000 (= t - 1).
# BN254:     u^2 = -1, xi = 9 + u, w^6 = xi  =>  w^12 - 18 w^6 + 82 = 0; D-twist, untwist (x w^2, y w^3);
#            plain ate pairing: Miller loop over t - 1 = 6 x^2, x = 4965661367192848881 (no Frobenius end
#            steps needed; any non-degenerate bilinear pairing decides product-equals-one checks).
CURVES = {
    "bls12_381": dict(P=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
                      R=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
                      LOOP=0xd201000000010000, XI0=1, MTWIST=True),
    "bn254": dict(P=21888242871839275222246405745257275088696311157297823662689037894645226208583,
                  R=21888242871839275222246405745257275088548364400416034343698204186575808495617,
                  LOOP=6 * 4965661367192848881 ** 2, XI0=9, MTWIST=False),
}
P = R_ORDER = ATE_LOOP = XI0 = MTWIST = None
MOD_COEFFS = None


def use_curve(name):
    global P, R_ORDER, ATE_LOOP, XI0, MTWIST, MOD_COEFFS
    c = CURVES[name]
    P, R_ORDER, ATE_LOOP, XI0, MTWIST = c["P"], c["R"], c["LOOP"], c["XI0"], c["MTWIST"]
    # (w^6 - XI0)^2 = -1  =>  w^12 = 2 XI0 w^6 - (XI0^2 + 1)
    MOD_COEFFS = [XI0 * XI0 + 1, 0, 0, 0, 0, 0, -2 * XI0, 0, 0, 0, 0, 0]


use_curve("bls12_381")


class F12:
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [x % P for x in c]

    @staticmethod
    def one():
        return F12([1] + [0] * 11)

    @staticmethod
    def zero():
        return F12([0] * 12)

    def __add__(self, o):
        return F12([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return F12([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return F12([-a for a in self.c])

    def __eq__(self, o):
        return self.c == o.c

    def scale(self, k):
        return F12([a * k for a in self.c])

    def __mul__(self, o):
        if isinstance(o, int):
            return self.scale(o)
        b = [0] * 23
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        m6, m0 = -MOD_COEFFS[6], -MOD_COEFFS[0]           # w^12 = m6 w^6 + m0
        for k in range(22, 11, -1):
            t = b[k]
            if t:
                b[k - 6] += m6 * t
                b[k - 12] += m0 * t
        return F12(b[:12])

    def inv(self):
        # extended Euclid over Fp[w]
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], [x % P for x in MOD_COEFFS] + [1]
        deg = lambda p: max([i for i, x in enumerate(p) if x] or [0])
        while deg(low):
            # r = high / low
            dl, dh = deg(low), deg(high)
            r = [0] * 13
            temp = list(high)
            inv_lead = pow(low[dl], -1, P)
            for i in range(dh - dl, -1, -1):
                r[i] = temp[dl + i] * inv_lead % P
                for c in range(dl + 1):
                    temp[c + i] = (temp[c + i] - low[c] * r[i]) % P
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                    new[i + j] = (new[i + j] - low[i] * r[j]) % P
            lm, low, hm, high = nm, new, lm, low
        k = pow(low[0], -1, P)
        return F12([x * k for x in lm[:12]])

    def __truediv__(self, o):
        return self * o.inv()

    def __pow__(self, e):
        out, base = F12.one(), self
        while e:
            if e & 1:
                out = out * base
            base = base * base
            e >>= 1
        return out


W = F12([0, 1] + [0] * 10)


def embed_fp(x):
    return F12([x] + [0] * 11)


def embed_fp2(a, b):
    """a + b u  ->  (a - XI0 b) + b w^6   (u = w^6 - XI0)"""
    return F12([a - XI0 * b] + [0] * 5 + [b] + [0] * 5)


def untwist(q):
    """G2 point ((x0, x1), (y0, y1)) on the sextic twist -> point of E(Fp12)."""
    (x0, x1), (y0, y1) = q
    w2 = W * W
    if MTWIST:
        return (embed_fp2(x0, x1) / w2, embed_fp2(y0, y1) / (w2 * W))
    return (embed_fp2(x0, x1) * w2, embed_fp2(y0, y1) * (w2 * W))


def _double(pt):
    x, y = pt
    m = (x * x).scale(3) / y.scale(2)
    nx = m * m - x.scale(2)
    return (nx, m * (x - nx) - y)


def _add(p1, p2):
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _line(p1, p2, t):
    x1, y1 = p1; x2, y2 = p2; xt, yt = t
    if not (x1 == x2):
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1).scale(3) / y1.scale(2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(q_g2, p_g1):
    """q_g2: ((x0,x1),(y0,y1)) affine on the twist; p_g1: (x, y) affine.  No final exponentiation."""
    if q_g2 is None or p_g1 is None:
        return F12.one()
    Q = untwist(q_g2)
    Pt = (embed_fp(p_g1[0]), embed_fp(p_g1[1]))
    Rr, f = Q, F12.one()
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        f = f * f * _line(Rr, Rr, Pt)
        Rr = _double(Rr)
        if (ATE_LOOP >> i) & 1:
            f = f * _line(Rr, Q, Pt)
            Rr = _add(Rr, Q)
    return f


def final_exp(f):
    return f ** ((P ** 12 - 1) // R_ORDER)


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for pairs [(g1_affine, g2_affine)]"""
    f = F12.one()
    for p1, q2 in pairs:
        f = f * miller_loop(q2, p1)
    return final_exp(f) == F12.one()


def g2_decode_arkworks_uncompressed(b):
    """BN254, arkworks default SW format: x.c0 || x.c1 || y.c0 || y.c1, 32-byte little-endian each, flags in
    the two top bits of the last byte."""
    if b[-1] & 0x40:
        return None
    v = [int.from_bytes(b[32 * i: 32 * i + 32], "little") for i in range(4)]
    v[3] &= (1 << 254) - 1
    return ((v[0], v[1]), (v[2], v[3]))


def g2_decode_zcash_uncompressed(b):
    """192 bytes: x.c1 || x.c0 || y.c1 || y.c0, 48-byte big-endian each."""
    v = [int.from_bytes(b[48 * i: 48 * i + 48], "big") for i in range(4)]
    if b[0] & 0x40:
        return None
    v[0] &= (1 << 381) - 1
    return ((v[1], v[0]), (v[3], v[2]))


# ---- G2 on the twist E'(Fp2): y^2 = x^3 + b', affine, Fp2 elements as (c0, c1) with u^2 = -1.
# Used for `tau * g2` of Kzg::setup (src/ring.rs:359-374 -> w3f-pcs `URS::generate`) and by the GPU pairing parity tests.

def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_inv(a):
    d = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def g2_add(q1, q2):
    """affine addition on the twist (None = infinity); the curve coefficient is not needed (a = 0)."""
    if q1 is None:
        return q2
    if q2 is None:
        return q1
    (x1, y1), (x2, y2) = q1, q2
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        m = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_mul((2, 0), y1)))
    else:
        m = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(m, m), x1), x2)
    return (x3, f2_sub(f2_mul(m, f2_sub(x1, x3)), y1))


def g2_mul(q, k):
    acc = None
    for bit in bin(k)[2:]:
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, q)
    return acc


def g2_encode_arkworks_uncompressed(q):
    """inverse of g2_decode_arkworks_uncompressed (BN254): flag bit 7 of the last byte = y is the 'larger' root
    (Fp2 ordering of ark-ff: compare c1 first, then c0), bit 6 = infinity."""
    if q is None:
        out = bytearray(128); out[-1] |= 0x40
        return bytes(out)
    (x0, x1), (y0, y1) = q
    neg = ((-y0) % P, (-y1) % P)
    larger = (y1, y0) > (neg[1], neg[0])
    out = bytearray(b"".join(v.to_bytes(32, "little") for v in (x0, x1, y0, y1)))
    if larger:
        out[-1] |= 0x80
    return bytes(out)


def g2_encode_zcash_uncompressed(q):
    """inverse of g2_decode_zcash_uncompressed (BLS12-381): x.c1 || x.c0 || y.c1 || y.c0 big-endian."""
    if q is None:
        out = bytearray(192); out[0] |= 0x40
        return bytes(out)
    (x0, x1), (y0, y1) = q
    return b"".join(v.to_bytes(48, "big") for v in (x1, x0, y1, y0))


def _f2_largest(y):
    """y is the larger of {y, -y} with Fp2 ordered by (c1, c0) (ark-ff QuadExtField Ord; zcash 'lexicographically largest')."""
    neg = ((-y[0]) % P, (-y[1]) % P)
    return (y[1], y[0]) > (neg[1], neg[0])


def g2_encode_compressed(q, zcash):
    """serialize_compressed of a G2 point: zcash (BLS12-381) x.c1 || x.c0 big-endian with flag bits in byte 0
    (0x80 compressed, 0x40 infinity, 0x20 largest y); arkworks (BN254) x.c0 || x.c1 little-endian, flags in the last byte
    (0x80 largest y, 0x40 infinity)."""
    n = 48 if zcash else 32
    if q is None:
        out = bytearray(2 * n)
        if zcash:
            out[0] = 0xC0
        else:
            out[-1] = 0x40
        return bytes(out)
    (x0, x1), y = q
    if zcash:
        out = bytearray(x1.to_bytes(n, "big") + x0.to_bytes(n, "big")); out[0] |= 0x80
        if _f2_largest(y):
            out[0] |= 0x20
    else:
        out = bytearray(x0.to_bytes(n, "little") + x1.to_bytes(n, "little"))
        if _f2_largest(y):
            out[-1] |= 0x80
    return bytes(out)
