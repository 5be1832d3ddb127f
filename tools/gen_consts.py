#!/usr/bin/env python3
"""Generates ark_vrf_amd/csrc/consts_gen.h: Montgomery constants (R = 2^256) for the four
256-bit prime fields on the path and the twisted-Edwards suite constants, as 32-bit limb
arrays (device) -- the same arrays are reassembled into 64-bit limbs on the host.

Numbers come from SURVEY.md Appendix B / the reference's suite files
(src/suites/bandersnatch.rs:13-14,72-104, src/suites/baby_jubjub.rs:12-13,63-94).
Run:  python tools/gen_consts.py
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark_vrf_amd", "csrc", "consts_gen.h")
R = 1 << 256


def limbs(x):
    return ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(8))


def field(name, p):
    ninv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    s = []
    s.append(f"struct {name} {{")
    s.append(f"  static constexpr uint32_t P[8] = {{{limbs(p)}}};")
    s.append(f"  static constexpr uint32_t ONE[8] = {{{limbs(R % p)}}};   /* R mod p */")
    s.append(f"  static constexpr uint32_t R2[8] = {{{limbs(R * R % p)}}};    /* R^2 mod p */")
    s.append(f"  static constexpr uint32_t HALF[8] = {{{limbs((p - 1) // 2)}}};  /* (p-1)/2, plain */")
    s.append(f"  static constexpr uint32_t NINV = 0x{ninv:08x}u;  /* -p^-1 mod 2^32 */")
    s.append(f"  static constexpr int BITS = {p.bit_length()};")
    s.append(f"  static constexpr bool FULL = {'true' if p.bit_length() == 256 else 'false'};  /* top bit of the modulus set: sums and Montgomery products carry into bit 256 (fp256.h) */")
    # Tonelli-Shanks data
    t, tw = p - 1, 0
    while t % 2 == 0:
        t //= 2
        tw += 1
    g = 2
    while pow(g, (p - 1) // 2, p) != p - 1:
        g += 1
    s.append(f"  static constexpr int TWO_ADICITY = {tw};")
    s.append(f"  static constexpr uint32_t T_MINUS1_HALF[8] = {{{limbs((t - 1) // 2)}}};")
    s.append(f"  static constexpr uint32_t ROOT[8] = {{{limbs(pow(g, t, p) * R % p)}}};  /* g^t, Montgomery */")
    s.append(f"  static constexpr uint32_t PM2[8] = {{{limbs(p - 2)}}};  /* p-2 (inversion exponent) */")
    if name.startswith("Fq"):
        # Branch-free square roots (fp256.h fp_sqrt_ratio_nf): with p - 1 = 2^s t and g = ROOT of order 2^s, a^t = g^e for every
        # a != 0; e is read off in windows of SQRT_W = 8 bits from the low end (Pohlig-Hellman in the 2-group): the window's value j
        # is LOOKED UP from d = c^(2^(s - 8i - w)) = h^(j 2^(hw - w)), h = g^(2^(s - hw)) of order 2^hw (hw = min(8, s)) -- the low word
        # of d's canonical Montgomery form, multiplied by SQRT_HMUL, indexes SQRT_HIDX with its top SQRT_HBITS bits (a perfect hash of
        # the 2^hw roots of unity, searched below) -- then stripped by SQRT_G[i][j] = g^(-j 256^i); SQRT_GH[i][j] = g^(-j 256^i / 2)
        # accumulates g^(-e/2) (e odd <=> a is a non-residue: window 0 shows it).  Windows of 8 bits instead of 4: 48 squarings
        # instead of 112 on a field with s = 32, no compare loops.
        W8 = 8
        hw = min(W8, tw)
        steps = (tw + W8 - 1) // W8
        gg = pow(g, t, p)
        ginv = pow(gg, -1, p)
        h = pow(gg, 1 << (tw - hw), p)                                        # order 2^hw
        roots = [pow(h, j, p) * R % p for j in range(1 << hw)]
        lows = [r & 0xFFFFFFFF for r in roots]
        assert len(set(lows)) == len(lows)
        hbits = hw + 4
        import random
        rng = random.Random(p & 0xFFFF)
        while True:
            mul = rng.randrange(1 << 32) | 1
            idx = [((lo * mul) & 0xFFFFFFFF) >> (32 - hbits) for lo in lows]
            if len(set(idx)) == len(idx):
                break
        tabh = [0] * (1 << hbits)
        for j, k in enumerate(idx):
            tabh[k] = j
        s.append(f"  static constexpr int SQRT_W = {W8};")
        s.append(f"  static constexpr int SQRT_HW = {hw};")
        s.append(f"  static constexpr int SQRT_STEPS = {steps};")
        s.append(f"  static constexpr int SQRT_HBITS = {hbits};")
        s.append(f"  static constexpr uint32_t SQRT_HMUL = 0x{mul:08x}u;")
        s.append(f"  static constexpr uint8_t SQRT_HIDX[{1 << hbits}] = {{" + ",\n    ".join(", ".join(str(x) for x in tabh[k:k + 64]) for k in range(0, 1 << hbits, 64)) + "};")
        rows_g, rows_gh = [], []
        for i in range(steps):
            rg, rgh = [], []
            for j in range(1 << hw):
                e = j << (W8 * i)
                rg.append("{" + limbs(pow(ginv, e, p) * R % p) + "}")
                rgh.append("{" + limbs((pow(ginv, e // 2, p) if e % 2 == 0 else 0) * R % p) + "}")
            rows_g.append("{" + ",\n     ".join(rg) + "}"); rows_gh.append("{" + ",\n     ".join(rgh) + "}")
        s.append(f"  static constexpr uint32_t SQRT_G[{steps}][{1 << hw}][8] = {{" + ",\n    ".join(rows_g) + "};")
        s.append(f"  static constexpr uint32_t SQRT_GH[{steps}][{1 << hw}][8] = {{" + ",\n    ".join(rows_gh) + "};")
    s.append("};")
    return "\n".join(s)


def mont(x, p):
    return x * R % p


def field_n(name, p, nl):
    """generic N-limb field (N x u32), Montgomery R = 2^(32 N)"""
    Rn = 1 << (32 * nl)
    lim = lambda x: ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(nl))
    ninv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    s = [f"struct {name} {{", f"  static constexpr int N = {nl};",
         f"  static constexpr uint32_t P[{nl}] = {{{lim(p)}}};",
         f"  static constexpr uint32_t ONE[{nl}] = {{{lim(Rn % p)}}};",
         f"  static constexpr uint32_t R2[{nl}] = {{{lim(Rn * Rn % p)}}};",
         f"  static constexpr uint32_t HALF[{nl}] = {{{lim((p - 1) // 2)}}};",
         f"  static constexpr uint32_t PM2[{nl}] = {{{lim(p - 2)}}};",
         f"  static constexpr uint32_t NINV = 0x{ninv:08x}u;",
         f"  static constexpr int BITS = {p.bit_length()};",
         f"  static constexpr bool FULL = {'true' if p.bit_length() == 32 * nl else 'false'};", "};"]
    return "\n".join(s)


def suite(name, sid, sid_str, fq, fr, q, r, a_kind, d, pts, cof, ell2=None, glv=None, sw=None, shake=False, sha256=False):
    sid_bytes = ", ".join(str(b) for b in sid_str.encode())
    s = [f"struct {name} {{", f"  using Fq = {fq}; using Fr = {fr};",
         f"  static constexpr int SUITE_ID_LEN = {len(sid_str)};",
         f"  static constexpr uint8_t SUITE_ID[{len(sid_str)}] = {{{sid_bytes}}};  /* {sid_str} */",
         f"  static constexpr int ID = {sid};",
         f"  static constexpr int A_KIND = {a_kind};  /* 0: a = 1, 1: a = -5, 2: a = -1 */",
         f"  static constexpr int COFACTOR = {cof};",
         f"  static constexpr uint32_t D[8] = {{{limbs(mont(d, q))}}};"]
    for nm, (x, y) in pts.items():
        s.append(f"  static constexpr uint32_t {nm}_X[8] = {{{limbs(mont(x, q))}}};")
        s.append(f"  static constexpr uint32_t {nm}_Y[8] = {{{limbs(mont(y, q))}}};")
        s.append(f"  static constexpr uint32_t {nm}_K[8] = {{{limbs(mont(d * x * y % q, q))}}};  /* d*x*y */")
        comp = y | ((1 << 255) if x > (q - 1) // 2 else 0)
        s.append(f"  static constexpr uint32_t {nm}_C[8] = {{{limbs(comp)}}};  /* ark-serialize compressed encoding, LE words */")
    gx, gy = pts["G"]
    if sw:                                    # te_to_sw(G): 33-byte compressed SW form
        u = (1 + gy) * pow(1 - gy, -1, q) % q
        v = (1 + gy) * pow(gx * (1 - gy), -1, q) % q
        xs, ys = sw["MONT_BINV"] * (u + sw["MONT_A3"]) % q, sw["MONT_BINV"] * v % q
        assert (ys * ys - xs ** 3 - sw["SW_A"] * xs - sw["SW_B"]) % q == 0
        enc = xs.to_bytes(32, "little") + bytes([0x80 if ys > (q - 1) // 2 else 0])
    else:
        enc = (gy | ((1 << 255) if gx > (q - 1) // 2 else 0)).to_bytes(32, "little")
    s.append(f"  static constexpr uint8_t G_ENC[{len(enc)}] = {{{', '.join(str(b) for b in enc)}}};  /* serialize_compressed(generator) */")
    # hash-to-curve (src/utils/hash_to_curve.rs): Elligator2 over the Montgomery model (J, K), Z = 5; or try-and-increment
    s.append(f"  static constexpr int H2C_ELL2 = {1 if ell2 else 0};")
    j, k = ell2 if ell2 else (0, 1)
    kinv = pow(k, -1, q)
    s.append(f"  static constexpr uint32_t ELL2_JK[8] = {{{limbs(mont(j * kinv % q, q))}}};     /* J / K */")
    s.append(f"  static constexpr uint32_t ELL2_K[8] = {{{limbs(mont(k, q))}}};")
    s.append(f"  static constexpr uint32_t ELL2_KINV2[8] = {{{limbs(mont(kinv * kinv % q, q))}}};  /* 1 / K^2 */")
    s.append(f"  static constexpr bool XOF_SHAKE = {'true' if shake else 'false'};  /* Suite::Transcript = Shake128Transcript instead of HashTranscript<Sha512> */")
    s.append(f"  static constexpr bool TR_SHA256 = {'true' if sha256 else 'false'};  /* Suite::Transcript = HashTranscript<Sha256> */")
    s.append(f"  static constexpr bool HOST_WEIGHTS = {'true' if (shake or sha256) else 'false'};  /* batch verifiers: the host squeezes the weight stream */")
    # short-Weierstrass presentation (src/suites/bandersnatch_sw.rs, src/utils/te_sw_map.rs): serialised points are 33-byte SW
    # forms; arithmetic stays twisted-Edwards through the maps (x, y) -> (B x - A/3, B y) -> (u / v, (u - 1) / (u + 1))
    s.append(f"  static constexpr bool SW_CODEC = {'true' if sw else 'false'};")
    s.append("  static constexpr bool SW_NATIVE = false;  /* a twisted-Edwards curve (possibly presented as short Weierstrass) */")
    s.append(f"  static constexpr int POINT_LEN = {33 if sw else 32};  /* serialize_compressed size of the suite's Affine */")
    if sw:
        for nm in ("MONT_B", "MONT_A3", "MONT_BINV", "SW_A", "SW_B"):
            s.append(f"  static constexpr uint32_t {nm}[8] = {{{limbs(mont(sw[nm], q))}}};")
    # GLV (per-item scalar multiplications): k = k1 + k2 * lambda (mod r), |k1|, |k2| < 2^127, psi = [lambda] as a rational map
    s.append(f"  static constexpr bool HAS_GLV = {'true' if glv else 'false'};")
    if glv:
        lw = lambda v, n: ", ".join(f"0x{(v >> (32 * i)) & 0xffffffff:08x}u" for i in range(n))
        for nm, v, n in (("GLV_G1", glv["g1"], 4), ("GLV_G2", glv["g2"], 5), ("GLV_A1", glv["a1"], 4), ("GLV_A2", glv["a2"], 4),
                         ("GLV_B1N", glv["b1n"], 4), ("GLV_B2", glv["b2"], 4)):
            assert 0 <= v < 1 << (32 * n)
            s.append(f"  static constexpr uint32_t {nm}[{n}] = {{{lw(v, n)}}};")
        s.append(f"  static constexpr uint32_t ENDO_B[8] = {{{limbs(mont(glv['eb'], q))}}};  /* psi(x, y) = (c (1 - y^2) / (x y), b (y^2 + b) / (y^2 - b)) */")
        s.append(f"  static constexpr uint32_t ENDO_C[8] = {{{limbs(mont(glv['ec'], q))}}};")
    # Subgroup membership by 2-descent (vrf_single.hip te_in_subgroup): when E(Fq) = Z2 x Z2 x Zr (cofactor 4, full rational
    # 2-torsion) the prime-order subgroup is exactly 2 E(Fq), and on the Montgomery model B v^2 = u (u - alpha)(u - beta),
    # u = (1 + y) / (1 - y), a point is a double iff B u, B (u - alpha), B (u - beta) are squares -- two Legendre symbols in y
    # alone: chi(1 - y^2) = chi((c0 + c1 y)(1 - y)) = chi(B), c0 = 1 - alpha, c1 = 1 + alpha -- instead of the 253-bit r P.
    td = two_descent(q, r, a_kind, d, pts["G"]) if cof == 4 else None
    s.append(f"  static constexpr bool HAS_2DESCENT = {'true' if td else 'false'};")
    if td:
        s.append(f"  static constexpr uint32_t TD_C0[8] = {{{limbs(mont(td['c0'], q))}}};  /* 1 - alpha */")
        s.append(f"  static constexpr uint32_t TD_C1[8] = {{{limbs(mont(td['c1'], q))}}};  /* 1 + alpha */")
        s.append(f"  static constexpr int TD_WANT = {td['want']};  /* chi(B), B = 4 / (a - d) */")
    s.append("};")
    return "\n".join(s)


def legendre(x, q):
    x %= q
    return 0 if x == 0 else (1 if pow(x, (q - 1) // 2, q) == 1 else -1)


def sqrt_mod(x, q):
    """Tonelli-Shanks; None for a non-residue"""
    x %= q
    if x == 0:
        return 0
    if legendre(x, q) != 1:
        return None
    s, t = 0, q - 1
    while t % 2 == 0:
        t //= 2; s += 1
    z = 2
    while legendre(z, q) != -1:
        z += 1
    c, R, tt, m = pow(z, t, q), pow(x, (t + 1) // 2, q), pow(x, t, q), s
    while tt != 1:
        i, t2 = 0, tt
        while t2 != 1:
            t2 = t2 * t2 % q; i += 1
        b = pow(c, 1 << (m - i - 1), q)
        R, c = R * b % q, b * b % q
        tt, m = tt * c % q, i
    return R


def two_descent(q, r, a_kind, d, g):
    """constants of the 2-descent subgroup test, or None when the 2-torsion is not fully rational; checked against r P on
    points of every coset (k G, k G + (0, -1), and random curve points, which fall into all four)"""
    import random
    a = {0: 1, 1: q - 5, 2: q - 1}[a_kind]
    A, B = 2 * (a + d) * pow(a - d, -1, q) % q, 4 * pow(a - d, -1, q) % q
    sd = sqrt_mod(A * A - 4, q)
    if sd is None:
        return None
    alpha = (-A + sd) * pow(2, -1, q) % q
    assert (alpha * alpha + A * alpha + 1) % q == 0
    c0, c1, want = (1 - alpha) % q, (1 + alpha) % q, legendre(B, q)

    def add(p1, p2):
        x1, y1 = p1; x2, y2 = p2
        t = d * x1 * x2 * y1 * y2 % q
        return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, q) % q, (y1 * y2 - a * x1 * x2) * pow(1 - t, -1, q) % q)

    def in_subgroup(p):
        acc = (0, 1)
        try:
            for bit in bin(r)[2:]:
                acc = add(acc, acc)
                if bit == "1":
                    acc = add(acc, p)
        except ValueError:                     # the sum left the affine chart: a point at infinity (2-torsion), not the identity
            return False
        return acc == (0, 1)

    def test(p):
        y = p[1]
        return legendre(1 - y * y, q) == want and legendre((c0 + c1 * y) * (1 - y), q) == want
    rng = random.Random(7)
    pts, seen = [], set()
    for _ in range(6):
        P = te_affine_mul(q, a, d, g, rng.randrange(1, r))
        pts += [P, ((-P[0]) % q, (-P[1]) % q)]             # P and P + (0, -1)
    while len(pts) < 60:
        y = rng.randrange(q)
        x = sqrt_mod((1 - y * y) * pow(a - d * y * y, -1, q), q)
        if x is not None:
            pts.append((x, y))
    for P in pts:
        ok = in_subgroup(P)
        seen.add(ok)
        assert test(P) == ok, "2-descent criterion disagrees with r P"
    assert seen == {True, False}
    assert not test((0, q - 1))                # the 2-torsion point (0, -1): u = 0
    return {"c0": c0, "c1": c1, "want": want}


def suite_sw_native(name, sid, sid_str, fq, fr, q, a, b, pts, sha256=True):
    """a genuinely short-Weierstrass suite (src/suites/secp256r1.rs:49-70): y^2 = x^3 + a x + b with a = -3, cofactor 1.  The
    kernels are the twisted-Edwards suites' kernels: te.h gives te_ext / te_pre / te_aff a second meaning for S::SW_NATIVE (XYZZ
    coordinates, affine (x, y), (0, 0) = infinity), sw_map.h's 33-byte codec becomes the identity map."""
    assert a % q == q - 3
    sid_bytes = ", ".join(str(c) for c in sid_str.encode())
    zero = limbs(0)
    s = [f"struct {name} {{", f"  using Fq = {fq}; using Fr = {fr};",
         f"  static constexpr int SUITE_ID_LEN = {len(sid_str)};",
         f"  static constexpr uint8_t SUITE_ID[{len(sid_str)}] = {{{sid_bytes}}};  /* {sid_str} */",
         f"  static constexpr int ID = {sid};",
         "  static constexpr int A_KIND = 0;  /* unused (twisted-Edwards coefficient) */",
         "  static constexpr int COFACTOR = 1;",
         f"  static constexpr uint32_t D[8] = {{{zero}}};  /* unused */"]
    for nm, (x, y) in pts.items():
        assert (y * y - x ** 3 - a * x - b) % q == 0
        s.append(f"  static constexpr uint32_t {nm}_X[8] = {{{limbs(mont(x, q))}}};")
        s.append(f"  static constexpr uint32_t {nm}_Y[8] = {{{limbs(mont(y, q))}}};")
        s.append(f"  static constexpr uint32_t {nm}_K[8] = {{{zero}}};  /* unused */")
    gx, gy = pts["G"]
    enc = gx.to_bytes(32, "little") + bytes([0x80 if gy > (q - 1) // 2 else 0])
    s.append(f"  static constexpr uint8_t G_ENC[33] = {{{', '.join(str(c) for c in enc)}}};  /* serialize_compressed(generator) */")
    s.append("  static constexpr int H2C_ELL2 = 0;")
    for nm in ("ELL2_JK", "ELL2_K", "ELL2_KINV2"):
        s.append(f"  static constexpr uint32_t {nm}[8] = {{{zero}}};  /* unused */")
    s.append("  static constexpr bool XOF_SHAKE = false;")
    s.append(f"  static constexpr bool TR_SHA256 = {'true' if sha256 else 'false'};  /* Suite::Transcript = HashTranscript<Sha256> */")
    s.append(f"  static constexpr bool HOST_WEIGHTS = {'true' if sha256 else 'false'};")
    s.append("  static constexpr bool SW_CODEC = true;   /* serialised points: 33-byte SWAffine form */")
    s.append("  static constexpr bool SW_NATIVE = true;  /* ... of the curve's own points: no twisted-Edwards model behind it */")
    s.append("  static constexpr int POINT_LEN = 33;")
    for nm in ("MONT_B", "MONT_A3", "MONT_BINV"):
        s.append(f"  static constexpr uint32_t {nm}[8] = {{{zero}}};  /* unused */")
    s.append(f"  static constexpr uint32_t SW_A[8] = {{{limbs(mont(a % q, q))}}};")
    s.append(f"  static constexpr uint32_t SW_B[8] = {{{limbs(mont(b, q))}}};")
    s.append("  static constexpr bool HAS_GLV = false;")
    s.append("  static constexpr bool HAS_2DESCENT = false;")
    s.append("};")
    return "\n".join(s)


def glv_bandersnatch(q, r):
    """Lattice basis of {(a, b): a + b lambda = 0 mod r} for lambda = sqrt(-2) mod r (the eigenvalue of the degree-2 endomorphism
    psi of Bandersnatch, Masson-Sanso-Zhang 2021), the rounding multipliers g_i = round(2^256 b_i' / r) of the decomposition
    c1 = (k g1) >> 256, c2 = (k g2) >> 256, k1 = k - c1 a1 - c2 a2, k2 = c1 |b1| - c2 b2, and the two constants of psi in
    twisted-Edwards coordinates.  Everything is re-derived here and checked: lambda^2 = -2, psi(G) = [lambda] G (affine
    arithmetic below), |k1|, |k2| < 2^127 on random scalars."""
    import math
    import random
    lam = 0x13b4f3dc4a39a493edf849562b38c72bcfc49db970a5056ed13d21408783df05
    assert (lam * lam + 2) % r == 0
    r0, r1, t0, t1, rows = r, lam, 0, 1, []
    while r1:
        qq = r0 // r1; r0, r1 = r1, r0 - qq * r1; t0, t1 = t1, t0 - qq * t1
        rows.append((r0, -t0))
    sq = math.isqrt(r)
    l = next(i for i in range(len(rows) - 1) if rows[i][0] >= sq > rows[i + 1][0])
    v1, c0, c2 = rows[l + 1], rows[l], rows[l + 2]
    v2 = c0 if c0[0] ** 2 + c0[1] ** 2 <= c2[0] ** 2 + c2[1] ** 2 else c2
    (a1, b1), (a2, b2) = v1, v2
    if a1 * b2 - a2 * b1 < 0:
        (a1, b1), (a2, b2) = (a2, b2), (a1, b1)
    assert a1 * b2 - a2 * b1 == r and (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
    assert a1 > 0 and a2 > 0 and b1 < 0 and b2 > 0
    g1, g2 = (b2 << 256) // r, (-b1 << 256) // r
    rng = random.Random(7)
    for i in range(20000):
        k = rng.randrange(r) if i > 5 else [0, 1, r - 1, r - 2, lam, r // 2][i]
        c1, c2_ = (k * g1 + (1 << 255)) >> 256, (k * g2 + (1 << 255)) >> 256
        k1, k2 = k - c1 * a1 - c2_ * a2, c1 * (-b1) - c2_ * b2
        assert (k1 + k2 * lam - k) % r == 0 and abs(k1) < 1 << 127 and abs(k2) < 1 << 127
    eb = 0x52c9f28b828426a561f00d3a63511a882ea712770d9af4d6ee0f014d172510b4
    ec = 0x6cc624cf865457c3a97c6efd6c17d1078456abcfff36f4e9515c806cdf650b3d
    return dict(lam=lam, a1=a1, a2=a2, b1n=-b1, b2=b2, g1=g1, g2=g2, eb=eb, ec=ec)


def te_affine_mul(q, a, d, pt, k):
    """k * pt on a x^2 + y^2 = 1 + d x^2 y^2 (affine, unified addition) -- generator-side check of psi only"""
    def add(p1, p2):
        x1, y1 = p1; x2, y2 = p2
        t = d * x1 * x2 * y1 * y2 % q
        return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, q) % q, (y1 * y2 - a * x1 * x2) * pow(1 - t, -1, q) % q)
    acc = (0, 1)
    for bit in bin(k)[2:]:
        acc = add(acc, acc)
        if bit == "1":
            acc = add(acc, pt)
    return acc


def main():
    q_b = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    r_b = 0x1cfb69d4ca675f520cce760202687600ff8f87007419047174fd06b52876e7e1
    q_j = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    r_j = 2736030358979909402780800718157159386076813972158567259200215660948447373041
    out = ["/* GENERATED by tools/gen_consts.py -- do not edit. */", "#pragma once", "#include <stdint.h>", "",
           "namespace avrf {", ""]
    out.append(field("FqBandersnatch", q_b))
    out.append(field("FrBandersnatch", r_b))
    out.append(field("FqBabyJubJub", q_j))
    out.append(field("FrBabyJubJub", r_j))
    r_jj = 6554484396890773809930967563523245729705921265872317281365359162392183254199     # JubJub prime-order subgroup
    out.append(field("FrJubJub", r_jj))
    q_e = 2 ** 255 - 19                                                                       # edwards25519 (RFC 8032)
    r_e = 2 ** 252 + 27742317777372353535851937790883648493
    out.append(field("FqEd25519", q_e))
    out.append(field("FrEd25519", r_e))
    p_bls = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    p_bn = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    out.append(field_n("FqBls12381", p_bls, 12))
    out.append(field_n("FqBn254", p_bn, 8))
    for nm, fq, fr, b, root, tw in (("G1Bls12381", "FqBls12381", "FqBandersnatch", 4,
                                     10238227357739495823651030575849232062558860180284477541189508159991286009131, 32),
                                    ("G1Bn254", "FqBn254", "FqBabyJubJub", 3,
                                     19103219067921713944291392827692070036145651957329286315305642004821462161904, 28)):
        pp = p_bls if fq == "FqBls12381" else p_bn
        rr = q_b if fr == "FqBandersnatch" else q_j
        nl = 12 if fq == "FqBls12381" else 8
        Rn = 1 << (32 * nl)
        lim = lambda x, n=nl: ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(n))
        out.append(f"struct {nm} {{ using Fq = {fq}; using Fr = {fr};  /* y^2 = x^3 + {b} */")
        out.append(f"  static constexpr uint32_t B[{nl}] = {{{lim(b * Rn % pp)}}};")
        out.append(f"  static constexpr int TWO_ADICITY = {tw};")
        out.append(f"  static constexpr uint32_t ROOT_OF_UNITY[8] = {{{limbs(root * R % rr)}}};  /* 2^{tw}-th root of unity in Fr, Montgomery */")
        # pairing (host_pairing.h): Fp2 = Fp[u]/(u^2+1), xi = XI0 + u, Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v)
        bls = fq == "FqBls12381"
        loop = 0xd201000000010000 if bls else 6 * 4965661367192848881 ** 2      # t - 1 (plain ate pairing)
        fexp = (pp ** 12 - 1) // rr
        l64 = lambda x: ", ".join("0x%016xULL" % ((x >> (64 * i)) & (2 ** 64 - 1)) for i in range((x.bit_length() + 63) // 64))
        out.append(f"  static constexpr int XI0 = {1 if bls else 9};  static constexpr bool MTWIST = {'true' if bls else 'false'};")
        out.append(f"  static constexpr int ATE_LOOP_BITS = {loop.bit_length()};")
        out.append(f"  static constexpr uint64_t ATE_LOOP[{(loop.bit_length() + 63) // 64}] = {{{l64(loop)}}};")
        out.append(f"  static constexpr int FINAL_EXP_BITS = {fexp.bit_length()};")
        out.append(f"  static constexpr uint64_t FINAL_EXP[{(fexp.bit_length() + 63) // 64}] = {{{l64(fexp)}}};  /* (p^12 - 1) / r */")
        # (p^12 - 1) / r = (p^6 - 1) (p^2 + 1) * HARD: the first two factors cost a conjugation, an inversion and a p^2-Frobenius
        hard = (pp ** 4 - pp ** 2 + 1) // rr
        assert (pp ** 6 - 1) * (pp ** 2 + 1) * hard == fexp
        xi0 = 1 if bls else 9
        gamma = pow(xi0 * xi0 + 1, (pp - 1) // 6, pp)                    # xi^((p^2-1)/6) = N(xi)^((p-1)/6), in Fp
        out.append(f"  static constexpr int HARD_EXP_BITS = {hard.bit_length()};")
        out.append(f"  static constexpr uint64_t HARD_EXP[{(hard.bit_length() + 63) // 64}] = {{{l64(hard)}}};  /* (p^4 - p^2 + 1) / r */")
        out.append(f"  static constexpr uint32_t FROB2_GAMMA[{nl}] = {{{lim(gamma * Rn % pp)}}};  /* xi^((p^2-1)/6), Montgomery */")
        # p-Frobenius coefficients gamma1^k, gamma1 = xi^((p-1)/6) in Fp2 = Fp[u]/(u^2+1) (BLS12 hard part by the x-chain)
        def f2mul(a, b):
            return ((a[0] * b[0] - a[1] * b[1]) % pp, (a[0] * b[1] + a[1] * b[0]) % pp)
        def f2pow(a, e):
            r = (1, 0)
            while e:
                if e & 1:
                    r = f2mul(r, a)
                a = f2mul(a, a); e >>= 1
            return r
        g1 = f2pow((xi0, 1), (pp - 1) // 6)
        gk, rows = (1, 0), []
        for k in range(1, 6):
            gk = f2mul(gk, g1)
            rows.append("{{" + lim(gk[0] * Rn % pp) + "}, {" + lim(gk[1] * Rn % pp) + "}}")
        out.append(f"  static constexpr uint32_t FROB1_GAMMA[5][2][{nl}] = {{{', '.join(rows)}}};  /* xi^(k(p-1)/6), k = 1..5: (re, im), Montgomery */")
        out.append(f"  static constexpr bool X_CHAIN = {'true' if bls else 'false'};  /* hard part = (x-1)^2 (x+p) (x^2+p^2-1) + 3 (BLS12) */")
        out.append(f"  static constexpr uint64_t X_ABS = 0x{(0xd201000000010000 if bls else 4965661367192848881):x}ULL;  static constexpr bool X_NEG = {'true' if bls else 'false'};")
        out.append("};")
    out.append("")
    glv_b = glv_bandersnatch(q_b, r_b)
    {   # psi(G) = [lambda] G: checks the two constants of the rational map against plain affine arithmetic
        "check": (lambda G, d: (lambda x, y, L: (
            (glv_b["ec"] * (1 - y * y) * pow(x * y, -1, q_b) % q_b, glv_b["eb"] * (y * y + glv_b["eb"]) * pow(y * y - glv_b["eb"], -1, q_b) % q_b) == L
            or (_ for _ in ()).throw(AssertionError("psi(G) != [lambda] G"))))(G[0], G[1], te_affine_mul(q_b, q_b - 5, d, G, glv_b["lam"])))(
                (18886178867200960497001835917649091219057080094937609519140440539760939937304,
                 19188667384257783945677642223292697773471335439753913231509108946878080696678),
                45022363124591815672509500913686876175488063829319466900776701791074614335719)}
    out.append(suite("SuiteBandersnatch", 0, "Bandersnatch-SHA512-ELL2-v1", "FqBandersnatch", "FrBandersnatch", q_b, r_b, 1,
                     45022363124591815672509500913686876175488063829319466900776701791074614335719,
                     {"G": (18886178867200960497001835917649091219057080094937609519140440539760939937304,
                            19188667384257783945677642223292697773471335439753913231509108946878080696678),
                      "B": (23335687741101763108036518445642207119627658113885888016488710494487028845889,
                            5552214580375038693022409684979828600325210968745774080859660443337357929963),
                      "ACC": (14056632001415368875257708737821299882600475929746323097150942355715730684350,
                              10322661992765989500407719465917595459409463902187386706652408883505670839210),
                      "PAD": (26913883415342152801331916189968962157924271221160514298872262294143390094043,
                              30874728313203001508631936119690348239461579770372782660098261717479009115354)}, 4,
                     ell2=(29978822694968839326280996386011761570173833766074948509196803838190355340952,
                           25465760566081946422412445027709227188579564747101592991722834452325077642517),
                     glv=glv_b))
    out.append(suite("SuiteBabyJubJub", 1, "BabyJubJub-SHA512-TAI-v1", "FqBabyJubJub", "FrBabyJubJub", q_j, r_j, 0,
                     9706598848417545097372247223557719406784115219466060233080913168975159366771,
                     {"G": (19698561148652590122159747500897617769866003486955115824547446575314762165298,
                            19298250018296453272277890825869354524455968081175474282777126169995084727839),
                      "B": (15549380791300914366206471199568039679131690710803662429646809536753521087193,
                            15218614024055502695611547593111691164731001864276292210438920202280814188379),
                      "ACC": (6402374321243162085389111671722843560682527921646684137786768606010797479351,
                              9735581299071570006712034490635195155689931359428941496570758703259384062170),
                      "PAD": (11167490195257431015694161063225325511805242064780376648595733691987293447528,
                              18403369502642103292159933062507105566469227524991433735553439433605496057425)}, 8))
    # JubJub-SHA512-TAI-v1 (src/suites/jubjub.rs:56-95): ark-ed-on-bls12-381, a = -1, d = -(10240/10241), base field = Fr(BLS12-381)
    out.append(suite("SuiteJubJub", 2, "JubJub-SHA512-TAI-v1", "FqBandersnatch", "FrJubJub", q_b, r_jj, 2,
                     19257038036680949359750312669786877991949435402254120286184196891950884077233,
                     {"G": (8076246640662884909881801758704306714034609987455869804520522091855516602923,
                            13262374693698910701929044844600465831413122818447359594527400194675274060458),
                      "B": (38206460563694846719174258613922853630278999941532690543235578292520143148532,
                            34254498978062207918041301829525626783549813531091321004550549786528984401675),
                      "ACC": (48142684311216766702182564801462043940571084233680216669499475549492432046964,
                              34380560660182334518990118617091967209302636551264477863958902286043397647879),
                      "PAD": (17348704025397475127937572481155408456556065464328870407269802701696798733683,
                              24318278422173803457621119807961883607097742387673491974779969503617097905596)}, 8))
    # Ed25519-SHA512-TAI-v1 (src/suites/ed25519.rs:44-66): ark-ed25519, a = -1, d = -121665/121666; Tiny / Thin / Pedersen only
    # (no pairing-friendly curve carries its base field): ACC / PAD are placeholders, the ring entry points reject the suite
    d_e = (-121665 * pow(121666, -1, q_e)) % q_e
    g_e = (15112221349535400772501151409588531511454012693041857206046113283949847762202,
           46316835694926478169428394003475163141307993866256225615783033603165251855960)
    out.append(suite("SuiteEd25519", 3, "Ed25519-SHA512-TAI-v1", "FqEd25519", "FrEd25519", q_e, r_e, 2, d_e,
                     {"G": g_e,
                      "B": (45003173884697328536089278691112838614164406922820087464913813433380838325453,
                            31256014272390301975555524011230972931324093235775711248505761870355310252869),
                      "ACC": g_e, "PAD": g_e}, 8))
    # Bandersnatch-SW-SHA512-TAI-v1 (src/suites/bandersnatch_sw.rs:60-112): suite 0's curve in its short-Weierstrass presentation.
    # The maps take the SW generator to the TE generator, so G, D, the GLV data are suite 0's; the suite points are given as SW
    # coordinates in the reference and mapped here.
    A_m = 29978822694968839326280996386011761570173833766074948509196803838190355340952
    B_m = 25465760566081946422412445027709227188579564747101592991722834452325077642517
    a3 = 9992940898322946442093665462003920523391277922024982836398934612730118446984
    binv = 41180284393978236561320365279764246793818536543197771097409483252169927600582
    assert a3 * 3 % q_b == A_m and binv * B_m % q_b == 1
    sw_a = (3 - A_m * A_m) * pow(3 * B_m * B_m, -1, q_b) % q_b
    sw_b = (2 * A_m ** 3 - 9 * A_m) * pow(27 * B_m ** 3, -1, q_b) % q_b

    def sw_to_te(x, y):
        assert (y * y - x ** 3 - sw_a * x - sw_b) % q_b == 0
        mx, my = (B_m * x - a3) % q_b, B_m * y % q_b
        return mx * pow(my, -1, q_b) % q_b, (mx - 1) * pow(mx + 1, -1, q_b) % q_b
    g0 = (18886178867200960497001835917649091219057080094937609519140440539760939937304,
          19188667384257783945677642223292697773471335439753913231509108946878080696678)
    out.append(suite("SuiteBandersnatchSW", 4, "Bandersnatch-SW-SHA512-TAI-v1", "FqBandersnatch", "FrBandersnatch", q_b, r_b, 1,
                     45022363124591815672509500913686876175488063829319466900776701791074614335719,
                     {"G": g0,
                      "B": sw_to_te(28115362618644671219696075022370511395136332234538034358311199318506963235315,
                                    3900851469868158154936962463930962496000252801946757953905982128670530185313),
                      "ACC": sw_to_te(13189182432637108534251278524663360416811744717379968387043749958796254980045,
                                      14483286006782706188671626508232161325054303360192563232232823772738911894793),
                      "PAD": sw_to_te(20496180070424734470560955314776462366297546779079302509428101119888111900885,
                                      8839106592405352067483360946162273985142890146060814748321063063028225641813)}, 4,
                     glv=glv_b, sw=dict(MONT_B=B_m, MONT_A3=a3, MONT_BINV=binv, SW_A=sw_a, SW_B=sw_b)))
    # Bandersnatch-SHAKE128-ELL2-v1 (src/suites/bandersnatch_shake128.rs): suite 0's curve and Elligator2 map, the SHAKE128 sponge
    # as the transcript and expand_message_xof in hash-to-curve
    out.append(suite("SuiteBandersnatchShake", 5, "Bandersnatch-SHAKE128-ELL2-v1", "FqBandersnatch", "FrBandersnatch", q_b, r_b, 1,
                     45022363124591815672509500913686876175488063829319466900776701791074614335719,
                     {"G": g0,
                      "B": (6153734995852631824944342602386415873379775188383988340041079006556670120775,
                            27204351599954061630605768787803524395123895650061061132592995395630473050754),
                      "ACC": (27631238720955528589004064829276283990465032040945349648037876197995278250917,
                              37605358688136619817560700742505556266961225274493904038881144193539047100140),
                      "PAD": (1834402953989431481748983728202937234471322740714585873803966488035889514523,
                              52100941849053769665273763352270294131006971127418863694682093199651869272752)}, 4,
                     ell2=(29978822694968839326280996386011761570173833766074948509196803838190355340952,
                           25465760566081946422412445027709227188579564747101592991722834452325077642517),
                     glv=glv_b, shake=True))
    # Testing-SHA256-TAI-v1 (src/suites/testing.rs): the crate's own test suite -- edwards25519 with HashTranscript<Sha256>
    out.append(suite("SuiteTesting", 6, "Testing-SHA256-TAI-v1", "FqEd25519", "FrEd25519", q_e, r_e, 2, d_e,
                     {"G": g_e,
                      "B": (3310617998588019043596181043598335786888094217571323926547956053100032777190,
                            16824531136491949759823061604778551593864344614632277377095388820423530178202),
                      "ACC": g_e, "PAD": g_e}, 8, sha256=True))
    # Secp256r1-SHA256-TAI-v1 (src/suites/secp256r1.rs:49-70): NIST P-256 (SP 800-186 3.2.1.3), both fields 256 bits with the top bit set
    q_p = 0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff
    r_p = 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551
    out.insert(out.index(next(x for x in out if x.startswith("struct FqBls12381"))), field("FqSecp256r1", q_p))
    out.insert(out.index(next(x for x in out if x.startswith("struct FqBls12381"))), field("FrSecp256r1", r_p))
    g_p = (0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296, 0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5)
    out.append(suite_sw_native("SuiteSecp256r1", 7, "Secp256r1-SHA256-TAI-v1", "FqSecp256r1", "FrSecp256r1", q_p, q_p - 3,
                               0x5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b,
                               {"G": g_p,
                                "B": (100063053743935619201936855760019111820847755970243670581468062459849338000,
                                      113675507039234898358330549589155441528265243038226986303017485279501143145422),
                                "ACC": g_p, "PAD": g_p}))
    out += ["", "}  // namespace avrf", ""]
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
