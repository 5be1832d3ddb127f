#!/usr/bin/env python3
"""Model of ark_vrf_amd/csrc/fpu.h / fpu_te.h / fpu_g1.h (unsaturated signed limbs) in exact integer arithmetic.

  python tools/fpu_model.py            # all checks

What it proves, for every field / curve of consts_gen.h the kernels instantiate:
  1. fu_mul / fu_sqr: on random and extreme limb vectors inside the documented magnitude bounds, no column accumulator leaves
     the signed 64-bit range, the result is a b 2^(-W L) mod p with limbs 0..L-2 in [0, 2^W) and |value| < |a b| / 2^(W L) + p;
     the worst case column sum is also bounded analytically.
  2. teu_madd (twisted Edwards, a in {1, -1, -5}): chains of mixed additions with random signs agree with the affine group law;
     at every product the limb magnitudes are within (2^30, 2^29 + 4); the value bounds |X| < 1.5p, |Y| < 1.7p, |T| < 2.4p,
     |Z| < 1.3p are inductive (interval arithmetic over the formulas, worst case) and fu_to_packed returns the canonical value.
  3. g1u_madd (XYZZ, a = 0): the same for the short-Weierstrass mixed addition (value bounds printed), incl. the closing
     conversion by two constant multiplications.
"""
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONSTS = os.path.join(ROOT, "ark_vrf_amd", "csrc", "consts_gen.h")
I64 = 1 << 63


def parse():
    txt = open(CONSTS).read()
    out = {}
    for m in re.finditer(r"struct (\w+) \{(.*?)\n\};", txt, re.S):
        name, body = m.group(1), m.group(2)
        d = {}
        for a in re.finditer(r"uint32_t (\w+)\[(\d+)\] = \{([^}]*)\}", body):
            d[a.group(1)] = sum(int(x.strip().rstrip("u"), 16) << (32 * i) for i, x in enumerate(a.group(3).split(",")))
            d[a.group(1) + "_n"] = int(a.group(2))
        for a in re.finditer(r"(?:int|uint32_t|bool) (\w+) = (\w+)u?;", body):
            v = a.group(2)
            d[a.group(1)] = {"true": 1, "false": 0}.get(v, None) if v in ("true", "false") else int(v.rstrip("u"), 0)
        a = re.search(r"uint8_t SQRT_HIDX\[\d+\] = \{([^}]*)\}", body)
        if a:
            d["SQRT_HIDX_tab"] = [int(x) for x in a.group(1).replace("\n", " ").split(",")]
        u = re.search(r"using Fq = (\w+);", body)
        if u:
            d["Fq"] = u.group(1)
        out[name] = d
    return out


class Field:
    def __init__(self, name, d):
        self.name, self.p, self.N = name, d["P"], d["P_n"]
        self.W, self.L = (29, 9) if self.N == 8 else (28, 14)
        self.SH = self.W * self.L - 32 * self.N
        self.MASK = (1 << self.W) - 1
        self.ninv = d["NINV"] & self.MASK
        assert (self.p * self.ninv + 1) % (1 << self.W) == 0
        self.R = 1 << (32 * self.N)            # saturated form
        self.Ru = 1 << (self.W * self.L)       # unsaturated form
        self.pl = self.slice_pos(self.p)
        self.max_col = 0
        # p = 1 mod 2^W (Bandersnatch's base field): the reduction SUBTRACTS lo_k p, lo_k = the column's low W bits as they are -- no
        # negation, no multiply for m_k, and the column's own term lo_k p_0 = lo_k is what the arithmetic shift drops (fpu.h fu_mul)
        self.p0_one = self.pl[0] == 1 and self.ninv == self.MASK

    def slice_pos(self, v):
        """limbs of a non-negative value (top limb takes the rest)"""
        return [(v >> (self.W * i)) & self.MASK for i in range(self.L - 1)] + [v >> (self.W * (self.L - 1))]

    def slice(self, words_value, S):
        v = words_value << S
        assert v < 1 << (self.W * self.L)
        return self.slice_pos(v)

    def val(self, limbs):
        return sum(l << (self.W * i) for i, l in enumerate(limbs))

    def acc_ok(self, acc):
        assert -I64 <= acc < I64, "int64 overflow in a column"
        self.max_col = max(self.max_col, abs(acc))

    def mul(self, a, b, sqr=False):
        L, W = self.L, self.W
        for x in a + b:
            assert -(1 << 31) <= x < (1 << 31)
        acc, m, r = 0, [0] * L, [0] * L
        for k in range(2 * L - 1):
            lo, hi = (0, k) if k < L else (k - L + 1, L - 1)
            for i in range(lo, hi + 1):
                acc += a[i] * b[k - i]; self.acc_ok(acc)
            for i in range(lo, min(hi, k - 1) + 1):
                acc += (-m[i] if self.p0_one else m[i]) * self.pl[k - i]; self.acc_ok(acc)
            if k < L and self.p0_one:
                m[k] = acc & self.MASK                         # (acc - m_k) >> W == acc >> W: the floor of the arithmetic shift
            elif k < L:
                m[k] = ((acc & 0xffffffff) * self.ninv) & self.MASK
                acc += m[k] * self.pl[0]; self.acc_ok(acc)
                assert acc & self.MASK == 0
            else:
                r[k - L] = acc & self.MASK
            acc >>= W
        assert -(1 << 31) <= acc < (1 << 31)
        r[L - 1] = acc
        va, vb, vr = self.val(a), self.val(b), self.val(r)
        assert (vr * self.Ru - va * vb) % self.p == 0
        assert abs(vr) < abs(va * vb) // self.Ru + self.p + 1
        return r

    def carry(self, a):
        L, W = self.L, self.W
        r = [a[0] & self.MASK] + [(a[i] & self.MASK) + (a[i - 1] >> W) for i in range(1, L - 1)] + [a[L - 1] + (a[L - 2] >> W)]
        assert self.val(r) == self.val(a)
        return r

    def to_packed(self, a, KB=2):
        v = self.val(a)
        assert abs(v) < (1 << KB) * self.p, "fu_to_packed<KB> wants |value| < 2^KB p"
        L, W = self.L, self.W
        pk = self.slice_pos((1 << KB) * self.p)
        u, c = [0] * L, 0
        for i in range(L - 1):
            t = a[i] + pk[i] + c
            assert -(1 << 31) <= t < (1 << 31)
            u[i] = t & self.MASK; c = t >> W
        u[L - 1] = a[L - 1] + pk[L - 1] + c
        assert 0 <= u[L - 1] < 1 << 32
        x = self.val(u)
        assert x == v + (1 << KB) * self.p and x < 1 << (32 * (self.N + 1))
        K = 1 << KB
        while K:
            if x >= K * self.p:
                x -= K * self.p
            K >>= 1
        assert x == v % self.p
        return x


def rand_limbs(f, rng, bound, top_bound):
    pick = lambda b: rng.choice([b, -b, b - 1, 0, rng.randrange(-b, b + 1)])
    return [pick(bound) for _ in range(f.L - 1)] + [pick(top_bound)]


def check_mul(f, rng, rounds=300):
    f.max_col = 0
    A, B = 1 << 30, (1 << 29) + 4
    # analytic worst case: L products |a_i b_j| + L products m_i p_j + the carry from below
    worst = f.L * A * B + f.L * f.MASK * f.MASK + (1 << 36)
    assert worst < I64, (f.name, worst / I64)
    for r in range(rounds):
        if r < 8:                                            # every limb at its extreme, all sign patterns of the two operands
            a = [A if (r & 1) else -A] * f.L; b = [B if (r & 2) else -B] * f.L
            a[-1] = b[-1] = (1 << 28) * (1 if r & 4 else -1)
        else:
            a, b = rand_limbs(f, rng, A, 1 << 28), rand_limbs(f, rng, B, 1 << 28)
        f.mul(a, b)
    for r in range(rounds // 3):
        a = rand_limbs(f, rng, B, 1 << 28)
        f.mul(a, a, sqr=True)
    print(f"  {f.name}: {f.L} x {f.W}-bit limbs, SH = {f.SH}; fu_mul ok on {rounds} operand pairs, largest column |acc| = 2^{f.max_col.bit_length() - 1}."
          f"{(f.max_col >> (f.max_col.bit_length() - 8)) & 0x7f:02x}.. (analytic worst case {worst / I64:.3f} of 2^63)")


# ---------------------------------------------------------------------------------------------- twisted Edwards

def te_bounds(p_bits_frac, a_kind, SH, WL):
    """interval arithmetic over teu_madd in units of p: returns the output bounds for input bounds (x, y, t, z)"""
    ratio = 2.0 ** (WL - p_bits_frac)                      # R' / p
    kb = 2.0 ** SH / ratio                                 # |A| < kb |X| + 1 for a base coordinate < p sliced with the shift
    x, y, t, z = 1.5, 1.7, 2.4, 1.3
    A, B, C = kb * x + 1, kb * y + 1, kb * t + 1
    E = 2 * kb * (x + y) + 1 + A + B
    FG = z + C
    H = {0: B + A, 2: B + A, 1: B + 5 * A}[a_kind]
    return (E * FG / ratio + 1, FG * H / ratio + 1, E * H / ratio + 1, FG * FG / ratio + 1), (E, FG, H)


class TE:
    def __init__(self, name, d, f):
        self.name, self.f, self.a_kind = name, f, d["A_KIND"]
        p = f.p
        rinv = pow(f.R, -1, p)
        self.a = {0: 1, 1: p - 5, 2: p - 1}[self.a_kind]
        self.d = d["D"] * rinv % p
        self.G = (d["G_X"] * rinv % p, d["G_Y"] * rinv % p)
        x, y = self.G
        assert (self.a * x * x + y * y - 1 - self.d * x * x * y * y) % p == 0

    def add(self, P, Q):
        p = self.f.p
        (x1, y1), (x2, y2) = P, Q
        k = self.d * x1 * x2 * y1 * y2 % p
        return ((x1 * y2 + y1 * x2) * pow(1 + k, -1, p) % p, (y1 * y2 - self.a * x1 * x2) * pow(1 - k, -1, p) % p)

    def neg(self, P):
        return ((-P[0]) % self.f.p, P[1])

    def pre(self, P):
        """saturated Montgomery words of (x, y, d x y) as integers"""
        f = self.f
        return (P[0] * f.R % f.p, P[1] * f.R % f.p, self.d * P[0] * P[1] % f.p * f.R % f.p)

    def limb_check(self, a, A, b, B):
        assert max(abs(v) for v in a) <= A and max(abs(v) for v in b) <= B, (self.name, "limb bound")

    def madd(self, acc, q, neg):
        f, L = self.f, self.f.L
        X, Y, T, Z, s = acc
        e = -1 if neg else 0
        flip = s ^ e
        cneg = lambda v, m: [(l ^ m) - m for l in v]
        X1, T1 = cneg(X, flip), cneg(T, flip)
        qx, qy, qk = q
        xy = qx + qy
        assert xy < 1 << 256
        n29, n30 = (1 << 29) + 4, 1 << 30
        sx, sy, sk, sxy = (f.slice(v, f.SH) for v in (qx, qy, qk, xy))
        self.limb_check(X1, n29, sx, n29); A = f.mul(X1, sx)
        self.limb_check(Y, n29, sy, n29); B = f.mul(Y, sy)
        self.limb_check(T1, n29, sk, n29); C = f.mul(T1, sk)
        XY = [a + b for a, b in zip(X1, Y)]
        self.limb_check(XY, n30, sxy, n29); E = f.mul(XY, sxy)
        E = [e_ - a - b for e_, a, b in zip(E, A, B)]
        F = [z - c for z, c in zip(Z, C)]; G = [z + c for z, c in zip(Z, C)]
        if self.a_kind == 1:
            H = TEChain(self).hsum(A, B)
        elif self.a_kind == 2:
            H = f.carry([b + a for a, b in zip(A, B)])
        else:
            H = [b - a for a, b in zip(A, B)]
        self.limb_check(E, n30, F, n29); X3 = f.mul(E, F)
        self.limb_check(G, n30, H, n29); Y3 = f.mul(G, H)
        self.limb_check(E, n30, H, n29); T3 = f.mul(E, H)
        self.limb_check(G, n30, F, n29); Z3 = f.mul(G, F)
        for v, b in ((X3, 1.5), (Y3, 1.7), (T3, 2.4), (Z3, 1.3)):
            assert abs(f.val(v)) < b * f.p, (self.name, "value bound", abs(f.val(v)) / f.p, b)
        return (X3, Y3, T3, Z3, e)

    def from_pre(self, q, neg):
        f = self.f
        x, y = f.slice(q[0], 0), f.slice(q[1], 0)
        return (x, y, f.mul(x, f.slice(q[1], f.SH)), f.slice(f.R % f.p, 0), -1 if neg else 0)

    def to_affine(self, acc):
        f = self.f
        X, Y, T, Z, s = acc
        cneg = lambda v, m: [(l ^ m) - m for l in v]
        x, y, t, z = f.to_packed(cneg(X, s)), f.to_packed(Y), f.to_packed(cneg(T, s)), f.to_packed(Z)
        assert (x * y - t * z) % f.p == 0
        zi = pow(z, -1, f.p)
        return (x * zi % f.p, y * zi % f.p)


def i32(v):
    assert -(1 << 31) <= v < (1 << 31), "int32 overflow in a limb expression"
    return v


def u32(v):
    assert 0 <= v < (1 << 32), "uint32 overflow in a limb expression"
    return v


class TEChain:
    """teu4_dbl / teu4_add_sat / teu4_madd_pre of fpu_te.h: the running point of the per-item kernels' scalar multiplications (and
    the sum B - a A of teu_madd), with the C types' ranges asserted (a = -5: unsigned limbs, fu_carry_u)"""
    INV = (1.6, 1.8, 2.5, 1.9)

    def __init__(self, te):
        self.te, self.f = te, te.f

    def lim(self, a, A, b, B):
        assert max(abs(v) for v in a) <= A and max(abs(v) for v in b) <= B, (self.te.name, "limb bound (chain)")

    def carry_u(self, h, top):
        f = self.f
        return [h[0] & f.MASK] + [(h[i] & f.MASK) + (h[i - 1] >> f.W) for i in range(1, f.L - 1)] + [top + (h[f.L - 2] >> f.W)]

    def hsum(self, A, B):
        f, k = self.f, self.te.a_kind
        if k == 1:
            r = self.carry_u([u32(b + 5 * a) for a, b in zip(A[:-1], B[:-1])], i32(B[-1] + 5 * A[-1]))
            assert f.val(r) == f.val(B) + 5 * f.val(A)
            return r
        if k == 2:
            return f.carry([i32(a + b) for a, b in zip(A, B)])
        return [i32(b - a) for a, b in zip(A, B)]

    def check_out(self, r):
        f = self.f
        for v, b in zip(r, self.INV):
            assert abs(f.val(v)) < b * f.p, (self.te.name, "chain value bound", abs(f.val(v)) / f.p, b)
        return r

    def dbl(self, P):
        f, k = self.f, self.te.a_kind
        X, Y, T, Z = P
        n29, n30 = (1 << f.W) + 4, 1 << (f.W + 1)
        self.lim(X, n29, Y, n29); self.lim(Z, n29, Z, n29)
        A, B, Zs = f.mul(X, X, sqr=True), f.mul(Y, Y, sqr=True), f.mul(Z, Z, sqr=True)
        s = f.carry([i32(a + b) for a, b in zip(X, Y)])
        Sq = f.mul(s, s, sqr=True)
        E = [i32(q - a - b) for q, a, b in zip(Sq, A, B)]
        if k == 1:
            A5 = self.carry_u([u32(5 * a) for a in A[:-1]], i32(5 * A[-1]))
            assert f.val(A5) == 5 * f.val(A)
            G = [i32(b - a) for a, b in zip(A5, B)]; H = [i32(-a - b) for a, b in zip(A5, B)]
        else:
            D = [(-a if k == 2 else a) for a in A]
            G = [i32(d + b) for d, b in zip(D, B)]; H = [i32(d - b) for d, b in zip(D, B)]
        F = [i32(g - 2 * z) for g, z in zip(G, Zs)]
        H, F = f.carry(H), f.carry(F)
        self.lim(E, n30, F, n29); self.lim(G, n30, H, n29)
        return self.check_out((f.mul(E, F), f.mul(G, H), f.mul(E, H), f.mul(F, G)))

    def add_sat(self, P, e):
        """e = (x, y, t, z) saturated canonical Montgomery values"""
        f, te = self.f, self.te
        X, Y, T, Z = P
        n29, n30 = (1 << f.W) + 4, 1 << (f.W + 1)
        ex, ey, et, ez = e
        sl = lambda v: f.slice(v, f.SH)
        A, B = f.mul(X, sl(ex)), f.mul(Y, sl(ey))
        C = f.mul(f.mul(T, sl(et)), sl(te.d * f.R % f.p))
        D = f.mul(Z, sl(ez))
        XY = [i32(a + b) for a, b in zip(X, Y)]
        self.lim(XY, n30, sl(ex + ey), n29)
        E = f.mul(XY, sl(ex + ey))
        E = [i32(q - a - b) for q, a, b in zip(E, A, B)]
        F = [i32(d - c) for d, c in zip(D, C)]; G = [i32(d + c) for d, c in zip(D, C)]
        H = self.hsum(A, B)
        self.lim(E, n30, F, n29); self.lim(G, n30, H, n29)
        return self.check_out((f.mul(E, F), f.mul(G, H), f.mul(E, H), f.mul(F, G)))

    def add_gen(self, P, Q):
        """qu_add of te_quad.h (the four-lanes-per-point general addition of the reduction tails, one coordinate per lane): both
        operands are running values (domain R, any common scale each); the operand order of every product is the kernel's"""
        f, te = self.f, self.te
        n29, n30 = (1 << f.W) + 4, 1 << (f.W + 1)
        for c in P + Q:
            assert max(abs(v) for v in c) <= n29, (te.name, "quad operand not normalised")
        A, B, TT, D = (f.mul(a, b) for a, b in zip(P, Q))
        sa = [i32(a + b) for a, b in zip(P[0], P[1])]
        sb = f.carry([i32(a + b) for a, b in zip(Q[0], Q[1])])
        self.lim(sa, n30, sb, n29)
        Ep = f.mul(sa, sb)
        C = f.mul(TT, f.slice(te.d * f.R % f.p, f.SH))
        E = f.carry([i32(q - a - b) for q, a, b in zip(Ep, A, B)])
        H = self.hsum(A, B) if te.a_kind != 0 else f.carry([i32(b - a) for a, b in zip(A, B)])
        if te.a_kind == 2:
            H = H                                               # hsum carried it
        G = f.carry([i32(d + c) for d, c in zip(D, C)]); F = f.carry([i32(d - c) for d, c in zip(D, C)])
        for v in (E, F, G, H):
            assert max(abs(x) for x in v) <= n29, (te.name, "quad closing operand")
        return self.check_out((f.mul(E, F), f.mul(H, G), f.mul(E, H), f.mul(F, G)))

    def from_ext(self, e):
        return tuple(self.f.slice(v, 0) for v in e)

    def to_affine(self, P):
        f = self.f
        x, y, t, z = (f.to_packed(v) for v in P)
        assert (x * y - t * z) % f.p == 0
        zi = pow(z, -1, f.p)
        return (x * zi % f.p, y * zi % f.p)


def chain_bounds(te):
    """interval arithmetic: INV is closed under teu4_dbl and teu4_add_sat (units of p)"""
    f = te.f
    ratio = f.Ru / f.p
    kb = 2.0 ** f.SH / ratio
    x, y, t, z = TEChain.INV
    am = {0: 1, 1: 5, 2: 1}[te.a_kind]
    sq = lambda v: v * v / ratio + 1
    A, B, Zs, Sq = sq(x), sq(y), sq(z), (x + y) ** 2 / ratio + 1
    E = Sq + A + B; G = am * A + B; H = G; F = G + 2 * Zs
    d = (E * F / ratio + 1, G * H / ratio + 1, E * H / ratio + 1, F * G / ratio + 1)
    A, B = kb * x + 1, kb * y + 1
    C = kb * (kb * t + 1) + 1; D = kb * z + 1
    E = 2 * kb * (x + y) + 1 + A + B; FG = D + C; H = B + am * A
    a = (E * FG / ratio + 1, FG * H / ratio + 1, E * H / ratio + 1, FG * FG / ratio + 1)
    for out in (d, a):
        assert all(o < i for o, i in zip(out, TEChain.INV)), (te.name, "chain invariant not closed", out)
    return d, a


def check_te_chain(te, rng, n_scalars=4):
    ch = TEChain(te)
    f = te.f
    d, a = chain_bounds(te)
    sat = lambda P: (P[0] * f.R % f.p, P[1] * f.R % f.p, P[0] * P[1] % f.p * f.R % f.p, f.R % f.p)
    pts = [te.G]
    for _ in range(5):
        pts.append(te.add(pts[-1], te.G))
    tab = [None] + pts                                               # tab[d] = d G
    for _ in range(n_scalars):
        k = rng.getrandbits(48) | (1 << 47)
        acc = ch.from_ext(sat((0, 1)))
        ref = (0, 1)
        for w in reversed(range(0, 48, 2)):                           # 2-bit windows: two doublings, one table addition
            acc = ch.dbl(ch.dbl(acc)); r2 = te.add(ref, ref); ref = te.add(r2, r2)
            dgt = (k >> w) & 3
            if dgt:
                acc = ch.add_sat(acc, sat(tab[dgt])); ref = te.add(ref, tab[dgt])
        assert ch.to_affine(acc) == ref, te.name
    # the reduction tails' general addition: sums of running points, of freshly loaded (sliced) points, and a tree of them
    run = []
    for i in range(6):
        qa = ch.from_ext(sat(pts[i % len(pts)])); ra = pts[i % len(pts)]
        for j in range(i + 1):
            qa = ch.add_gen(qa, ch.from_ext(sat(pts[(i + j) % len(pts)]))); ra = te.add(ra, pts[(i + j) % len(pts)])
        run.append((qa, ra))
    while len(run) > 1:
        (qa, ra), (qb, rb) = run.pop(), run.pop()
        c = ch.add_gen(qa, qb); assert ch.to_affine(c) == te.add(ra, rb), (te.name, "general addition")
        c2 = ch.add_gen(c, c); assert ch.to_affine(c2) == te.add(te.add(ra, rb), te.add(ra, rb))          # P + P: the law is complete
        ident = ch.from_ext(sat((0, 1)))
        assert ch.to_affine(ch.add_gen(c, ident)) == te.add(ra, rb)
        run.insert(0, (c, te.add(ra, rb)))
    print(f"  {te.name}: doubling chains (teu4_dbl / teu4_add_sat) and the reduction tails' general addition (qu_add) == affine law; invariant |X|,|Y|,|T|,|Z| < {TEChain.INV} p closed "
          f"(worst case out: dbl {d[0]:.2f} {d[1]:.2f} {d[2]:.2f} {d[3]:.2f}, add {a[0]:.2f} {a[1]:.2f} {a[2]:.2f} {a[3]:.2f})")


def check_te(te, rng, chains=6, length=40):
    f = te.f
    (ox, oy, ot, oz), (E, FG, H) = te_bounds(_log2(f.p), te.a_kind, f.SH, f.W * f.L)
    assert ox < 1.5 and oy < 1.7 and ot < 2.4 and oz < 1.3, (te.name, ox, oy, ot, oz)
    assert max(E, FG, H) * f.p < 1 << (f.W * (f.L - 1) + 28)       # top limbs stay below 2^28
    pts = [te.G]
    for _ in range(12):
        pts.append(te.add(pts[-1], te.G))
    for c in range(chains):
        neg0 = bool(rng.getrandbits(1))
        P = rng.choice(pts)
        acc = te.from_pre(te.pre(P), neg0)
        ref = te.neg(P) if neg0 else P
        if c == 0:
            acc = (f.slice(0, 0), f.slice(f.R % f.p, 0), f.slice(0, 0), f.slice(f.R % f.p, 0), 0); ref = (0, 1)
        for _ in range(length):
            Q, neg = rng.choice(pts), bool(rng.getrandbits(1))
            acc = te.madd(acc, te.pre(Q), neg)
            ref = te.add(ref, te.neg(Q) if neg else Q)
        assert te.to_affine(acc) == ref, te.name
    print(f"  {te.name}: a kind {te.a_kind}; {chains} chains of {length} mixed additions == affine law; inductive bounds "
          f"|X|,|Y|,|T|,|Z| < 1.5p, 1.7p, 2.4p, 1.3p (worst case out: {ox:.2f} {oy:.2f} {ot:.2f} {oz:.2f})")


def _log2(v):
    import math
    return math.log2(v)


# ---------------------------------------------------------------------------------------------- short Weierstrass, XYZZ

class G1:
    """y^2 = x^3 + b, XYZZ.  fpu_g1.h: the accumulator holds (2^a R X, 2^b R Y, 2^g R ZZ, 2^d R ZZZ) (R = the saturated form's
    Montgomery radix) with 3a - 2b = SH, g = 2m, d = 3m; a base coordinate is sliced as x R 2^sx / y R 2^sy, sx = a + SH - g,
    sy = b + SH - d: every product then lands on the same weighted-projective class (see the header for the algebra)."""
    INV = (4.6, 2.6, 1.5, 1.5)                               # inductive value bounds |X|, |Y|, |ZZ|, |ZZZ| in units of p

    def __init__(self, name, d, f):
        self.name, self.f = name, f
        self.b = d["B"] * pow(f.R, -1, f.p) % f.p
        SH = f.SH
        self.a, self.bb = next((a, (3 * a - SH) // 2) for a in range(0, 9) if 3 * a >= SH and (3 * a - SH) % 2 == 0)
        self.m = max((self.a + 1) // 2, (self.bb + 2) // 3)
        self.g, self.d = 2 * self.m, 3 * self.m
        self.sx, self.sy = self.a + SH - self.g, self.bb + SH - self.d
        assert 0 <= self.sx <= SH and 0 <= self.sy <= SH, (name, self.sx, self.sy)

    def add(self, P, Q):
        p = self.f.p
        if P is None: return Q
        if Q is None: return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0: return None
            l = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            l = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (l * l - x1 - x2) % p
        return (x3, (l * (x1 - x3) - y1) % p)

    def rand_point(self, rng):
        p = self.f.p
        assert p % 4 == 3
        while True:
            x = rng.randrange(p); r = (x ** 3 + self.b) % p
            y = pow(r, (p + 1) // 4, p)
            if y * y % p == r: return (x, y)

    def lim(self, a, A, b, B):
        assert max(abs(v) for v in a) <= A and max(abs(v) for v in b) <= B, (self.name, "limb bound")

    def zero_mod_p(self, v, operand_value):
        """the kernel's test on a product's output: limbs all zero, or equal to p's -- valid because |operand|^2 / R' < p"""
        f = self.f
        assert operand_value * operand_value < f.Ru * f.p * 9 // 10, (self.name, "zero test out of range")
        is0 = all(l == 0 for l in v) or all(l == pl for l, pl in zip(v, f.pl))
        assert is0 == (f.val(v) % f.p == 0)
        return is0

    def madd(self, acc, q, neg, track=None):
        """acc + (neg ? -q : q), q = saturated Montgomery words (x R, y R); returns None for the exceptional cases (the caller
        of the model skips them; the kernel takes the saturated path)"""
        f, L = self.f, self.f.L
        X, Y, ZZ, ZZZ = acc
        qx, qy = q
        nrm, dbl = (1 << f.W) + 4, (1 << (f.W + 1)) + 8
        sx, sy = f.slice(qx, self.sx), f.slice(qy, self.sy)
        if neg: sy = [-v for v in sy]
        self.lim(ZZ, nrm, sx, nrm); U2 = f.mul(ZZ, sx)
        self.lim(ZZZ, nrm, sy, nrm); S2 = f.mul(ZZZ, sy)
        P = [u - x for u, x in zip(U2, X)]; R = [s - y for s, y in zip(S2, Y)]
        self.lim(P, dbl, R, dbl)
        PP = f.mul(P, P, sqr=True)
        if self.zero_mod_p(PP, abs(f.val(P))):
            return None
        self.lim(P, dbl, PP, nrm); PPP = f.mul(P, PP)
        self.lim(X, nrm, PP, nrm); Q = f.mul(X, PP)
        ZZ3 = f.mul(ZZ, PP); ZZZ3 = f.mul(ZZZ, PPP)
        self.lim(Y, nrm, PPP, nrm); T = f.mul(Y, PPP)
        RR = f.mul(R, R, sqr=True)
        self.zero_mod_p(RR, abs(f.val(R)))
        X3 = [r - a - 2 * b for r, a, b in zip(RR, PPP, Q)]
        QX = f.carry([a - b for a, b in zip(Q, X3)])
        self.lim(R, dbl, QX, nrm); Y3 = f.mul(R, QX)
        Y3 = [a - b for a, b in zip(Y3, T)]
        X3 = f.carry(X3)
        self.lim(X3, nrm, Y3, nrm)
        if track is not None:
            track.append(tuple(abs(f.val(v)) / f.p for v in (X3, Y3, ZZ3, ZZZ3)))
        return (X3, Y3, ZZ3, ZZZ3)

    def from_affine(self, q, neg):
        f = self.f
        y = f.slice(q[1], self.bb)
        return (f.slice(q[0], self.a), [-v for v in y] if neg else y, f.slice((f.R << self.g) % f.p, 0), f.slice((f.R << self.d) % f.p, 0))

    def from_xyzz(self, X, Y, ZZ, ZZZ):
        """a canonical saturated XYZZ point (the doubling branch computes one) in the accumulator's scaling: constant multiplications"""
        f = self.f
        cm = lambda v, k: f.mul(f.slice(v, 0), f.slice((f.Ru << k) % f.p, 0))
        return (cm(X, self.a), cm(Y, self.bb), cm(ZZ, self.g), cm(ZZZ, self.d))

    def to_saturated(self, acc):
        """what the readers of a partial sum do: the canonical saturated XYZZ point (mu = 2^m): X 2^(2m - a), Y 2^(3m - b) by constant multiplications"""
        f = self.f
        X, Y, ZZ, ZZZ = acc
        cm = lambda v, k: f.mul(v, f.slice((f.Ru << k) % f.p, 0)) if k else v
        out = [f.to_packed(cm(X, 2 * self.m - self.a), 4 if 2 * self.m == self.a else 2), f.to_packed(cm(Y, 3 * self.m - self.bb)), f.to_packed(ZZ), f.to_packed(ZZZ)]
        Rinv = pow(f.R, -1, f.p)
        return [v * Rinv % f.p for v in out]

    def to_affine(self, acc):
        f = self.f
        X, Y, ZZ, ZZZ = self.to_saturated(acc)
        assert (ZZ ** 3 - ZZZ ** 2) % f.p == 0
        return (X * pow(ZZ, -1, f.p) % f.p, Y * pow(ZZZ, -1, f.p) % f.p)


def g1_bounds(g, x, y, zz, zzz):
    """interval arithmetic over g1u_madd in units of p: output bounds, and the operands of the two zero tests"""
    f = g.f
    ratio = f.Ru / f.p
    U2, S2 = zz * 2.0 ** g.sx / ratio + 1, zzz * 2.0 ** g.sy / ratio + 1
    P, R = U2 + x, S2 + y
    PP = P * P / ratio + 1
    PPP = P * PP / ratio + 1; Q = x * PP / ratio + 1
    ZZ3 = zz * PP / ratio + 1; ZZZ3 = zzz * PPP / ratio + 1; T = y * PPP / ratio + 1
    RR = R * R / ratio + 1
    X3 = RR + PPP + 2 * Q
    Y3 = R * (Q + X3) / ratio + 1 + T
    assert P * P < 0.9 * ratio and R * R < 0.9 * ratio, (g.name, "zero test", P, R, ratio)
    assert max(P, R, X3, Y3) * f.p < 1 << (f.W * (f.L - 1) + f.W - 1)
    return X3, Y3, ZZ3, ZZZ3


def check_g1(g, rng, chains=4, length=25):
    f = g.f
    b = g1_bounds(g, *g.INV)
    assert all(o < i for o, i in zip(b, g.INV)), (g.name, b)
    # entry state outside the invariant: a sliced affine point (a converted doubling result is reduced); it must fall into it
    for st in ((2.0 ** g.a, 2.0 ** g.bb, 1.0, 1.0),):
        steps = 0
        while not all(o < i for o, i in zip(st, g.INV)):
            st = tuple(max(o, 0.0) for o in g1_bounds(g, *[max(s, i) if k < 0 else s for k, (s, i) in enumerate(zip(st, g.INV))]))
            steps += 1
            assert steps < 6, (g.name, "entry state does not contract", st)
    pts = [g.rand_point(rng) for _ in range(6)]
    sat = lambda P: (P[0] * f.R % f.p, P[1] * f.R % f.p)
    worst = [0.0] * 4
    for c in range(chains):
        P, neg0 = rng.choice(pts), bool(rng.getrandbits(1))
        acc = g.from_affine(sat(P), neg0)
        ref = (P[0], (-P[1]) % f.p) if neg0 else P
        if c == 1:                                           # start from a doubled point handed over in saturated XYZZ form (X, Y, 1, 1)
            ref = g.add(P, P); acc = g.from_xyzz(ref[0] * f.R % f.p, ref[1] * f.R % f.p, f.R % f.p, f.R % f.p)
        for _ in range(length):
            Q, neg = rng.choice(pts), bool(rng.getrandbits(1))
            Qs = (Q[0], (-Q[1]) % f.p) if neg else Q
            tr = []
            nxt = g.madd(acc, sat(Q), neg, tr)
            if nxt is None:
                assert ref is not None and Qs[0] == ref[0]  # exactly the exceptional inputs
                continue
            assert ref is not None and Qs[0] != ref[0]
            acc, ref = nxt, g.add(ref, Qs)
            worst = [max(w, t) for w, t in zip(worst, tr[0])]
        assert g.to_affine(acc) == ref, g.name
    print(f"  {g.name}: a, b, g, d = {g.a}, {g.bb}, {g.g}, {g.d}; base shifts {g.sx}, {g.sy}; {chains} chains of {length} XYZZ mixed additions == affine law "
          f"(P = +-Q detected exactly by the limb test); invariant |X|,|Y|,|ZZ|,|ZZZ| < {g.INV} p (interval worst case out: "
          f"{b[0]:.2f} {b[1]:.2f} {b[2]:.2f} {b[3]:.2f}; seen {worst[0]:.2f} {worst[1]:.2f} {worst[2]:.2f} {worst[3]:.2f})")


def check_sqrt_chain(f, d, rng, n=6):
    """fu_sqrt_ratio_nf of fpu_sqrt.h: u, v enter as (value R) * 2^SH (below 2^SH p), every later operand is a product's output or
    a slice, results leave through a product with R mod p and fu_to_packed<2>.  Runs the routine on random squares and
    non-squares with the model's multiplier (columns in 63 bits, limbs in range) and checks x^2 v = u."""
    p, L = f.p, f.L
    S = d["TWO_ADICITY"]
    t = (p - 1) >> S
    e = (t - 1) // 2
    g = d["ROOT"] * pow(f.R, -1, p) % p                         # a generator of the 2^S-th roots of unity (Montgomery form in the header)
    sl = lambda v: f.slice(v, f.SH)
    to_r = f.slice_pos(f.R % p)
    big = 0.0

    def val(x):                                                 # residue of a lazily reduced element in the R' domain
        return f.val(x) * pow(f.Ru, -1, p) % p

    def mul(x, y, sqr=False):
        nonlocal big
        big = max(big, abs(f.val(x)) / p, abs(f.val(y)) / p)
        return f.mul(x, y, sqr=sqr)

    def packed(x):
        y = mul(x, to_r)
        assert abs(f.val(y)) < 4 * p
        return f.to_packed(y) * pow(f.R, -1, p) % p

    for it in range(n):
        x0 = rng.randrange(1, p); v0 = rng.randrange(1, p)
        u0 = x0 * x0 * v0 % p if it % 3 else rng.randrange(1, p)
        uu, vv = sl(u0 * f.R % p), sl(v0 * f.R % p)
        a = mul(uu, vv)
        tab = [None, a]
        for k in range(2, 16):
            tab.append(mul(tab[-1], a))
        digits = [(e >> (4 * i)) & 15 for i in range(64)]
        top = max(i for i in range(64) if digits[i])
        w = tab[digits[top]]
        for i in range(top - 1, -1, -1):
            for _ in range(4):
                w = mul(w, w, sqr=True)
            if digits[i]:
                w = mul(w, tab[digits[i]])
        assert val(w) == pow(u0 * v0 % p, e, p)
        c = mul(a, mul(w, w, sqr=True)); r = mul(uu, w)
        odd = False
        steps = (S + 7) // 8
        for i in range(steps):
            wd = min(8, S - 8 * i)
            dd = c
            for _ in range(S - 8 * i - wd):
                dd = mul(dd, dd, sqr=True)
            dc = packed(dd)
            h = pow(g, 1 << (S - wd), p)                           # generator of the 2^wd-th roots
            j = next(j for j in range(1 << wd) if pow(h, j, p) == dc)
            if i == 0 and j & 1:
                odd = True
            hw = min(8, S)                                         # the device finds j by a perfect hash of the 2^hw-th roots (consts_gen.h)
            lo = (dc * f.R % p) & 0xFFFFFFFF
            assert d["SQRT_HIDX_tab"][((lo * d["SQRT_HMUL"]) & 0xFFFFFFFF) >> (32 - d["SQRT_HBITS"])] >> (hw - wd) == j
            gs = pow(g, -(j << (8 * i)), p); gh = pow(g, -((j << (8 * i)) >> 1), p) if not (i == 0 and j & 1) else 1
            c = mul(c, sl(gs * f.R % p)); r = mul(r, sl(gh * f.R % p))
        x = packed(r)
        is_sq = pow(u0 * pow(v0, -1, p) % p, (p - 1) // 2, p) == 1
        assert odd == (not is_sq)
        if is_sq:
            assert x * x % p * v0 % p == u0
    print(f"  {f.name}: sqrt(u / v) on unsaturated limbs (4-bit-window power, {S}-bit discrete log in 8-bit windows, looked up through the header's hash table) on {n} random inputs; largest operand {big:.1f} p (the inputs sliced with the shift)")


class G1Red:
    """g1r_add / g1r_dbl of fpu_g1.h: the general XYZZ addition and doubling of the fixed-base reduction kernels (k_bucket_sum,
    k_heavy_sum, k_wsum*) in the Montgomery domain R' = 2^(W L) -- every coordinate is value * R' lazily reduced, nothing scaled.
    A point is (X, Y, ZZ, ZZZ) limb lists or None (the identity flag)."""
    VB = 128.0                                               # results stay below VB * p in magnitude (the doubling of a freshly loaded
                                                             # point: V = 4 Y^2 / R' < 4 * 2^(2 SH) p / (R' / p)); a loaded (sliced)
                                                             # coordinate is value * R * 2^SH as an integer: below 2^SH p

    def __init__(self, g):
        self.g, self.f = g, g.f

    def chk(self, P, vb=None):
        f = self.f
        n = (1 << f.W) + 16
        for c in P:
            assert max(abs(v) for v in c) <= n and abs(f.val(c)) < (vb or self.VB) * f.p, (self.g.name, "reduction coordinate out of range")
        return P

    def from_sat(self, X, Y, ZZ, ZZZ, extreme=False):
        """extreme: the representative of the same residues with the largest value a load can produce (just below 2^SH p)"""
        f = self.f
        if extreme:
            top = ((1 << f.SH) - 1) * f.p
            return self.chk(tuple(f.slice_pos(((v << f.SH) % f.p) + top) for v in (X, Y, ZZ, ZZZ)), vb=float(1 << f.SH))
        return self.chk(tuple(f.slice(v, f.SH) for v in (X, Y, ZZ, ZZZ)), vb=float(1 << f.SH))

    def zero(self, v, operand):
        """fu_is_zero_mod_p2: a square's output is 0 mod p iff its limbs are 0, p's or 2p's -- valid while operand^2 / R' < 2 p
        (two loaded coordinates can differ by up to 2 * 2^(2 SH) p^2 / R' + 2p)"""
        f = self.f
        ov = abs(f.val(operand))
        assert ov * ov < f.Ru * f.p * 19 // 10, (self.g.name, "zero test out of range", ov / f.p)
        self.worst_zero = max(getattr(self, "worst_zero", 0.0), ov * ov / (f.Ru * f.p))
        p2 = f.slice_pos(2 * f.p)
        is0 = all(l == 0 for l in v) or all(l == pl for l, pl in zip(v, f.pl)) or all(l == pl for l, pl in zip(v, p2))
        assert is0 == (f.val(v) % f.p == 0)
        return is0

    def dbl(self, A):
        f = self.f
        X, Y, ZZ, ZZZ = A
        YY = f.mul(Y, Y, sqr=True)
        V = f.carry([i32(4 * v) for v in YY])
        U = [i32(2 * v) for v in Y]
        Wv = f.mul(U, V)
        S = f.mul(X, V)
        M = f.carry([i32(3 * v) for v in f.mul(X, X, sqr=True)])
        X3 = [i32(m - 2 * s) for m, s in zip(f.mul(M, M, sqr=True), S)]
        SX = f.carry([i32(s - x) for s, x in zip(S, X3)])
        Y3 = [i32(a - b) for a, b in zip(f.mul(M, SX), f.mul(Wv, Y))]
        return self.chk((f.carry(X3), Y3, f.mul(V, ZZ), f.mul(Wv, ZZZ)))

    def add(self, A, B):
        f = self.f
        if A is None: return B
        if B is None: return A
        X1, Y1, ZZ1, ZZZ1 = A
        X2, Y2, ZZ2, ZZZ2 = B
        U1, U2 = f.mul(X1, ZZ2), f.mul(X2, ZZ1)
        S1, S2 = f.mul(Y1, ZZZ2), f.mul(Y2, ZZZ1)
        P = [i32(a - b) for a, b in zip(U2, U1)]; R = [i32(a - b) for a, b in zip(S2, S1)]
        PP = f.mul(P, P, sqr=True)
        if self.zero(PP, P):
            return self.dbl(A) if self.zero(f.mul(R, R, sqr=True), R) else None
        PPP, Q = f.mul(P, PP), f.mul(U1, PP)
        ZZ3 = f.mul(f.mul(ZZ1, ZZ2), PP); ZZZ3 = f.mul(f.mul(ZZZ1, ZZZ2), PPP)
        T = f.mul(S1, PPP); RR = f.mul(R, R, sqr=True)
        X3 = [i32(r - a - 2 * b) for r, a, b in zip(RR, PPP, Q)]
        QX = f.carry([i32(a - b) for a, b in zip(Q, X3)])
        Y3 = [i32(a - b) for a, b in zip(f.mul(R, QX), T)]
        return self.chk((f.carry(X3), Y3, ZZ3, ZZZ3))

    def to_affine(self, A):
        if A is None: return None
        f = self.f
        X, Y, ZZ, ZZZ = (f.val(c) % f.p for c in A)            # the common factor R' cancels in X / ZZ and Y / ZZZ
        if ZZ == 0: return None
        assert (ZZ ** 3 - ZZZ ** 2 * f.Ru) % f.p == 0          # (zz R')^3 = (zzz R')^2 R'  <=>  zz^3 = zzz^2
        return (X * pow(ZZ, -1, f.p) % f.p, Y * pow(ZZZ, -1, f.p) % f.p)


def check_g1_red(g, rng, rounds=60):
    r, f = G1Red(g), g.f
    pts = [g.rand_point(rng) for _ in range(5)]
    sat = lambda P, ex=False: r.from_sat(P[0] * f.R % f.p, P[1] * f.R % f.p, f.R % f.p, f.R % f.p, extreme=ex)
    pool = [(sat(P), P) for P in pts] + [(sat(P, True), P) for P in pts] + [(None, None)]
    for (A, a) in list(pool[:-1]):                                                # loaded x loaded, every pair incl. the extreme representatives
        for (B, b) in list(pool[:-1]):
            assert r.to_affine(r.add(A, B)) == g.add(a, b), (g.name, "loaded x loaded")
        assert r.to_affine(r.dbl(A)) == g.add(a, a)
    cases = {"add": 0, "dbl": 0, "inverse": 0, "identity": 0}
    for i in range(rounds):
        (A, a), (B, b) = rng.choice(pool), rng.choice(pool)
        if i % 7 == 3: (B, b) = (A, a)                                            # P + P through the addition's own test
        if i % 7 == 5 and a is not None:                                          # P + (-P)
            B = (A[0], [-v for v in A[1]], A[2], A[3]); b = (a[0], (-a[1]) % f.p)
        Cc, c = r.add(A, B), g.add(a, b)
        assert r.to_affine(Cc) == c, (g.name, "general addition")
        cases["identity" if A is None or B is None else "inverse" if c is None else "dbl" if a == b else "add"] += 1
        if A is not None:
            D = r.dbl(A); assert r.to_affine(D) == g.add(a, a); pool.append((D, g.add(a, a)))
        if Cc is not None:
            pool.append((Cc, c))
        pool = pool[-24:] + [(None, None)]
    assert all(v > 0 for v in cases.values()), cases
    # the widest difference two LOADED points can produce: U <= (2^SH p)^2 / R' + p, |P| <= 2 U
    u = (1 << (2 * f.SH)) * f.p / f.Ru + 1
    assert (2 * u) ** 2 * f.p / f.Ru < 1.9, (g.name, "loaded operands leave the range of the zero test", (2 * u) ** 2 * f.p / f.Ru)
    print(f"  {g.name}: general XYZZ additions / doublings in the R' domain == affine law ({cases}); limbs within 2^W + 16, |values| < {G1Red.VB} p, incl. every pair of freshly loaded points at their largest representatives (zero test worst case {r.worst_zero:.2f} < 1.9)")


def main():
    C = parse()
    rng = random.Random(5)
    fields = {}
    print("fu_mul / fu_sqr:")
    for name, d in C.items():
        if "P" in d and name.startswith("Fq") and not d.get("FULL"):
            fields[name] = Field(name, d)
            check_mul(fields[name], rng)
    print("teu_madd:")
    for name, d in C.items():
        if name.startswith("Suite") and not d.get("SW_NATIVE") and d.get("Fq") in fields and "D" in d:
            check_te(TE(name, d, fields[d["Fq"]]), rng)
    print("per-item doubling chains:")
    for name, d in C.items():
        if name.startswith("Suite") and not d.get("SW_NATIVE") and d.get("Fq") in fields and "D" in d:
            check_te_chain(TE(name, d, fields[d["Fq"]]), rng)
    print("g1u_madd:")
    for name, d in C.items():
        if name.startswith("G1") and d.get("Fq") in fields:
            check_g1(G1(name, d, fields[d["Fq"]]), rng)
    print("fu_sqrt_ratio_nf (point decompression):")
    for name in ("FqBandersnatch", "FqBabyJubJub", "FqEd25519"):
        if name in fields and "ROOT" in C[name]:
            check_sqrt_chain(fields[name], C[name], rng)
    print("g1r_add / g1r_dbl (fixed-base reductions):")
    for name, d in C.items():
        if name.startswith("G1") and d.get("Fq") in fields:
            check_g1_red(G1(name, d, fields[d["Fq"]]), rng)
    print("all checks passed")


if __name__ == "__main__":
    main()
