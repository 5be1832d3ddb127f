"""oracle/ring_py.py -- pure-Python restatement of the Ring-VRF SNARK path (TEST ORACLE, small cases).

Restates what ark-vrf delegates to the un-vendored crate `w3f-ring-proof 0.0.10` (+ w3f-plonk-common,
w3f-pcs, ark-transcript) at src/ring.rs:220 (`ring_prover.prove`), :242 (`verifier.verify`),
:404,416 (`ring_proof::index`), following the byte-exact specification in SURVEY.md Appendix A.5,
A.7, A.8, and is PINNED to the reference's ring vectors (tests/golden/*_ring.json + the two SRS
files): ring commitment and the complete 592-byte / 480-byte ring proof of all 7 vectors of both
suites (tests/test_oracle_ring.py).  Big-int Python: fine for the vectors' domain N = 512; the C
port (orc_ring.c) handles benchmark sizes.

Only tests/ import this module.
"""
import hashlib

# ---------------------------------------------------------------------------------------------
# curves


class Suite:
    pass


def _bander():
    s = Suite()
    s.name = "bandersnatch"
    s.suite_id = b"Bandersnatch-SHA512-ELL2-v1"
    s.r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001   # Fr(BLS12-381) = Fq(Bandersnatch)
    s.te_a = s.r - 5
    s.te_d = 45022363124591815672509500913686876175488063829319466900776701791074614335719
    s.te_order_bits = 253
    s.blinding_base = (23335687741101763108036518445642207119627658113885888016488710494487028845889,
                       5552214580375038693022409684979828600325210968745774080859660443337357929963)
    s.accumulator_base = (14056632001415368875257708737821299882600475929746323097150942355715730684350,
                          10322661992765989500407719465917595459409463902187386706652408883505670839210)
    s.padding = (26913883415342152801331916189968962157924271221160514298872262294143390094043,
                 30874728313203001508631936119690348239461579770372782660098261717479009115354)
    s.p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    s.g1_b = 4
    s.fp_bytes = 48
    s.zcash = True
    s.two_adicity = 32
    s.root_of_unity = 10238227357739495823651030575849232062558860180284477541189508159991286009131
    return s


def _bjj():
    s = Suite()
    s.name = "babyjubjub"
    s.suite_id = b"BabyJubJub-SHA512-TAI-v1"
    s.r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001      # Fr(BN254)
    s.te_a = 1
    s.te_d = 9706598848417545097372247223557719406784115219466060233080913168975159366771
    s.te_order_bits = 251
    s.blinding_base = (15549380791300914366206471199568039679131690710803662429646809536753521087193,
                       15218614024055502695611547593111691164731001864276292210438920202280814188379)
    s.accumulator_base = (6402374321243162085389111671722843560682527921646684137786768606010797479351,
                          9735581299071570006712034490635195155689931359428941496570758703259384062170)
    s.padding = (11167490195257431015694161063225325511805242064780376648595733691987293447528,
                 18403369502642103292159933062507105566469227524991433735553439433605496057425)
    s.p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    s.g1_b = 3
    s.fp_bytes = 32
    s.zcash = False
    s.two_adicity = 28
    s.root_of_unity = 19103219067921713944291392827692070036145651957329286315305642004821462161904
    return s


def _jubjub():
    """JubJub-SHA512-TAI-v1 (src/suites/jubjub.rs:56-95): twisted Edwards a = -1 over Fr(BLS12-381), ring proofs over BLS12-381."""
    s = _bander()
    s.name = "jubjub"
    s.suite_id = b"JubJub-SHA512-TAI-v1"
    s.te_a = s.r - 1
    s.te_d = 19257038036680949359750312669786877991949435402254120286184196891950884077233
    s.te_order_bits = 252
    s.blinding_base = (38206460563694846719174258613922853630278999941532690543235578292520143148532,
                       34254498978062207918041301829525626783549813531091321004550549786528984401675)
    s.accumulator_base = (48142684311216766702182564801462043940571084233680216669499475549492432046964,
                          34380560660182334518990118617091967209302636551264477863958902286043397647879)
    s.padding = (17348704025397475127937572481155408456556065464328870407269802701696798733683,
                 24318278422173803457621119807961883607097742387673491974779969503617097905596)
    return s


SUITES = {0: _bander(), 1: _bjj(), 2: _jubjub()}

# ---- twisted Edwards (affine, big-int)


def te_add(s, P, Q):
    (x1, y1), (x2, y2) = P, Q
    r = s.r
    dxy = s.te_d * x1 * x2 % r * y1 % r * y2 % r
    x3 = (x1 * y2 + y1 * x2) * pow(1 + dxy, -1, r) % r
    y3 = (y1 * y2 - s.te_a * x1 * x2) * pow(1 - dxy, -1, r) % r
    return (x3, y3)


def te_dbl_pow2(s, P, n):
    out = []
    for _ in range(n):
        out.append(P)
        P = te_add(s, P, P)
    return out


def te_decode(s, b):
    r = s.r
    y = int.from_bytes(b, "little")
    neg = y >> 255
    y &= (1 << 255) - 1
    x2 = (1 - y * y) * pow(s.te_a - s.te_d * y * y, -1, r) % r
    x = sqrt_mod(x2, r)
    assert x is not None
    if (x > (r - 1) // 2) != bool(neg):
        x = r - x
    return (x, y)


def sqrt_mod(a, p):
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    q, s_ = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s_ += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, rr = s_, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, rr = t * c % p, rr * b % p
    return rr

# ---- G1 (short Weierstrass y^2 = x^3 + b over Fp), Jacobian; None = infinity


def g1_dbl(p, P):
    if P is None:
        return None
    X, Y, Z = P
    if Y == 0:
        return None
    A = X * X % p; B = Y * Y % p; C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p; F = E * E % p
    X3 = (F - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def g1_add(p, P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    X1, Y1, Z1 = P; X2, Y2, Z2 = Q
    Z1Z1 = Z1 * Z1 % p; Z2Z2 = Z2 * Z2 % p
    U1 = X1 * Z2Z2 % p; U2 = X2 * Z1Z1 % p
    S1 = Y1 * Z2 % p * Z2Z2 % p; S2 = Y2 * Z1 % p * Z1Z1 % p
    if U1 == U2:
        return g1_dbl(p, P) if S1 == S2 else None
    H = (U2 - U1) % p; I = 4 * H * H % p; J = H * I % p
    rr = 2 * (S2 - S1) % p; V = U1 * I % p
    X3 = (rr * rr - J - 2 * V) % p
    Y3 = (rr * (V - X3) - 2 * S1 * J) % p
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) % p * H % p
    return (X3, Y3, Z3)


def g1_neg(p, P):
    return None if P is None else (P[0], (-P[1]) % p, P[2])


def g1_affine(p, P):
    if P is None:
        return None
    zi = pow(P[2], -1, p)
    return (P[0] * zi * zi % p, P[1] * zi * zi % p * zi % p)


def g1_mul(p, P, k):
    acc = None
    for bit in bin(k)[2:]:
        acc = g1_dbl(p, acc)
        if bit == "1":
            acc = g1_add(p, acc, P)
    return acc


def g1_msm(p, pts, scalars, c=8):
    """sum scalars[i] * pts[i]; pts affine (x, y) or None.  Plain Pippenger."""
    pairs = [(P, k) for P, k in zip(pts, scalars) if P is not None and k]
    if not pairs:
        return None
    nbits = max(k.bit_length() for _, k in pairs)
    total = None
    for w in reversed(range((nbits + c - 1) // c)):
        for _ in range(c):
            total = g1_dbl(p, total)
        buckets = [None] * (1 << c)
        for P, k in pairs:
            d = (k >> (w * c)) & ((1 << c) - 1)
            if d:
                buckets[d] = g1_add(p, buckets[d], (P[0], P[1], 1))
        run = acc = None
        for d in range((1 << c) - 1, 0, -1):
            run = g1_add(p, run, buckets[d])
            acc = g1_add(p, acc, run)
        total = g1_add(p, total, acc)
    return total

# ---- G1 codecs


def g1_encode(s, P, compressed):
    n = s.fp_bytes
    if s.zcash:                                        # BLS12-381: big-endian, flags in the first byte
        if P is None:
            out = bytearray(n if compressed else 2 * n); out[0] |= 0x40
            if compressed:
                out[0] |= 0x80
            return bytes(out)
        x, y = P
        if compressed:
            out = bytearray(x.to_bytes(n, "big")); out[0] |= 0x80
            if y > (s.p - 1) // 2:
                out[0] |= 0x20
            return bytes(out)
        return x.to_bytes(n, "big") + y.to_bytes(n, "big")
    # BN254: arkworks default SW format, little-endian, flags in the last byte
    if P is None:
        out = bytearray(n if compressed else 2 * n); out[-1] |= 0x40
        return bytes(out)
    x, y = P
    neg = 0x80 if y > (s.p - 1) // 2 else 0
    if compressed:
        out = bytearray(x.to_bytes(n, "little")); out[-1] |= neg
        return bytes(out)
    out = bytearray(x.to_bytes(n, "little") + y.to_bytes(n, "little")); out[-1] |= neg
    return bytes(out)


def g1_decode_uncompressed(s, b):
    n = s.fp_bytes
    if s.zcash:
        if b[0] & 0x40:
            return None
        return (int.from_bytes(b[:n], "big"), int.from_bytes(b[n:2 * n], "big"))
    if b[2 * n - 1] & 0x40:
        return None
    yb = bytearray(b[n:2 * n]); yb[-1] &= 0x3f
    return (int.from_bytes(b[:n], "little"), int.from_bytes(yb, "little"))


def g1_decode_compressed(s, b):
    n = s.fp_bytes
    p = s.p
    if s.zcash:
        if b[0] & 0x40:
            return None
        big = b[0] & 0x20
        xb = bytearray(b); xb[0] &= 0x1f
        x = int.from_bytes(xb, "big")
    else:
        if b[-1] & 0x40:
            return None
        big = b[-1] & 0x80
        xb = bytearray(b); xb[-1] &= 0x3f
        x = int.from_bytes(xb, "little")
    y = sqrt_mod((x * x * x + s.g1_b) % p, p)
    assert y is not None
    if (y > (p - 1) // 2) != bool(big):
        y = p - y
    return (x, y)


class Srs:
    """URS { powers_in_g1: Vec<G1>, powers_in_g2: Vec<G2> }, serialize_uncompressed."""

    def __init__(self, s, data):
        n = s.fp_bytes
        cnt = int.from_bytes(data[:8], "little")
        off = 8
        self.g1 = [g1_decode_uncompressed(s, data[off + 2 * n * i: off + 2 * n * (i + 1)]) for i in range(cnt)]
        off += 2 * n * cnt
        cnt2 = int.from_bytes(data[off: off + 8], "little")
        off += 8
        self.g2_raw = [data[off + 4 * n * i: off + 4 * n * (i + 1)] for i in range(cnt2)]
        assert off + 4 * n * cnt2 == len(data)

# ---------------------------------------------------------------------------------------------
# polynomials over Fr


def domain_root(s, n):
    lg = n.bit_length() - 1
    assert 1 << lg == n
    return pow(s.root_of_unity, 1 << (s.two_adicity - lg), s.r)


def fft(s, a, w):
    """evaluations of the polynomial with coefficients a on {w^i}; len(a) a power of two."""
    r = s.r
    n = len(a)
    a = list(a)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w, n // length, r)
        for i in range(0, n, length):
            x = 1
            for k in range(length // 2):
                u, v = a[i + k], a[i + k + length // 2] * x % r
                a[i + k], a[i + k + length // 2] = (u + v) % r, (u - v) % r
                x = x * wl % r
        length <<= 1
    return a


def ifft(s, ev, w):
    r = s.r
    n = len(ev)
    out = fft(s, ev, pow(w, -1, r))
    ninv = pow(n, -1, r)
    return [x * ninv % r for x in out]


def poly_eval(s, c, x):
    acc = 0
    for v in reversed(c):
        acc = (acc * x + v) % s.r
    return acc


def poly_div_linear(s, c, z):
    """(c(X) - c(z)) / (X - z) by synthetic division."""
    r = s.r
    out = [0] * (len(c) - 1)
    acc = 0
    for i in range(len(c) - 1, 0, -1):
        acc = (c[i] + acc * z) % r
        out[i - 1] = acc
    return out

# ---------------------------------------------------------------------------------------------
# ark-transcript over SHAKE128


class ArkTranscript:
    def __init__(self, label):
        self.h = hashlib.shake_128()
        self.len = None
        self.label(label)

    def write(self, b):
        self.h.update(b)
        self.len = (self.len or 0) + len(b)

    def separate(self):
        if self.len is not None:
            self.h.update(self.len.to_bytes(4, "big"))
            self.len = None

    def label(self, l):
        self.separate(); self.write(l); self.separate()

    def append(self, data):
        self.separate(); self.write(data); self.separate()

    def challenge_bytes(self, l, n):
        self.label(l)
        self.write(b"challenge")
        out = self.h.copy().digest(n)
        self.separate()
        return out

    def challenge_fr(self, s, l):
        return int.from_bytes(self.challenge_bytes(l, 48), "big") % s.r

# ---------------------------------------------------------------------------------------------
# PIOP parameters, index, prover, verifier


class Params:
    def __init__(self, s, ring_size=None, domain_size=None):
        L = s.te_order_bits
        if domain_size is None:
            need = ring_size + 4 + L                      # src/ring.rs:810-821
            domain_size = 1 << (need - 1).bit_length()
        self.s = s
        self.N = domain_size
        self.L = L
        self.capacity = self.N - 3                        # ZK_ROWS = 3
        self.keyset_part_size = self.capacity - L - 1
        self.w = domain_root(s, self.N)
        self.w4 = domain_root(s, 4 * self.N)
        self.h_pows = te_dbl_pow2(s, s.blinding_base, L)  # 2^i * H


def index(prm, srs, keys):
    """ring_proof::index (src/ring.rs:404,416): fixed columns + their KZG commitments."""
    s = prm.s
    assert len(keys) <= prm.keyset_part_size
    points = list(keys) + [s.padding] * (prm.keyset_part_size - len(keys)) + prm.h_pows
    assert len(points) == prm.capacity - 1
    xs = [P[0] for P in points] + [0] * (prm.N - len(points))
    ys = [P[1] for P in points] + [0] * (prm.N - len(points))
    sel = [1] * prm.keyset_part_size + [0] * (prm.N - prm.keyset_part_size)
    cols = {"points": points, "x": xs, "y": ys, "sel": sel}
    cols["x_poly"] = ifft(s, xs, prm.w); cols["y_poly"] = ifft(s, ys, prm.w); cols["sel_poly"] = ifft(s, sel, prm.w)
    commit = lambda c: g1_affine(s.p, g1_msm(s.p, srs.g1[: len(c)], c))
    cols["C"] = [commit(cols["x_poly"]), commit(cols["y_poly"]), commit(cols["sel_poly"])]
    return cols


def commitment_bytes(s, cols):
    return b"".join(g1_encode(s, C, True) for C in cols["C"])


def _transcript_prelude(prm, srs, cols):
    s = prm.s
    t = ArkTranscript(s.suite_id)
    t.label(b"vk")
    vk = g1_encode(s, srs.g1[0], False) + srs.g2_raw[0] + srs.g2_raw[1] + b"".join(g1_encode(s, C, False) for C in cols["C"])
    t.append(vk)
    return t


def _le32(x):
    return x.to_bytes(32, "little")


def prove(prm, srs, cols, key_index, blinding_scalar):
    """RingProver::prove with blinding disabled (src/ring.rs:220, 273-275).  Returns proof bytes."""
    s = prm.s
    r, N, cap, w = s.r, prm.N, prm.capacity, prm.w
    points, sel = cols["points"], cols["sel"]
    # -- witness (A.7 step 1)
    bits = [0] * (cap - 1)
    bits[key_index] = 1
    for i in range(prm.L):
        bits[prm.keyset_part_size + i] = (blinding_scalar >> i) & 1
    ip = [0] * cap
    for i in range(cap - 1):
        ip[i + 1] = (ip[i] + sel[i] * bits[i]) % r
    acc = [s.accumulator_base]
    for i in range(cap - 1):
        acc.append(te_add(s, acc[i], points[i]) if bits[i] else acc[i])
    result = acc[cap - 1]
    neg_seed = ((-s.accumulator_base[0]) % r, s.accumulator_base[1])
    instance = te_add(s, result, neg_seed)                          # Yb = pk_k + b*H
    pad = lambda v: list(v) + [0] * (N - len(v))                    # private column, blinding disabled
    ev = {"bits": pad(bits), "ip": pad(ip), "ax": pad([P[0] for P in acc]), "ay": pad([P[1] for P in acc])}
    poly = {k: ifft(s, v, w) for k, v in ev.items()}
    poly["px"], poly["py"], poly["sel"] = cols["x_poly"], cols["y_poly"], cols["sel_poly"]
    commit = lambda c: g1_affine(s.p, g1_msm(s.p, srs.g1[: len(c)], c))
    C = [commit(poly[k]) for k in ("bits", "ip", "ax", "ay")]
    # -- transcript
    t = _transcript_prelude(prm, srs, cols)
    t.label(b"instance"); t.append(_le32(instance[0]) + _le32(instance[1]))
    t.label(b"committed_cols"); t.append(b"".join(g1_encode(s, c, False) for c in C))
    alphas = [t.challenge_fr(s, b"constraints_aggregation") for _ in range(7)]
    # -- constraints on the 4N domain (A.7 step 4)
    M = 4 * N
    w4 = prm.w4
    e4 = {k: fft(s, poly[k] + [0] * (M - N), w4) for k in poly}
    sh = lambda v, i: v[(i + 4) % M]                                # c(wX) on the 4N domain
    xs4 = [1] * M
    for i in range(1, M):
        xs4[i] = xs4[i - 1] * w4 % r
    w_last = pow(w, cap - 1, r)
    # Lagrange basis polynomials of rows 0 and cap-1 over the size-N domain
    lf = ifft(s, [1] + [0] * (N - 1), w)
    ll_ev = [0] * N; ll_ev[cap - 1] = 1
    ll = ifft(s, ll_ev, w)
    lf4 = fft(s, lf + [0] * (M - N), w4); ll4 = fft(s, ll + [0] * (M - N), w4)
    a = s.te_a
    seed = s.accumulator_base
    agg = [0] * M
    for i in range(M):
        b = e4["bits"][i]; x1 = e4["ax"][i]; y1 = e4["ay"][i]; x2 = e4["px"][i]; y2 = e4["py"][i]
        x3 = sh(e4["ax"], i); y3 = sh(e4["ay"], i)
        nl = (xs4[i] - w_last) % r
        c0 = (sh(e4["ip"], i) - e4["ip"][i] - e4["sel"][i] * b) % r * nl % r
        c1 = (b * (x3 * ((y1 * y2 + a * x1 % r * x2) % r) - x1 * y1 - x2 * y2) + (1 - b) * (x3 - x1)) % r * nl % r
        c2 = (b * (y3 * ((x1 * y2 - x2 * y1) % r) - x1 * y1 + x2 * y2) + (1 - b) * (y3 - y1)) % r * nl % r
        c3 = b * (1 - b) % r
        c4 = (lf4[i] * (x1 - seed[0]) + ll4[i] * (x1 - result[0])) % r
        c5 = (lf4[i] * (y1 - seed[1]) + ll4[i] * (y1 - result[1])) % r
        c6 = (lf4[i] * e4["ip"][i] + ll4[i] * (e4["ip"][i] - 1)) % r
        agg[i] = (alphas[0] * c0 + alphas[1] * c1 + alphas[2] * c2 + alphas[3] * c3 + alphas[4] * c4 + alphas[5] * c5 + alphas[6] * c6) % r
    aggc = ifft(s, agg, w4)
    # * prod_{i=N-3}^{N-1} (X - w^i), / (X^N - 1)
    num = aggc + [0, 0, 0, 0]
    for i in (N - 3, N - 2, N - 1):
        z = pow(w, i, r)
        nxt = [0] * len(num)
        for k in range(len(num) - 1):
            nxt[k + 1] = (nxt[k + 1] + num[k]) % r
            nxt[k] = (nxt[k] - z * num[k]) % r
        num = nxt
    while num and num[-1] == 0:
        num.pop()
    q = [0] * (len(num) - N)
    rem = list(num)
    for k in range(len(num) - 1, N - 1, -1):                         # divide by X^N - 1
        q[k - N] = rem[k]
        rem[k - N] = (rem[k - N] + rem[k]) % r
        rem[k] = 0
    assert not any(rem), "quotient not exact"
    Cq = commit(q)
    t.label(b"quotient"); t.append(g1_encode(s, Cq, False))
    zeta = t.challenge_fr(s, b"evaluation_point")
    order = ("px", "py", "sel", "bits", "ip", "ax", "ay")
    evals = [poly_eval(s, poly[k], zeta) for k in order]
    t.label(b"register_evaluations"); t.append(b"".join(_le32(v) for v in evals))
    x2, y2, _, b, _, x1, y1 = evals
    nl_z = (zeta - w_last) % r
    k1 = (b * ((y1 * y2 + a * x1 % r * x2) % r) + 1 - b) % r
    k2 = (b * ((x1 * y2 - x2 * y1) % r) + 1 - b) % r
    lin = [nl_z * ((alphas[0] * poly["ip"][i] + alphas[1] * k1 % r * poly["ax"][i] + alphas[2] * k2 % r * poly["ay"][i]) % r) % r for i in range(N)]
    zw = zeta * w % r
    lin_zw = poly_eval(s, lin, zw)
    t.label(b"shifted_linearization_evaluation"); t.append(_le32(lin_zw))
    nus = [t.challenge_fr(s, b"kzg_aggregation") for _ in range(8)]
    aggz = [0] * len(q)
    for nu, k in zip(nus, order):
        for i, v in enumerate(poly[k]):
            aggz[i] = (aggz[i] + nu * v) % r
    for i, v in enumerate(q):
        aggz[i] = (aggz[i] + nus[7] * v) % r
    pi1 = commit(poly_div_linear(s, aggz, zeta))
    pi2 = commit(poly_div_linear(s, lin, zw))
    proof = b"".join(g1_encode(s, c, True) for c in C) + b"".join(_le32(v) for v in evals) + g1_encode(s, Cq, True) + \
        _le32(lin_zw) + g1_encode(s, pi1, True) + g1_encode(s, pi2, True)
    return proof, instance


# ---------------------------------------------------------------------------------------------
# verifier: RingVerifier::verify, src/ring.rs:242

def _g1_lincomb(s, terms):
    """sum k_i * P_i for [(k, affine P)]"""
    acc = None
    for k, Pt in terms:
        if Pt is None or k % s.r == 0:
            continue
        acc = g1_add(s.p, acc, g1_mul(s.p, (Pt[0], Pt[1], 1), k % s.r))
    return acc


def verify(prm, srs, fixed_commitments, proof, instance, pairing=None):
    """Returns True iff the ring proof verifies for `instance` (the Pedersen key commitment Yb as a TE
    point) under the verifier key (srs.g1[0], srs.g2[0..2], fixed_commitments)."""
    from . import pairing_py as PP
    s = prm.s
    PP.use_curve("bls12_381" if s.zcash else "bn254")
    g2_decode = PP.g2_decode_zcash_uncompressed if s.zcash else PP.g2_decode_arkworks_uncompressed
    r, N, cap, w = s.r, prm.N, prm.capacity, prm.w
    n = s.fp_bytes
    off = 0
    C = []
    for _ in range(4):
        C.append(g1_decode_compressed(s, proof[off: off + n])); off += n
    evals = [int.from_bytes(proof[off + 32 * i: off + 32 * i + 32], "little") for i in range(7)]; off += 32 * 7
    Cq = g1_decode_compressed(s, proof[off: off + n]); off += n
    lin_zw = int.from_bytes(proof[off: off + 32], "little"); off += 32
    pi1 = g1_decode_compressed(s, proof[off: off + n]); off += n
    pi2 = g1_decode_compressed(s, proof[off: off + n]); off += n
    assert off == len(proof)
    if any(v >= r for v in evals) or lin_zw >= r:
        return False
    cols = {"C": fixed_commitments}
    t = _transcript_prelude(prm, srs, cols)
    t.label(b"instance"); t.append(_le32(instance[0]) + _le32(instance[1]))
    t.label(b"committed_cols"); t.append(b"".join(g1_encode(s, c, False) for c in C))
    alphas = [t.challenge_fr(s, b"constraints_aggregation") for _ in range(7)]
    t.label(b"quotient"); t.append(g1_encode(s, Cq, False))
    zeta = t.challenge_fr(s, b"evaluation_point")
    t.label(b"register_evaluations"); t.append(b"".join(_le32(v) for v in evals))
    t.label(b"shifted_linearization_evaluation"); t.append(_le32(lin_zw))
    nus = [t.challenge_fr(s, b"kzg_aggregation") for _ in range(8)]
    x2, y2, sel, b, ip, x1, y1 = evals
    a = s.te_a
    w_last = pow(w, cap - 1, r)
    nl = (zeta - w_last) % r
    zn1 = (pow(zeta, N, r) - 1) % r
    ninv = pow(N, -1, r)
    lag = lambda i: pow(w, i, r) * zn1 % r * ninv % r * pow((zeta - pow(w, i, r)) % r, -1, r) % r
    lf, ll = lag(0), lag(cap - 1)
    seed = s.accumulator_base
    res = te_add(s, seed, instance)
    rest = [
        (-ip - sel * b) % r * nl % r,
        (b * (-x1 * y1 - x2 * y2) - (1 - b) * x1) % r * nl % r,
        (b * (-x1 * y1 + x2 * y2) - (1 - b) * y1) % r * nl % r,
        b * (1 - b) % r,
        (lf * (x1 - seed[0]) + ll * (x1 - res[0])) % r,
        (lf * (y1 - seed[1]) + ll * (y1 - res[1])) % r,
        (lf * ip + ll * (ip - 1)) % r,
    ]
    agg_z = (sum(al * c for al, c in zip(alphas, rest)) + lin_zw) % r
    zk = 1
    for i in (N - 3, N - 2, N - 1):
        zk = zk * (zeta - pow(w, i, r)) % r
    q_z = agg_z * zk % r * pow(zn1, -1, r) % r
    col_commits = list(fixed_commitments) + C                      # px, py, sel, bits, ip, ax, ay
    C_agg = _g1_lincomb(s, [(nu, c) for nu, c in zip(nus, col_commits)] + [(nus[7], Cq)])
    v_agg = (sum(nu * e for nu, e in zip(nus, evals)) + nus[7] * q_z) % r
    k1 = (b * ((y1 * y2 + a * x1 % r * x2) % r) + 1 - b) % r
    k2 = (b * ((x1 * y2 - x2 * y1) % r) + 1 - b) % r
    C_lin = _g1_lincomb(s, [(nl * alphas[0] % r, C[1]), (nl * alphas[1] % r * k1 % r, C[2]), (nl * alphas[2] % r * k2 % r, C[3])])
    g1 = srs.g1[0]
    g2 = g2_decode(srs.g2_raw[0]); tg2 = g2_decode(srs.g2_raw[1])
    zw = zeta * w % r
    ok = True
    for Cm, z, v, pi in ((C_agg, zeta, v_agg, pi1), (C_lin, zw, lin_zw, pi2)):
        # e(C - v g1 + z pi, g2) * e(-pi, tau g2) == 1
        lhs = g1_add(s.p, Cm, g1_neg(s.p, g1_mul(s.p, (g1[0], g1[1], 1), v)))
        lhs = g1_add(s.p, lhs, g1_mul(s.p, (pi[0], pi[1], 1), z))
        lhs_a = g1_affine(s.p, lhs)
        npi = (pi[0], (-pi[1]) % s.p)
        ok &= PP.pairing_product_is_one([(lhs_a, g2), (npi, tg2)])
    return ok


# ---------------------------------------------------------------------------------------------
# RingSetup::from_seed (src/ring.rs:359-374) -- **PARITY UNPINNED**: the reference holds no vector for a seeded setup.
# Restated from the published code of the crates the reference calls:
#   from_seed:   t = S::Transcript::new(SUITE_ID); t.absorb_raw(seed); rng = t.to_rng()   (src/utils/transcript.rs:61-92: every draw is the
#                next bytes of the squeeze stream -- next_u32 4 bytes LE, next_u64 8 bytes LE)
#   from_rand:   Kzg::setup(pcs_domain_size - 1, rng)  ->  w3f-pcs URS::generate(n1 = pcs_domain_size, n2 = 2, rng):
#                tau = Fr::rand(rng); g1 = G1::rand(rng); g2 = G2::rand(rng); powers tau^i g1 (i < n1), tau^i g2 (i < 2)
#   ark-ff  Fp::rand:   loop { N x next_u64 little-endian limbs, the top limb masked to the modulus' bit length, taken AS THE MONTGOMERY
#                       REPRESENTATION (the element is limbs * R^-1); accept if limbs < p }
#   ark-ff  Fp2::rand:  c0 = Fp::rand, then c1 = Fp::rand
#   ark-ec  Projective::rand (short Weierstrass): loop { x = BaseField::rand; greatest = rng.gen::<bool>() (rand 0.8: top bit of
#                       next_u32); if x^3 + b is a square: y = the larger / smaller root by the field's Ord (Fp: canonical integers;
#                       Fp2: c1 first, then c0); return (x, y) * COFACTOR }

def seed_stream(s, seed, n):
    """first n bytes of the squeeze stream of HashTranscript<Sha512>::new(SUITE_ID) after absorb_raw(seed)
    (DigestXof: d = H(absorbed); block_i = H(d || LE64(i)), src/utils/transcript.rs:227-274)"""
    import hashlib
    d = hashlib.sha512(s.suite_id + seed).digest()
    out, i = b"", 0
    while len(out) < n:
        out += hashlib.sha512(d + i.to_bytes(8, "little")).digest(); i += 1
    return out[:n]


class _Draw:
    def __init__(self, s, seed):
        self.s, self.seed, self.pos, self.buf = s, seed, 0, b""

    def take(self, n):
        if self.pos + n > len(self.buf):
            self.buf = seed_stream(self.s, self.seed, max(4096, 2 * (self.pos + n)))
        b = self.buf[self.pos: self.pos + n]; self.pos += n
        return b

    def fp(self, p):
        """ark-ff Fp::rand"""
        limbs = (p.bit_length() + 63) // 64
        shave = 64 * limbs - p.bit_length()
        while True:
            v = int.from_bytes(self.take(8 * limbs), "little") & ((1 << (64 * limbs - shave)) - 1)
            if v < p:
                return v * pow(1 << (64 * limbs), -1, p) % p          # the limbs are the Montgomery representation

    def boolean(self):
        return int.from_bytes(self.take(4), "little") >> 31 == 1


G1_COFACTOR = {"bls12_381": 0x396c8c005555e1568c00aaab0000aaab, "bn254": 1}
G2_COFACTOR = {"bls12_381": 0x5d543a95414e7f1091d50792876a202cd91de4547085abaa68a205b2e5a7ddfa628f1cb4d9e82ef21537e293a6691ae1616ec6e786f0c70cf1c38e31c7238e5,
               "bn254": 0x30644e72e131a029b85045b68181585e06ceecda572a2489345f2299c0f9fa8d}


def _f2_sqrt(a, p):
    """a square root of a = (a0, a1) in Fp[u]/(u^2 + 1), p = 3 mod 4, or None"""
    a0, a1 = a
    if a1 == 0:
        r = sqrt_mod(a0, p)
        if r is not None:
            return (r, 0)
        r = sqrt_mod((-a0) % p, p)
        return None if r is None else (0, r)
    n = sqrt_mod((a0 * a0 + a1 * a1) % p, p)
    if n is None:
        return None
    inv2 = pow(2, -1, p)
    for d in ((a0 + n) * inv2 % p, (a0 - n) * inv2 % p):
        x0 = sqrt_mod(d, p)
        if x0 is not None and x0 != 0:
            x1 = a1 * pow(2 * x0, -1, p) % p
            if ((x0 * x0 - x1 * x1) % p, 2 * x0 * x1 % p) == (a0 % p, a1 % p):
                return (x0, x1)
    return None


def srs_params_from_seed(s, seed):
    """(tau, g1 affine, g2 affine on the twist) of RingSetup::from_seed(_, seed)"""
    import oracle.pairing_py as pp
    curve = "bls12_381" if s.fp_bytes == 48 else "bn254"
    pp.use_curve(curve)
    p = s.p
    dr = _Draw(s, seed)
    tau = dr.fp(s.r)
    while True:                                                     # G1::rand
        x = dr.fp(p); greatest = dr.boolean()
        y = sqrt_mod((x * x * x + s.g1_b) % p, p)
        if y is None:
            continue
        lo, hi = min(y, p - y), max(y, p - y)
        g1 = g1_affine(p, g1_mul(p, (x, hi if greatest else lo, 1), G1_COFACTOR[curve]))
        break
    xi = (pp.XI0, 1)
    b2 = pp.f2_mul((s.g1_b, 0), xi) if pp.MTWIST else pp.f2_mul((s.g1_b, 0), pp.f2_inv(xi))      # twist coefficient b' = b xi or b / xi
    while True:                                                     # G2::rand
        x = (dr.fp(p), dr.fp(p)); greatest = dr.boolean()
        y = _f2_sqrt(pp.f2_add(pp.f2_mul(pp.f2_mul(x, x), x), b2), p)
        if y is None:
            continue
        neg = ((-y[0]) % p, (-y[1]) % p)
        larger = y if (y[1], y[0]) > (neg[1], neg[0]) else neg
        smaller = neg if larger is y else y
        g2 = pp.g2_mul((x, larger if greatest else smaller), G2_COFACTOR[curve])
        break
    return tau, g1, g2


def srs_from_seed(s, ring_size, seed, n_g1=None):
    """serialize_uncompressed(URS) of RingSetup::from_seed(ring_size, seed): n_g1 powers tau^i g1 (default: the setup's pcs domain
    size 3 N + 1) and the two G2 powers"""
    import oracle.pairing_py as pp
    tau, g1, g2 = srs_params_from_seed(s, seed)
    if n_g1 is None:
        n = 1 << (ring_size + 4 + s.te_order_bits - 1).bit_length()
        n_g1 = 3 * n + 1
    out = n_g1.to_bytes(8, "little")
    pt, t = (g1[0], g1[1], 1), 1
    for i in range(n_g1):
        out += g1_encode(s, g1_affine(s.p, g1_mul(s.p, pt, t)) if t != 1 else g1, False)
        t = t * tau % s.r
    out += (2).to_bytes(8, "little")
    enc = pp.g2_encode_zcash_uncompressed if s.zcash else pp.g2_encode_arkworks_uncompressed
    out += enc(g2) + enc(pp.g2_mul(g2, tau))
    return out
