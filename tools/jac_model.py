#!/usr/bin/env python3
"""Integer model of the two Jacobi-symbol loops of fp256.h (the single-bit binary form and the macro-step form with limb phases):
checks both against the Legendre symbol on random inputs and counts their steps / instructions per symbol (the figures quoted in fp256.h)."""
import random
P = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
def legendre(x, q):
    x %= q
    return 0 if x == 0 else (1 if pow(x, (q - 1) // 2, q) == 1 else -1)
def jac(a, n):
    t = 0; steps = 0
    while a != 0:
        a0 = a & 0xffffffff
        ctz = 32 if a0 == 0 else (a0 & -a0).bit_length() - 1
        z = min(ctz, 30)
        a >>= z
        nl = n & 0xffffffff
        x = nl ^ (nl >> 1)
        if z & 1: t ^= x & 2
        odd = a & 1
        lt = a < n
        sw = odd and lt
        if sw: t ^= (a & n) & 2
        if odd:
            d = n - a if lt else a - n
            if sw: n = a
            a = d
        steps += 1
    return (0 if n != 1 else (-1 if t & 2 else 1)), steps
rng = random.Random(1)
tot = 0; mx = 0
for i in range(3000):
    a = rng.randrange(P)
    r, s = jac(a, P)
    assert r == legendre(a, P)
    tot += s; mx = max(mx, s)
print("avg macro steps", tot / 3000, "max", mx)
for a in [0, 1, 2, P - 1, P - 2, 1 << 31, 1 << 32, (1 << 255) - 1 if (1<<255)-1 < P else 5, 3 << 200]:
    r, s = jac(a % P if a else 0, P)
    assert r == legendre(a, P), a
print("edge ok")

def wave_cost(vals):
    st = [(a, P, 0) for a in vals]
    cost = 0; steps = 0
    while any(a for a, n, t in st):
        k = max(((a | n).bit_length() + 31) // 32 if a else 1 for a, n, t in st)
        cost += 6 * k + 13
        steps += 1
        new = []
        for a, n, t in st:
            if a:
                a0 = a & 0xffffffff
                ctz = 32 if a0 == 0 else (a0 & -a0).bit_length() - 1
                a >>= min(ctz, 30)
                if a & 1:
                    if a < n: a, n = n - a, a
                    else: a = a - n
            new.append((a, n, t))
        st = new
    return cost, steps
cs = [wave_cost([rng.randrange(P) for _ in range(64)]) for _ in range(20)]
print("wave cost (instr per lane per symbol)", sum(c for c, s in cs) / 20, "steps", sum(s for c, s in cs) / 20)
