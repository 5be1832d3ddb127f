#!/usr/bin/env python3
"""Generates ark_vrf_amd/csrc/mont8_asm_gen.h: the 8- and 12-limb Montgomery multiplications (product scanning with
interleaved reduction, see mac96.h) as ONE inline-asm block per field.  The column accumulators live in a block of consecutive
VGPRs A[0..2N]: an EVEN column k accumulates in place in the 64-bit aligned pair (A[k], A[k+1]) with its carries in A[k+2], so
its result limb simply stays where it is and the next column starts one register up -- no shifts and no moves between columns
(the C++ form, whose accumulator is a uint64_t + uint32_t, spends ~8 move / shift instructions per column).  gfx950 wants
64-bit VGPR operands on even registers, so an ODD column accumulates in an aligned scratch triple and is folded into
(A[k], A[k+1], A[k+2]) by three add-with-carry instructions.  The modulus limbs live in SGPRs (a VOP3 instruction on gfx9
takes no 32-bit literal).

  python tools/gen_mont_asm.py          # rewrites the header from consts_gen.h
  python tools/gen_mont_asm.py --check  # also runs the generated instruction streams in a small emulator against
                                        # a * b * 2^(-32 N) mod p on random operands

Registers (all declared as clobbers): A = v[A0 : A0 + 2N], odd-column scratch T (4), m (N), modulus s[36 : 36 + N), -p^-1 next,
carries through vcc.  An asm statement takes at most 30 operands: the 12-limb form (12 + 12 inputs) returns its result as six
64-bit operands written by v_mov_b64.
"""
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONSTS = os.path.join(ROOT, "ark_vrf_amd", "csrc", "consts_gen.h")
OUT = os.path.join(ROOT, "ark_vrf_amd", "csrc", "mont8_asm_gen.h")
S0 = 36
ONES_BY_SUBTRACTION = False      # see mac_ones
# per limb count: first accumulator register (even), odd-column scratch (even), m registers
# squaring only: D = the limbs of 2a, L = the limbs a_j << 1 (no carry-in)
LAYOUT = {8: dict(A0=96, T=114, M0=118, D0=80, L0=87), 12: dict(A0=100, T=126, M0=130, D0=142, L0=154)}


def fields(N):
    txt = open(CONSTS).read()
    out = []
    for m in re.finditer(r"struct (F[qr]\w+) \{(.*?)\n\};", txt, re.S):
        name, body = m.group(1), m.group(2)
        pm = re.search(r"uint32_t P\[(\d+)\] = \{([^}]*)\}", body)
        nm = re.search(r"uint32_t NINV = (0x[0-9a-fA-F]+)u", body)
        if not pm or not nm or int(pm.group(1)) != N:
            continue
        limbs = [int(x.strip().rstrip("u"), 16) for x in pm.group(2).split(",")]
        if limbs[-1] >> 31:
            continue          # top bit of the modulus set (secp256r1): t < 2p needs a 257th bit -- those fields take mac96.h's C++ form
        out.append((name, limbs, int(nm.group(1), 16)))
    return out


def body(N, sqr=False, limbs=None, ninv=None):
    """the instruction list.  N = 8: %0..%7 = t (out, 32-bit), %8.. = a, %16.. = b.  N = 12: %0..%5 = t (out, 64-bit pairs),
    %6.. = a, %18.. = b.
    sqr: a only.  a^2 = sum_i a_i^2 B^2i + sum_i a_i B^i (2 A_>i) with A_>i = sum_{j>i} a_j B^j: the limbs of 2 A_>i are
    L_{i+1} = a_{i+1} << 1 at j = i + 1 and D_j = (a_j << 1) | (a_{j-1} >> 31) above (limb N is a_{N-1} >> 31 = 0: the top bit
    of the modulus is clear), so the N (N - 1) / 2 cross products are single multiply-adds: N (N + 1) / 2 products instead of N^2.

    limbs / ninv given: the stream is specialised to the modulus.  P[0] = 1 (hence -p^-1 = 2^32 - 1; the BLS12-381 scalar field
    = the Bandersnatch / JubJub base field): m_k = -lo is one subtraction instead of a multiplication, and m_k * P[0] = m_k only
    has to turn the column's low word into zero with the carry [lo != 0] -- a compare and two add-with-carry instead of a
    quarter-rate v_mad_u64_u32 + v_mul_lo_u32.  (P[1] = 2^32 - 1 of the same modulus is NOT special-cased: m * (2^32 - 1) =
    (m << 32) - m into a 96-bit accumulator is five full-rate carry instructions against one multiply-add + one carry, a wash at
    the measured 4.5 : 1 issue cost of v_mad_u64_u32 : v_add -- profiles/r2_ubench.txt.)"""
    p0_is_one = limbs is not None and limbs[0] == 1 and ninv == 0xffffffff
    lay = LAYOUT[N]
    A0, T_BASE, M0 = lay["A0"], lay["T"], lay["M0"]
    n_out = N if N == 8 else N // 2
    L = []
    A = lambda i: f"v{A0 + i}"
    pair = lambda k: f"v[{A0 + k}:{A0 + k + 1}]"
    M = lambda i: f"v{M0 + i}"
    P = lambda i: f"s{S0 + i}"
    T0, T1, T2, TMP = f"v{T_BASE}", f"v{T_BASE + 1}", f"v{T_BASE + 2}", f"v{T_BASE + 3}"
    TP = f"v[{T_BASE}:{T_BASE + 1}]"

    # registers of the accumulator block that hold a value: a word nothing has been written to yet reads as the literal 0 and
    # is never cleared by a v_mov (a column's first product into a fresh 64-bit pair takes the addend 0 and cannot carry, the
    # first carry into a fresh third word is written as 0 + 0 + vcc): 4-5 instructions fewer per odd column, 1 per even one
    written = set()
    src = lambda r: r if r in written else "0"

    def addc(dst, x, y):                                   # dst = x + y + vcc; a literal goes first (VOP2 wants a VGPR second)
        if y == "0":
            x, y = y, x
        L.append(f"v_addc_co_u32 {dst}, vcc, {x}, {y}, vcc")
        written.add(dst)

    def mac(k, x, y):
        lo, hi, c2, pr = (A(k), A(k + 1), A(k + 2), pair(k)) if k % 2 == 0 else (T0, T1, T2, TP)
        assert (lo in written) == (hi in written)
        fresh = lo not in written
        L.append(f"v_mad_u64_u32 {pr}, vcc, {x}, {y}, {'0' if fresh else pr}")
        written.update((lo, hi))
        if not fresh:
            addc(c2, "0", src(c2))

    def mac_ones(k, m):
        """column += m * (2^32 - 1) without the multiplier: subtract m at the column's low word (borrow through the two words
        above), add it one word up.  Five full-rate instructions for a quarter-rate multiply-add and its carry: the kernels are
        OFF: measured slower (tools/r3_run12.sh, Bandersnatch base field, 112 multiply-adds + 205 other instructions against
        120 + 173): k_accumulate 0.462 ms against 0.439 ms, k_thin_verify 3.03-3.12 ms against 2.89 ms -- the kernels are bound by
        the total issue time of the stream, not by the multiplier alone."""
        lo, hi, c2 = (A(k), A(k + 1), A(k + 2)) if k % 2 == 0 else (T0, T1, T2)
        assert lo in written and hi in written
        L.append(f"v_sub_co_u32 {lo}, vcc, {lo}, {m}")
        L.append(f"v_subbrev_co_u32 {hi}, vcc, 0, {hi}, vcc")
        L.append(f"v_subbrev_co_u32 {c2}, vcc, 0, {src(c2)}, vcc"); written.add(c2)
        L.append(f"v_add_co_u32 {hi}, vcc, {hi}, {m}")
        addc(c2, "0", c2)

    av = lambda i: f"%{n_out + i}"
    if sqr:
        D = lambda j: f"v{lay['D0'] + j}"
        LL = lambda j: f"v{lay['L0'] + j}"
        for j in range(1, N):
            L.append(f"v_lshlrev_b32 {LL(j)}, 1, {av(j)}")
            if j >= 2:
                L.append(f"v_alignbit_b32 {D(j)}, {av(j)}, {av(j - 1)}, 31")
    for k in range(2 * N - 1):
        if k % 2:
            written.difference_update((T0, T1, T2))
        lo, hi = (0, k) if k < N else (k - N + 1, N - 1)
        for i in range(lo, hi + 1):
            j = k - i
            if not sqr:
                mac(k, av(i), f"%{n_out + N + j}")
            elif i == j:
                mac(k, av(i), av(i))
            elif i < j:
                mac(k, av(i), LL(j) if j == i + 1 else D(j))
        for i in range(lo, min(hi, k - 1) + 1):
            if ONES_BY_SUBTRACTION and limbs is not None and limbs[k - i] == 0xffffffff:
                mac_ones(k, M(i))
            else:
                mac(k, M(i), P(k - i))

        def fold():
            assert A(k) in written and T0 in written
            L.append(f"v_add_co_u32 {A(k)}, vcc, {A(k)}, {T0}")
            addc(A(k + 1), src(A(k + 1)), T1)
            addc(A(k + 2), src(A(k + 2)), src(T2))
        if k < N and p0_is_one:
            if k % 2:
                fold()                                                   # the odd column's sum first: its low word is A(k)
            L.append(f"v_sub_u32 {M(k)}, 0, {A(k)}")                      # m_k = -lo mod 2^32
            L.append(f"v_cmp_ne_u32 vcc, 0, {A(k)}")                      # lo + m_k = 2^32 [lo != 0]: the word is done, its carry moves up
            addc(A(k + 1), "0", src(A(k + 1)))
            addc(A(k + 2), "0", src(A(k + 2)))
            continue
        if k < N:
            if k % 2 == 0:
                L.append(f"v_mul_lo_u32 {M(k)}, {A(k)}, s{S0 + N}")
            else:
                L.append(f"v_add_u32 {TMP}, {A(k)}, {T0}")
                L.append(f"v_mul_lo_u32 {M(k)}, {TMP}, s{S0 + N}")
            mac(k, M(k), P(0))
        if k % 2:
            fold()
    assert all(A(N + i) in written for i in range(N))
    # (the conditional subtraction of p stays in C++: done here -- the modulus limbs copied to VGPRs, because a carry instruction
    # already reads vcc over the constant bus, then v_sub / v_subb / v_cndmask -- it measured no faster: k_accumulate 0.445 ms
    # against 0.441, tools/r3_run12.sh)
    if N == 8:
        for i in range(N):
            L.append(f"v_mov_b32 %{i}, {A(N + i)}")
    else:
        for i in range(N // 2):
            L.append(f"v_mov_b64 %{i}, {pair(N + 2 * i)}")
    return L


def clobbers(N, sqr=False):
    lay = LAYOUT[N]
    extra = [f"v{lay['D0'] + j}" for j in range(2, N)] + [f"v{lay['L0'] + j}" for j in range(1, N)] if sqr else []
    return ([f"v{lay['A0'] + i}" for i in range(2 * N + 1)] + [f"v{lay['T'] + i}" for i in range(4)] + [f"v{lay['M0'] + i}" for i in range(N)]
            + extra + [f"s{S0 + i}" for i in range(N + 1)] + ["vcc"])


def written_registers(ins):
    """every register an instruction of the stream writes (destination operands incl. both halves of a pair, vcc)"""
    w = set()
    for line in ins:
        op, rest = line.split(" ", 1)
        o = [x.strip() for x in rest.split(",")]
        dst = [o[0]] + (["vcc"] if len(o) > 1 and o[1] == "vcc" and op.endswith(("_co_u32", "u64_u32")) else [])
        for d in dst:
            if d.startswith("v["):
                lo = int(d[2:d.index(":")]); hi = int(d[d.index(":") + 1:-1])
                w.update(f"v{r}" for r in range(lo, hi + 1))
            else:
                w.add(d)
    return w


def check_clobbers(N, pre, ins, cl, n_out, label):
    """the asm statement's contract with the register allocator: every physical register the stream writes is a declared
    clobber, every %n it writes is an output operand, no input operand (%n >= n_out) is written, and an output is not written
    before the last read of an input (outputs are not early-clobber, so the allocator may give an output an input's register)"""
    w = written_registers(pre + ins)
    phys = {r for r in w if not r.startswith("%")}
    undeclared = phys - set(cl)
    assert not undeclared, (label, "written but not in the clobber list", sorted(undeclared))
    outs = {r for r in w if r.startswith("%")}
    assert all(int(r[1:]) < n_out for r in outs), (label, "input operand written", sorted(outs))
    first_out = min(i for i, line in enumerate(ins) if line.split(" ", 1)[1].split(",")[0].strip().startswith("%"))
    import re as _re
    last_in = max(i for i, line in enumerate(ins) if any(int(x) >= n_out for x in _re.findall(r"%(\d+)", line.split(" ", 1)[1].split(",", 1)[1] if "," in line else "")))
    assert first_out > last_in, (label, "an output is written while inputs are still live", first_out, last_in)
    # reads of physical registers must have been written earlier in the stream (no dependence on the caller's values)
    seen = set()
    for line in pre + ins:
        op, rest = line.split(" ", 1)
        o = [x.strip() for x in rest.split(",")]
        srcs = o[1:] if not (len(o) > 1 and o[1] == "vcc" and op.endswith(("_co_u32", "u64_u32"))) else o[2:]
        for x in srcs:
            regs = []
            if x.startswith("v["):
                lo = int(x[2:x.index(":")]); hi = int(x[x.index(":") + 1:-1]); regs = [f"v{r}" for r in range(lo, hi + 1)]
            elif _re.fullmatch(r"[vs]\d+", x) or x == "vcc":
                regs = [x]
            for r in regs:
                assert r in seen, (label, "reads a register the stream has not written", r, line)
        seen |= written_registers([line])
    return len(phys)


def emulate(N, ins, limbs, ninv, a, b):
    """runs the instruction stream on one lane; returns the output limbs"""
    n_out = N if N == 8 else N // 2
    reg = {}
    for i in range(N):
        reg[f"s{S0 + i}"] = limbs[i]
        reg[f"%{n_out + i}"] = a[i]
        if b is not None:
            reg[f"%{n_out + N + i}"] = b[i]
    reg[f"s{S0 + N}"] = ninv
    vcc = 0
    MASK = 0xffffffff

    def rd(x):
        return 0 if x == "0" else reg[x]

    def prd(x):   # v[a:b]
        lo = int(x[2:x.index(":")]); return reg[f"v{lo}"] | (reg[f"v{lo + 1}"] << 32)

    def pwr(x, v):
        lo = int(x[2:x.index(":")]); reg[f"v{lo}"] = v & MASK; reg[f"v{lo + 1}"] = (v >> 32) & MASK

    for line in ins:
        op, rest = line.split(" ", 1)
        o = [x.strip() for x in rest.split(",")]
        if op == "v_mov_b32":
            reg[o[0]] = rd(o[1])
        elif op == "v_mov_b64":
            reg[o[0]] = prd(o[1])
        elif op == "v_mad_u64_u32":
            assert o[1] == "vcc" and int(o[0][2:o[0].index(":")]) % 2 == 0
            v = rd(o[2]) * rd(o[3]) + (0 if o[4] == "0" else prd(o[4])); vcc = v >> 64; pwr(o[0], v & ((1 << 64) - 1))
        elif op == "v_addc_co_u32":
            assert o[1] == "vcc" and o[4] == "vcc"
            v = rd(o[2]) + rd(o[3]) + vcc; reg[o[0]] = v & MASK; vcc = v >> 32
        elif op == "v_add_co_u32":
            v = rd(o[2]) + rd(o[3]); reg[o[0]] = v & MASK; vcc = v >> 32
        elif op == "v_sub_co_u32":
            v = rd(o[2]) - rd(o[3]); reg[o[0]] = v & MASK; vcc = 1 if v < 0 else 0
        elif op == "v_subbrev_co_u32":                      # dst = src1 - src0 - borrow
            assert o[1] == "vcc" and o[4] == "vcc"
            v = rd(o[3]) - rd(o[2]) - vcc; reg[o[0]] = v & MASK; vcc = 1 if v < 0 else 0
        elif op == "v_lshlrev_b32":
            reg[o[0]] = (rd(o[2]) << int(o[1])) & MASK
        elif op == "v_alignbit_b32":
            reg[o[0]] = (((rd(o[1]) << 32) | rd(o[2])) >> int(o[3])) & MASK
        elif op == "v_add_u32":
            reg[o[0]] = (rd(o[1]) + rd(o[2])) & MASK
        elif op == "v_sub_u32":
            reg[o[0]] = (rd(o[1]) - rd(o[2])) & MASK
        elif op == "v_cmp_ne_u32":
            assert o[0] == "vcc"
            vcc = 1 if rd(o[1]) != rd(o[2]) else 0
        elif op == "v_mul_lo_u32":
            reg[o[0]] = (rd(o[1]) * rd(o[2])) & MASK
        else:
            raise ValueError(line)
    if N == 8:
        return [reg[f"%{i}"] for i in range(N)]
    return [x for i in range(N // 2) for x in (reg[f"%{i}"] & MASK, reg[f"%{i}"] >> 32)]


def check(N, fs, sqr=False, rounds=200):
    rng = random.Random(1)
    split = lambda v: [(v >> (32 * i)) & 0xffffffff for i in range(N)]
    for name, limbs, ninv in fs:
        ins = body(N, sqr, limbs, ninv)
        pre = prologue(N, limbs, ninv)
        nreg = check_clobbers(N, pre, ins, clobbers(N, sqr), N if N == 8 else N // 2, name)
        p = sum(l << (32 * i) for i, l in enumerate(limbs))
        assert p < 1 << (32 * N - 1)
        rinv = pow(1 << (32 * N), -1, p)
        for r in range(rounds):
            a, b = (rng.randrange(p) if r > 3 else p - 1 - r), (rng.randrange(p) if r > 1 else p - 1)
            if r == 4:
                a = sum(0x80000000 << (32 * i) for i in range(N - 1))           # every carry-in bit of D set
            if r == 5:
                a = 0                                                            # every m_k = 0: no carry out of a finished word
            if r == 6:
                a, b = 1, 1
            if r == 7:
                a, b = (1 << (32 * N)) % p, rng.randrange(p)                     # Montgomery one: low words cancel exactly
            if sqr:
                b = a
            t = emulate(N, ins, limbs, ninv, split(a), None if sqr else split(b))
            tv = sum(l << (32 * i) for i, l in enumerate(t))
            assert tv < 2 * p and tv % p == a * b * rinv % p, (name, hex(a), hex(b))
        print(f"  emulator: {name} ({N} limbs, {'square' if sqr else 'product'}) ok on {rounds} operands; {len(ins)} instructions, "
              f"{sum(1 for y in ins if y.startswith('v_mad'))} v_mad_u64_u32; {nreg} physical registers written, all declared")


def asm_fn(o, N, fname, pre, ins, cl, two):
    text = "\\n\\t".join(pre + ins)
    args = f"const uint32_t (&a)[{N}], const uint32_t (&b)[{N}]" if two else f"const uint32_t (&a)[{N}]"
    o.append(f"  static __device__ __forceinline__ void {fname}(uint32_t (&t)[{N}], {args}) {{")
    if N == 8:
        o.append(f'    asm("{text}"')
        o.append("        : " + ", ".join(f'"=v"(t[{i}])' for i in range(N)))
    else:
        o.append(f"    uint64_t o[{N // 2}];")
        o.append(f'    asm("{text}"')
        o.append("        : " + ", ".join(f'"=v"(o[{i}])' for i in range(N // 2)))
    o.append("        : " + ", ".join(f'"v"(a[{i}])' for i in range(N)) + (", " + ", ".join(f'"v"(b[{i}])' for i in range(N)) if two else ""))
    o.append("        : " + ", ".join(f'"{c}"' for c in cl) + ");")
    if N != 8:
        o.append("#pragma unroll")
        o.append(f"    for (int i = 0; i < {N // 2}; i++) {{ t[2 * i] = (uint32_t)o[i]; t[2 * i + 1] = (uint32_t)(o[i] >> 32); }}")
    o.append("  }")


def prologue(N, limbs, ninv):
    return [f"s_mov_b32 s{S0 + i}, 0x{limbs[i]:08x}" for i in range(N)] + [f"s_mov_b32 s{S0 + N}, 0x{ninv:08x}"]


def emit(o, N, fs):
    o.append(f"template <class F> struct MontAsm{N} {{ static constexpr bool value = false; }};")
    o.append("")
    cnt = lambda x: f"{len(x)} VALU instructions ({sum(1 for y in x if y.startswith('v_mad'))} v_mad_u64_u32)"
    for name, limbs, ninv in fs:
        ins, ins_sqr = body(N, False, limbs, ninv), body(N, True, limbs, ninv)
        pre = prologue(N, limbs, ninv)
        o.append(f"// {name}: product {cnt(ins)}, square {cnt(ins_sqr)}, + {N + 1} s_mov_b32 each")
        o.append(f"template <> struct MontAsm{N}<{name}> {{")
        o.append("  static constexpr bool value = true;")
        o.append(f"  // t = a * b / 2^{32 * N} mod p, t < 2p (the caller subtracts p once)")
        asm_fn(o, N, "mul", pre, ins, clobbers(N), True)
        o.append(f"  // t = a * a / 2^{32 * N} mod p, t < 2p")
        asm_fn(o, N, "sqr", pre, ins_sqr, clobbers(N, True), False)
        o.append("};")
        o.append("")


def main():
    o = ["// mont8_asm_gen.h -- GENERATED by tools/gen_mont_asm.py from consts_gen.h; do not edit.",
         "// 8- and 12-limb Montgomery multiplications as one inline-asm block per field (see the generator's docstring).",
         "#pragma once", "#include <hip/hip_runtime.h>", "#include <stdint.h>", '#include "consts_gen.h"', "",
         "namespace avrf {", ""]
    for N in (8, 12):
        fs = fields(N)
        if "--check" in sys.argv:
            check(N, fs)
            check(N, fs, True)
        emit(o, N, fs)
        print(f"{N} limbs: fields", [f[0] for f in fs])
    o.append("}  // namespace avrf")
    open(OUT, "w").write("\n".join(o) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
