#!/usr/bin/env python3
"""Generates ark_vrf_amd/csrc/fpu_asm_gen.h: fu_mul / fu_sqr of fpu.h (Montgomery product of two elements in 9 x 29-bit signed
limbs, product scanning with the reduction interleaved) as ONE inline-asm block per field.

Why: the compiler's form of the same C++ spends ~100 instructions per product beside the 162 multiply-adds (64-bit adds that join
column chains it split for latency, moves that build 64-bit addends, two shifts per column: profiles/r5_ubench.txt, 532
instructions for two products); the stream below is the algorithm as written -- 162 multiply-adds, two instructions for each
m_k, one 64-bit shift per column, one mask per result limb: 207 instructions (squaring: 126 multiply-adds, 180).  A field with
p = 1 mod 2^29 (Bandersnatch's base field) takes the subtractive form: m_k is the column's low 29 bits as they are (one v_and_b32), the
reduction subtracts m_k p through signed products with the negated modulus limbs, the column's own m_k p_0 = m_k is what the arithmetic
shift drops: 153 multiply-adds in 179 vector instructions (squaring 117 in 151).  The bucket
accumulation is bound by VALU issue, so instructions are what counts.

  column k < 9:   acc += sum_i a_i b_(k-i) + sum_(i<k) m_i p_(k-i);  m_k = (-p^-1 lo(acc)) mod 2^29;  acc += m_k p_0;  acc >>= 29
  column k >= 9:  acc += sum a_i b_(k-i) + sum m_i p_(k-i);  r_(k-9) = lo(acc) mod 2^29;  acc >>= 29          r_8 = lo(acc)

Registers: the accumulator is the fixed pair v[ACC:ACC+1] (an asm operand cannot name the halves of a 64-bit operand), the modulus
limbs sit in s[S0 : S0 + L) (a VOP3 instruction on gfx9 takes no 32-bit literal; the s_mov_b32 that load them issue on the scalar
unit), -p^-1 next to them; all declared as clobbers.  An asm statement takes at most 30 operands: see class Form for how the 9-limb
(8-word fields) and the 14-limb (BLS12-381 base field, 14 x 28 bits: 392 multiply-adds, 461 instructions against ~508) statements
stay below it.  a_i b_j are signed (v_mad_i64_i32); m_i p_j are non-negative.

  python tools/gen_fpu_asm.py            # rewrites the header from consts_gen.h
  python tools/gen_fpu_asm.py --check    # writes nothing: runs the streams in an emulator against tools/fpu_model.py's fu_mul, limb for
                                         # limb, and fails if the committed header is not what the generator produces
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fpu_model  # noqa: E402

OUT = os.path.join(ROOT, "ark_vrf_amd", "csrc", "fpu_asm_gen.h")
ACC, S0 = 30, 36                 # accumulator pair v[30:31]; modulus limbs from s36

M0 = 32                          # 14-limb form: m_k in v[32..45]
ACCP = f"v[{ACC}:{ACC + 1}]"
LO = f"v{ACC}"
# fields left to the compiler's form (none: BN254's base field, used by the G1 bucket accumulation alone, gains 2.6 % on the ring
# 4096 prover -- gpurun_out/r5/ab_bn254asm.txt)
SKIP = set()


class Form:
    """operand numbering of one statement.  9 limbs ("sep"): %0..%8 = r (early-clobber outputs; m_k lives in r_k's register: m_k is
    last read in column k + 8, r_k is written in column k + 9), then a, then b (squaring: r, the doubled limbs as further outputs,
    a).  14 limbs ("tied"): 14 + 14 + 14 operands would pass the 30-operand limit, so a is read-write and returns r (a_j is last
    read in column j + 13, r_j is written in column j + 14), b follows (squaring: the doubled limbs as scratch outputs), m_k
    sits in fixed clobbered VGPRs."""
    def __init__(self, f, sqr):
        self.L, self.W, self.sqr, self.tied = f.L, f.W, sqr, f.L > 9
        L = self.L
        if self.tied:
            self.R = self.A = lambda i: f"%{i}"
            self.B = lambda i: f"%{L + i}"
            self.M = lambda i: f"v{M0 + i}"
            self.D = lambda i: f"%{L + i}"                  # squaring: the doubled limbs are scratch outputs the compiler places
            self.n_written = 2 * L if sqr else L
        else:
            self.R = self.M = lambda i: f"%{i}"
            if sqr:
                self.D = lambda i: f"%{L + i}"
                self.A = lambda i: f"%{2 * L + i}"
                self.n_written = 2 * L
            else:
                self.A = lambda i: f"%{L + i}"
                self.B = lambda i: f"%{2 * L + i}"
                self.n_written = L

    def fixed_vgprs(self):
        v = [ACC, ACC + 1]
        if self.tied:
            v += [M0 + i for i in range(self.L)]
        return [f"v{x}" for x in v]


def body(f, sqr=False):
    fm = Form(f, sqr)
    L, W, MASK = f.L, f.W, f.MASK
    ins = []
    P = lambda j: f"s{S0 + j}"
    p0_one = f.pl[0] == 1 and f.ninv == MASK
    for j in range(L):
        if not (p0_one and j == 0):                          # p = 1 mod 2^W: the limbs are held NEGATED (the reduction subtracts lo_k p)
            ins.append(f"s_mov_b32 {P(j)}, 0x{((-f.pl[j]) & 0xffffffff) if p0_one else f.pl[j]:x}")
    if not p0_one:
        ins.append(f"s_mov_b32 s{S0 + L}, 0x{f.ninv:x}")
    if sqr:
        for j in range(1, L):
            ins.append(f"v_lshlrev_b32 {fm.D(j)}, 1, {fm.A(j)}")
    fresh = True

    def mad(op, x, y):
        nonlocal fresh
        ins.append(f"{op} {ACCP}, vcc, {x}, {y}, {'0' if fresh else ACCP}")
        fresh = False

    for k in range(2 * L - 1):
        lo, hi = (0, k) if k < L else (k - L + 1, L - 1)
        for i in range(lo, hi + 1):
            j = k - i
            if not sqr:
                mad("v_mad_i64_i32", fm.A(i), fm.B(j))
            elif i == j:
                mad("v_mad_i64_i32", fm.A(i), fm.A(i))
            elif i < j:
                mad("v_mad_i64_i32", fm.A(i), fm.D(j))
        for i in range(lo, min(hi, k - 1) + 1):
            mad("v_mad_i64_i32" if p0_one else "v_mad_u64_u32", fm.M(i), P(k - i))
        if k < L:
            if p0_one:                       # p = 1 mod 2^W: subtract lo_k p with lo_k = the column's low W bits as they are: lo_k p_0 = lo_k is
                ins.append(f"v_and_b32 {fm.M(k)}, 0x{MASK:x}, {LO}")        # exactly what the arithmetic shift below drops (floor), and the later
            else:                                                            # columns take lo_k (-p_j) as signed products: one instruction per m_k
                ins.append(f"v_mul_lo_u32 {fm.M(k)}, {LO}, s{S0 + L}")
                ins.append(f"v_and_b32 {fm.M(k)}, 0x{MASK:x}, {fm.M(k)}")
                ins.append(f"v_mad_u64_u32 {ACCP}, vcc, {fm.M(k)}, {P(0)}, {ACCP}")
        else:
            ins.append(f"v_and_b32 {fm.R(k - L)}, 0x{MASK:x}, {LO}")
        ins.append(f"v_ashrrev_i64 {ACCP}, {W}, {ACCP}")
    ins.append(f"v_mov_b32 {fm.R(L - 1)}, {LO}")
    return ins


def clobbers(f, sqr=False):
    L = f.L
    p0_one = f.pl[0] == 1 and f.ninv == f.MASK
    s = [f"s{S0 + j}" for j in range(L + 1) if not (p0_one and j in (0, L))]
    return Form(f, sqr).fixed_vgprs() + s + ["vcc"]


def emulate(ins, f, a, b, sqr=False):
    """one lane; registers hold 32-bit patterns; returns the signed result limbs"""
    fm = Form(f, sqr)
    L = f.L
    reg = {}
    M32, M64 = (1 << 32) - 1, (1 << 64) - 1
    sgn32 = lambda v: v - (1 << 32) if v >> 31 else v
    sgn64 = lambda v: v - (1 << 64) if v >> 63 else v
    for i in range(L):
        reg[fm.A(i)] = a[i] & M32
        if not sqr:
            reg[fm.B(i)] = b[i] & M32

    def rd(x):
        if x.startswith("0x"):
            return int(x, 16)
        if x.isdigit():
            return int(x)
        return reg[x]

    def prd(x):
        lo = int(x[2:x.index(":")]); return reg[f"v{lo}"] | (reg[f"v{lo + 1}"] << 32)

    def pwr(x, v):
        lo = int(x[2:x.index(":")]); reg[f"v{lo}"] = v & M32; reg[f"v{lo + 1}"] = (v >> 32) & M32

    for line in ins:
        op, rest = line.split(" ", 1)
        o = [x.strip() for x in rest.replace(ACCP, "PAIR").split(",")]
        o = [ACCP if x == "PAIR" else x for x in o]
        if op == "s_mov_b32" or op == "v_mov_b32":
            reg[o[0]] = rd(o[1])
        elif op == "v_lshlrev_b32":
            reg[o[0]] = (rd(o[2]) << rd(o[1])) & M32
        elif op == "v_mad_i64_i32":
            add = 0 if o[4] == "0" else sgn64(prd(o[4]))
            v = sgn32(rd(o[2])) * sgn32(rd(o[3])) + add
            assert -(1 << 63) <= v < (1 << 63), "signed 64-bit overflow"
            pwr(o[0], v & M64)
        elif op == "v_mad_u64_u32":
            add = 0 if o[4] == "0" else prd(o[4])
            pwr(o[0], (rd(o[2]) * rd(o[3]) + add) & M64)
        elif op == "v_lshl_add_u64":
            sp = o[3]; lo_s = int(sp[2:sp.index(":")])
            pwr(o[0], ((prd(o[1]) << rd(o[2])) + (reg[f"s{lo_s}"] | (reg[f"s{lo_s + 1}"] << 32))) & M64)
        elif op == "v_sub_u32":
            reg[o[0]] = (rd(o[1]) - rd(o[2])) & M32
        elif op == "v_mul_lo_u32":
            reg[o[0]] = (rd(o[1]) * rd(o[2])) & M32
        elif op == "v_and_b32":
            reg[o[0]] = rd(o[1]) & rd(o[2])
        elif op == "v_ashrrev_i64":
            pwr(o[0], (sgn64(prd(o[2])) >> rd(o[1])) & M64)
        else:
            raise SystemExit("emulator: " + line)
    return [sgn32(reg[fm.R(i)]) for i in range(L)]


def check_contract(ins, f, sqr, label):
    """every physical register written is a declared clobber; only output (or read-write) operands are written; every read of a
    physical register follows a write inside the stream; in the tied form a result limb is written only after the last read of
    the operand limb it replaces"""
    import re
    fm, cl = Form(f, sqr), clobbers(f, sqr)
    seen = set()
    last_read, first_write = {}, {}
    for n, line in enumerate(ins):
        op, rest = line.split(" ", 1)
        o = [x.strip() for x in rest.replace(ACCP, "PAIR").split(",")]
        has_vcc = op in ("v_mad_i64_i32", "v_mad_u64_u32")
        dst, srcs = o[0], (o[2:] if has_vcc else o[1:])
        for x in srcs:
            regs = [f"v{ACC}", f"v{ACC + 1}"] if x == "PAIR" else ([x] if re.fullmatch(r"[vs]\d+", x) else [])
            if re.fullmatch(r"s\[\d+:\d+\]", x):
                regs = [f"s{int(x[2:x.index(':')])}", f"s{int(x[2:x.index(':')]) + 1}"]
            for r in regs:
                assert r in seen, (label, "reads a register the stream has not written", r, line)
            if x.startswith("%"):
                last_read[x] = n
        w = [f"v{ACC}", f"v{ACC + 1}"] if dst == "PAIR" else [dst]
        if has_vcc:
            w.append("vcc")
        for r in w:
            if r.startswith("%"):
                assert int(r[1:]) < fm.n_written, (label, "input operand written", line)
                first_write.setdefault(r, n)
            else:
                assert r in cl, (label, "written but not a clobber", r)
            seen.add(r)
    if fm.tied:
        for i in range(f.L):
            assert first_write[fm.R(i)] > last_read[fm.A(i)], (label, "result limb written while its operand limb is live", i)


def emit(name, f, ins_mul, ins_sqr):
    L = f.L
    txt = lambda ins: "\n".join(f'      "{x}\\n\\t"' for x in ins)
    nv = lambda ins: len([x for x in ins if x.startswith("v_")])
    cm = ", ".join(f'"{c}"' for c in clobbers(f))
    cs = ", ".join(f'"{c}"' for c in clobbers(f, True))
    head = f"\ntemplate <> struct FuAsm<{name}> {{ static constexpr bool value = true; }};\n"
    if L > 9:
        rw = ", ".join(f'"+&v"(a.v[{i}])' for i in range(L))
        douts14 = ", ".join(f'"=&v"(d{i})' for i in range(L))
        dd14 = ", ".join(f"d{i}" for i in range(L))
        inb = ", ".join(f'"v"(b.v[{i}])' for i in range(L))
        return head + f"""AVRF_DI fu<{L}> fu_mul_asm({name}, fu<{L}> a, const fu<{L}> &b) {{   // {nv(ins_mul)} vector instructions; a's registers return the product
  asm(
{txt(ins_mul)}
      : {rw}
      : {inb}
      : {cm});
  return a;
}}
AVRF_DI fu<{L}> fu_sqr_asm({name}, fu<{L}> a) {{   // {nv(ins_sqr)} vector instructions
  int32_t {dd14};
  asm(
{txt(ins_sqr)}
      : {rw}, {douts14}
      :
      : {cs});
  (void)d0;
  return a;
}}
"""
    outs = ", ".join(f'"=&v"(r.v[{i}])' for i in range(L))
    ina = ", ".join(f'"v"(a.v[{i}])' for i in range(L))
    inb = ", ".join(f'"v"(b.v[{i}])' for i in range(L))
    douts = ", ".join(f'"=&v"(d{i})' for i in range(L))
    dd = ", ".join(f"d{i}" for i in range(L))
    return head + f"""AVRF_DI fu<{L}> fu_mul_asm({name}, const fu<{L}> &a, const fu<{L}> &b) {{   // {nv(ins_mul)} vector instructions
  fu<{L}> r;
  asm(
{txt(ins_mul)}
      : {outs}
      : {ina}, {inb}
      : {cm});
  return r;
}}
AVRF_DI fu<{L}> fu_sqr_asm({name}, const fu<{L}> &a) {{   // {nv(ins_sqr)} vector instructions
  fu<{L}> r;
  int32_t {dd};
  asm(
{txt(ins_sqr)}
      : {outs}, {douts}
      : {ina}
      : {cs});
  (void)d0;
  return r;
}}
"""


def main():
    C = fpu_model.parse()
    rng = random.Random(20261004)
    out = ["// fpu_asm_gen.h -- GENERATED by tools/gen_fpu_asm.py from consts_gen.h; do not edit.",
           "// fu_mul / fu_sqr (fpu.h) as single asm blocks: the algorithm of the C++ forms instruction for instruction, without the moves /",
           "// 64-bit adds / second shifts the compiler adds (see the generator's header).  Included by fpu.h.",
           "#pragma once", "namespace avrf {", "template <class F> struct FuAsm { static constexpr bool value = false; };"]
    n = 0
    for name, d in C.items():
        if not name.startswith("Fq") or name in SKIP or "P" not in d or d.get("P_n") not in (8, 12) or "NINV" not in d or (d["P"] >> (32 * d["P_n"] - 1)):
            continue
        f = fpu_model.Field(name, d)
        L = f.L
        im, isq = body(f), body(f, sqr=True)
        check_contract(im, f, False, name + " mul")
        check_contract(isq, f, True, name + " sqr")
        if "--check" in sys.argv:
            big, small = 1 << (f.W + 1), (1 << f.W) + 16
            for t in range(300):
                if t < 4:
                    a = [(-big, big)[(t >> 0) & 1]] * L; b = [(-small, small)[(t >> 1) & 1]] * L
                else:
                    a = [rng.randint(-big, big) for _ in range(L)]; b = [rng.randint(-small, small) for _ in range(L)]
                assert emulate(im, f, a, b) == f.mul(a, b), (name, "mul", a, b)
                assert emulate(isq, f, b, None, sqr=True) == f.mul(b, b), (name, "sqr", b)
            nv = lambda ins: len([x for x in ins if x.startswith("v_")])
            print(f"  {name}: {L} x {f.W} bits, mul {nv(im)} / sqr {nv(isq)} vector instructions; 300 random + extreme operand sets == fu_mul of "
                  f"tools/fpu_model.py, limb for limb")
        out.append(emit(name, f, im, isq)); n += 1
    out.append("}  // namespace avrf\n")
    text = "\n".join(out)
    cur = open(OUT).read() if os.path.exists(OUT) else None
    if "--check" in sys.argv:                              # nothing is written: the committed header must be what the generator produces
        if cur != text:
            raise SystemExit(f"{OUT} is stale: run python tools/gen_fpu_asm.py")
        print(f"{OUT} is current ({n} fields)")
    elif cur != text:
        open(OUT, "w").write(text)
        print(f"wrote {OUT}: {n} fields")
    else:
        print(f"{OUT} unchanged ({n} fields)")


if __name__ == "__main__":
    main()
