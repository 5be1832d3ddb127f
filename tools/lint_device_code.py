#!/usr/bin/env python3
"""Build-time fence for the compiler fragility of DESIGN.md section 7-5.

Field multiplications come in two forms: the GENERATED blocks (mont8_asm_gen.h: one asm statement per multiplication, carries
through vcc, clobber lists proven complete by tools/gen_mont_asm.py --check) and the GENERIC form (mac96.h: one asm statement
per limb product, each with an SGPR-pair carry output, scheduled by the compiler; the fields whose modulus has its top bit set
-- secp256r1 -- and a few wide products use it).  The generated blocks are inlined by the dozen into kernels AND into
out-of-line device functions (te_smul*: up to 14 700 multiply-adds) and have never misbehaved.  The generic form inlined FORTY
times into one out-of-line function did: tools/secp_probe.hip, `te_smul<SuiteSecp256r1>` with the short-Weierstrass law
inlined, ~5 000 SGPR-carry multiply-adds in one non-kernel function -> memory access fault on gfx950 with this compiler, while
the same body inlined into its kernel, or calling ONE out-of-line multiplier, is correct.  The rule since: an out-of-line
function holds at most a few generic multiplications (the short-Weierstrass law calls fp_mul_nf).  This script enforces it on
the code that ships: it unbundles the gfx950 code objects of libavrf.so (clang offload bundles in .hip_fatbin), disassembles
them and counts, per NON-KERNEL function, the v_mad_u64_u32 whose carry goes to an SGPR pair.  The largest such function that
is known good has 720 (f12_sqr of pairing.hip, exercised by every pairing test); the build fails above LIMIT = 1024, a fifth
of the count that faulted.  tools/sgpr_carry_repro.hip is the stand-alone reduction of that shape (40 / 80 inlined generic multiplications
in one non-kernel function, no product headers): it does NOT fault, with or without interprocedural register allocation -- the count alone is
not the cause; the limit fences the one shape known bad.

    python tools/lint_device_code.py [path/to/libavrf.so]      exit status 1 on a violation
"""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ark_vrf_amd", "libavrf.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
LIMIT = 1024              # SGPR-carry multiply-adds in one out-of-line function (kernels are exempt: they carry a descriptor <name>.kd)
ALLOW = ()


def code_objects(blob):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", blob, i + 24)
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24: off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" in triple and sz:
                yield blob[i + o: i + o + sz]
        pos = i + 24


def main():
    blob = open(LIB, "rb").read()
    bad, n_funcs, n_objs = [], 0, 0
    for co in code_objects(blob):
        n_objs += 1
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co); path = f.name
        try:
            syms = subprocess.run([OBJDUMP, "-t", path], capture_output=True, text=True).stdout
            kernels = {m.group(1) for m in re.finditer(r"\s(\S+)\.kd\s*$", syms, re.M)}
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
        finally:
            os.unlink(path)
        cur, cnt = None, 0
        def close():
            nonlocal cur, cnt
            if cur is not None and cur not in kernels:
                nonlocal_count(cur, cnt)
        def nonlocal_count(name, c):
            nonlocal n_funcs
            n_funcs += 1
            if c > LIMIT and not any(a in name for a in ALLOW):
                bad.append((name, c))
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                close(); cur, cnt = m.group(1), 0
            elif re.search(r"v_mad_u64_u32 v\[\d+:\d+\], s\[", line):
                cnt += 1
        close()
    print(f"lint_device_code: {n_objs} gfx950 code objects, {n_funcs} out-of-line device functions checked, limit {LIMIT} SGPR-carry multiply-adds each")
    if not n_objs:
        print("lint_device_code: no gfx950 code object found in", LIB); return 1
    for name, c in bad:
        print(f"  VIOLATION: {name}: {c} SGPR-carry v_mad_u64_u32 in a non-kernel function (generic multiplications inlined en masse: call fp_mul_nf instead)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
