#!/usr/bin/env python3
"""Instruction counts of the shipped kernels, taken from the disassembly of the objects libavrf.so is linked from:
   python tools/kernel_counts.py build/obj/msm.o [more objects] > ark_vrf_amd/kernel_counts.json      (the csrc Makefile runs this)
For every kernel whose name matches KERNELS: the whole kernel's and its LARGEST LOOP's instruction mix by class (multiply-adds,
other 64-bit integer, plain 32-bit VALU, LDS, vector memory, scalar).  bench.py prints `multiply-adds per mixed addition` from
this file, i.e. from the very binary whose launches it times."""
import json, os, re, subprocess, sys, tempfile
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_regs import code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNELS = ("k_accumulate",)
MAD = ("v_mad_u64_u32", "v_mad_i64_i32")
I64 = ("v_lshl_add_u64", "v_ashrrev_i64", "v_lshrrev_b64", "v_lshlrev_b64", "v_cmp_lt_u64", "v_cmp_le_u64")


def klass(op):
    if op in MAD: return "multiply_add"
    if op.split("_e")[0] in I64 or op in I64: return "int64_other"
    if op.startswith("v_mul_lo") or op.startswith("v_mul_hi"): return "mul32"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("v_"): return "valu32"
    return "scalar"


def functions(obj):
    blob = open(obj, "rb").read()
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co); path = f.name
        try:
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
        finally:
            os.unlink(path)
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            m = re.match(r"[0-9a-f]+ <(\S+)>:", fn)
            if not m or m.group(1).endswith(".kd"):
                continue
            ins = []
            for line in fn.split("\n")[1:]:
                mm = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if mm:
                    ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
            if ins:
                yield m.group(1), ins


def largest_loop(ins):
    """(first, last) instruction indices of the longest backward branch's span; SOPP branch targets are pc + 4 + 4 * simm16"""
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            t = re.match(r"(-?\d+)", args)
            if not t:
                continue
            off = int(t.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt in addr and addr[tgt] < i and (best is None or i - addr[tgt] > best[1] - best[0]):
                best = (addr[tgt], i)
    return best


def summarize(ins):
    c = Counter(klass(op) for _, op, _ in ins)
    return {"instructions": len(ins), "vector_alu": c["multiply_add"] + c["int64_other"] + c["mul32"] + c["valu32"], **{k: c[k] for k in ("multiply_add", "int64_other", "mul32", "valu32", "lds", "vmem", "scalar")}}


def main():
    out = {"objects": [os.path.basename(o) for o in sys.argv[1:]], "kernels": {}}
    for obj in sys.argv[1:]:
        for name, ins in functions(obj):
            if not any(k in name for k in KERNELS):
                continue
            e = {"whole": summarize(ins)}
            lp = largest_loop(ins)
            if lp:
                body = ins[lp[0]:lp[1] + 1]
                e["largest_loop"] = summarize(body)
                e["largest_loop"]["valu_mix"] = dict(Counter(op.replace("_e32", "").replace("_e64", "") for _, op, _ in body if op.startswith("v_")).most_common(40))
            out["kernels"][name] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
