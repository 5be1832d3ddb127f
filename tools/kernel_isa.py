#!/usr/bin/env python3
"""Disassembly statistics of one gfx950 kernel inside an object / shared library: instruction mix of the whole kernel and of its
largest loop (the span of the longest backward branch).   python tools/kernel_isa.py build/obj/msm.o <mangled-name-substring> [--dump]"""
import re, subprocess, sys, tempfile, os
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_regs import code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    blob = open(sys.argv[1], "rb").read()
    pat = sys.argv[2]
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co); path = f.name
        try:
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
        finally:
            os.unlink(path)
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            m = re.match(r"[0-9a-f]+ <(\S+)>:", fn)
            if not m or pat not in m.group(1) or m.group(1).endswith(".kd"):
                continue
            ins = []
            for line in fn.split("\n")[1:]:
                mm = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if mm:
                    ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
            addr = {a: i for i, (a, _, _) in enumerate(ins)}
            best = None
            for i, (a, op, args) in enumerate(ins):
                if op.startswith("s_cbranch") or op == "s_branch":
                    t = re.search(r"<\S+\+0x([0-9a-fA-F]+)>", args)
                    if t:
                        # target address = function start + offset
                        tgt = ins[0][0] + int(t.group(1), 16)
                        if tgt in addr and addr[tgt] < i and (best is None or i - addr[tgt] > best[1] - best[0]):
                            best = (addr[tgt], i)
            print(m.group(1)[:100], "instructions", len(ins), "bytes", ins[-1][0] - ins[0][0])
            print("  whole:", Counter(op for _, op, _ in ins).most_common(12))
            if best:
                body = ins[best[0]:best[1] + 1]
                print("  largest loop:", len(body), "instructions,", body[-1][0] - body[0][0], "bytes")
                print("  loop mix:", Counter(op for _, op, _ in body).most_common(25))
                if "--dump" in sys.argv:
                    for a, op, args in body:
                        print(f"    {op} {args}")


if __name__ == "__main__":
    main()
