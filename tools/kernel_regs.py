#!/usr/bin/env python3
"""Register / scratch use of the gfx950 kernels in an object or shared library (clang offload bundles): the numbers behind the
occupancy choices of curves.h.     python tools/kernel_regs.py build/obj/msm.o [name-substring]"""
import re, struct, subprocess, sys, tempfile, os

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


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
    blob = open(sys.argv[1], "rb").read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co); path = f.name
        try:
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
        finally:
            os.unlink(path)
        for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name or pat not in name.group(1):
                continue
            g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [0, "?"])[1]
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
            print(f"{dem[:110]:110s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} scratch {g('private_segment_fixed_size'):>5} spills v{g('vgpr_spill_count')} s{g('sgpr_spill_count')}")


if __name__ == "__main__":
    main()
