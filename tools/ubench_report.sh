#!/bin/bash
# The instruction-rate log bench.py reads its roofs from: builds tools/ubench.hip with the library's flags, runs it on the GPU
# and appends the multiply-add counts of the mixed-addition loops taken from the SAME binary's disassembly.
#   bash tools/ubench_report.sh > profiles/rN_ubench.txt        (on a GPU box: gpurun)
cd "$(dirname "$0")" || exit 1
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -enable-ipra=0 -I../ark_vrf_amd/csrc --save-temps=obj -o $T/ubench ubench.hip 2> $T/build.log || { cat $T/build.log; exit 1; }
$T/ubench
# the unsaturated-limb forms the bucket-accumulation kernels run on (round 5), same flags, same box
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -enable-ipra=0 -I../ark_vrf_amd/csrc --save-temps=obj -o $T/ubench_fpu ubench_fpu.hip 2> $T/build_fpu.log || { cat $T/build_fpu.log; exit 1; }
echo "=== tools/ubench_fpu.hip"
$T/ubench_fpu
cat $T/ubench_fpu-hip-amdgcn-amd-amdhsa-gfx950.s >> $T/ubench-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - $T/ubench-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
print("--- disassembly of these binaries: v_mad_u64_u32 + v_mad_i64_i32 per loop iteration / per mixed addition (k_madd / k_g1madd <.., 1> = the unsaturated forms; k_fumul / k_fmul_sat loops hold TWO multiplications)")
for f in re.split(r'\n(?=_Z[\w]+:)', s):
    name = f.split(':', 1)[0]
    if not any(k in name for k in ("k_opILi0", "k_mad_asm", "k_madd", "k_g1madd", "k_fumul", "k_fmul_sat")):
        continue
    lines = [l.strip() for l in f.split('\n')]
    isn = lambda l: re.match(r'^(v_|s_|ds_|global_|buffer_|flat_)', l) is not None
    mads = [i for i, l in enumerate(lines) if l.startswith("v_mad_u64_u32") or l.startswith("v_mad_i64_i32")]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r'^s_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a, b = labels[m.group(1)], i
            cnt = sum(1 for x in mads if a <= x < b)
            if best is None or cnt > best[0]:
                best = (cnt, sum(1 for x in lines[a:b] if isn(x)))
    extra = ""
    if "k_g1madd" in name:     # the loop body also holds the doubling branch (P == Q); the addition proper ends at the first long gap
        br = [i for i, l in enumerate(lines) if l.startswith("s_cbranch_execz")]
        cum = sorted(set(sum(1 for x in mads if x < i) for i in br))
        extra = "  cumulative multiply-adds at the exec branches: %s (first non-zero plateau = the mixed addition, the rest = the doubling branch)" % cum
    print("  %-58s loop: %s multiply-adds in %s instructions%s" % (name[:58], best[0] if best else "?", best[1] if best else "?", extra))
PY
rm -rf $T
