#!/bin/bash
# Regenerates the committed profile summaries of a round from the CURRENT build, on a GPU box:
#   make stamp                                         (build container: writes build/HEAD_STAMP = the commit being measured)
#   gpurun -- 'bash tools/refresh_profiles.sh r6 [what ...]'      (or: gpurun -- 'make profiles R=r6')
#   what: ubench issue_cost gather pmc_thin pmc_peritem pmc_ring pmc_ring_bn254 single default benchline ring ring_bn254 peritem latency validate (default: all, in
#   this order: bench.py reads its roofs and HBM traffic from profiles/<round>_ubench.txt / _pmc_*.json, so those are produced first and
#   copied into profiles/ of the box's snapshot before the bench line is taken)
# Outputs land in gpurun_out/<round>_profiles/, every one stamped with the commit (json: key "head"; txt / log: first line;
# csv: listed with its sha256 in <round>_STAMP.txt); copy them into profiles/ and commit.
R=${1:-r6}; shift; WHAT=${*:-ubench issue_cost gather pmc_thin pmc_peritem pmc_ring pmc_ring_bn254 single default benchline ring ring_bn254 peritem latency validate}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${R}_profiles; mkdir -p $OUT
HEAD=$(cat $ROOT/build/HEAD_STAMP 2>/dev/null | head -1); HEAD=${HEAD:-unknown}
export TMPDIR=/tmp AVRF_BLOCKING=1
SINGLE="python3 $ROOT/bench.py --gpus 1 --slots 1 --lanes 1 --host-threads 1 --hash-group 1 --steps 20 --warmup 5 --min-seconds 0.5 --no-ring --no-cpu-baseline --no-projection"
PASSES=("FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64")
stats() {   # name, command...: rocprofv3 --kernel-trace --stats, keep the kernel_stats csv
  name=$1; shift; d=$(mktemp -d /tmp/kt.XXXX)
  ( cd /tmp && timeout -k 20 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- "$@" > $OUT/${R}_$name.log 2>&1 )
  f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${R}_${name}_kernel_stats.csv
  python3 $ROOT/tools/kstats.py $d | sort -k4 -n -r -t'|' | head -14; rm -rf $d
}
stamp_json() { python3 - "$1" "$HEAD" "$2" <<'PY'
import json, sys
p, head, cmd = sys.argv[1:4]
try:
    d = json.load(open(p))
except Exception:
    d = json.loads(open(p).read().strip().splitlines()[-1])
d = {"head": head, "profile_command": cmd, **d}
json.dump(d, open(p, "w"), indent=1)
PY
}
stamp_txt() { if [ -s "$1" ]; then sed -i "1i # head $HEAD -- $2" "$1"; else echo "# head $HEAD -- $2 (no output)" > "$1"; fi; }
for w in $WHAT; do case $w in
  ubench) bash $ROOT/tools/ubench_report.sh > $OUT/${R}_ubench.txt 2>&1; stamp_txt $OUT/${R}_ubench.txt "bash tools/ubench_report.sh"; cp $OUT/${R}_ubench.txt $ROOT/profiles/; tail -30 $OUT/${R}_ubench.txt;;
  issue_cost) /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -o /tmp/issue_cost $ROOT/tools/issue_cost.hip && /tmp/issue_cost > $OUT/${R}_issue_cost.txt 2>&1; stamp_txt $OUT/${R}_issue_cost.txt "tools/issue_cost.hip";;
  gather) /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/gather_probe $ROOT/tools/gather_probe.hip && /tmp/gather_probe 200 > $OUT/${R}_gather_probe.txt 2>&1; stamp_txt $OUT/${R}_gather_probe.txt "tools/gather_probe.hip 200";;
  pmc_peritem) python3 $ROOT/tools/pmc.py $OUT/${R}_pmc_per_item.json "SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "VALUBusy" -- python3 $ROOT/tools/ped_bench.py 65536; stamp_json $OUT/${R}_pmc_per_item.json "tools/pmc.py <passes> -- python3 tools/ped_bench.py 65536"; cp $OUT/${R}_pmc_per_item.json $ROOT/profiles/;;
  single) stats single_context $SINGLE;;
  default) stats bench_default python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-ring --no-projection --no-cpu-baseline;;
  benchline) python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_default.json 2> $OUT/${R}_bench_default.err; stamp_json $OUT/${R}_bench_default.json "python bench.py --gpus 1 --steps 20 --warmup 5";;
  ring) stats ring_prove_2048 python3 $ROOT/tools/ring_bench.py 1024 2048 1;;
  ring_bn254) stats ring_bn254_prove_1024 python3 $ROOT/tools/ring_bench.py 4096 1024 1 1;;
  peritem) stats per_item python3 $ROOT/tools/ped_bench.py 65536;;
  latency) python3 $ROOT/tools/latency_report.py > $OUT/${R}_latency_report.txt 2>&1; stamp_txt $OUT/${R}_latency_report.txt "python tools/latency_report.py";;
  validate) python3 $ROOT/tools/validate_bench.py > $OUT/${R}_validate_bench.json 2> $OUT/${R}_validate_bench.err; stamp_json $OUT/${R}_validate_bench.json "python tools/validate_bench.py";;
  pmc_thin) python3 $ROOT/tools/pmc.py $OUT/${R}_pmc_thin.json "${PASSES[@]}" -- $SINGLE; stamp_json $OUT/${R}_pmc_thin.json "tools/pmc.py <passes> -- $SINGLE"; cp $OUT/${R}_pmc_thin.json $ROOT/profiles/;;
  pmc_ring) python3 $ROOT/tools/pmc.py $OUT/${R}_pmc_ring.json "${PASSES[@]}" -- python3 $ROOT/tools/ring_bench.py 1024 512 1; stamp_json $OUT/${R}_pmc_ring.json "tools/pmc.py <passes> -- python3 tools/ring_bench.py 1024 512 1"; cp $OUT/${R}_pmc_ring.json $ROOT/profiles/;;
  pmc_ring_bn254) python3 $ROOT/tools/pmc.py $OUT/${R}_pmc_ring_bn254.json "${PASSES[@]}" -- python3 $ROOT/tools/ring_bench.py 4096 512 1 1; stamp_json $OUT/${R}_pmc_ring_bn254.json "tools/pmc.py <passes> -- python3 tools/ring_bench.py 4096 512 1 1"; cp $OUT/${R}_pmc_ring_bn254.json $ROOT/profiles/;;
esac; done
{ echo "head $HEAD"; echo "date $(date -u +%Y-%m-%dT%H:%M:%SZ)"; echo "what $WHAT"; ( cd $OUT && sha256sum ${R}_* | grep -v STAMP ); } > $OUT/${R}_STAMP.txt
cat $OUT/${R}_STAMP.txt | head -5
