#!/bin/bash
# Regenerates the committed profile summaries of a round from the CURRENT build, on a GPU box:
#   gpurun -- 'bash tools/refresh_profiles.sh r4 [what ...]'      what: ubench single default ring peritem pmc_thin pmc_ring (default: all)
# Outputs land in gpurun_out/<round>_profiles/; copy them into profiles/ and commit.
R=${1:-r4}; shift; WHAT=${*:-ubench single default ring peritem pmc_thin pmc_ring}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${R}_profiles; mkdir -p $OUT
export TMPDIR=/tmp AVRF_BLOCKING=1
SINGLE="python3 $ROOT/bench.py --gpus 1 --slots 1 --lanes 1 --host-threads 1 --hash-group 1 --steps 20 --warmup 5 --min-seconds 0.5 --no-ring --no-cpu-baseline --no-projection"
PASSES=("FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum")
stats() {   # name, command...: rocprofv3 --kernel-trace --stats, keep the kernel_stats csv
  name=$1; shift; d=$(mktemp -d /tmp/kt.XXXX)
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- "$@" > $OUT/${R}_$name.log 2>&1 )
  f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${R}_${name}_kernel_stats.csv
  python3 $ROOT/tools/kstats.py $d | sort -k4 -n -r -t'|' | head -14; rm -rf $d
}
for w in $WHAT; do case $w in
  ubench) bash $ROOT/tools/ubench_report.sh > $OUT/${R}_ubench.txt 2>&1; tail -30 $OUT/${R}_ubench.txt;;
  single) stats single_context $SINGLE;;
  default) stats bench_default python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-ring --no-projection --no-cpu-baseline;;
  ring) stats ring_prove_2048 python3 $ROOT/tools/ring_bench.py 1024 2048 1;;
  peritem) stats per_item python3 $ROOT/tools/ped_bench.py 65536;;
  pmc_thin) python3 $ROOT/tools/pmc.py $OUT/${R}_pmc_thin.json "${PASSES[@]}" -- $SINGLE;;
  pmc_ring) python3 $ROOT/tools/pmc.py $OUT/${R}_pmc_ring.json "${PASSES[@]}" -- python3 $ROOT/tools/ring_bench.py 1024 512 1;;
esac; done
