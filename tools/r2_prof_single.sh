# single-context kernel trace of the headline workload (one BatchVerifier context, kernels uncontended)
OUT=${1:-gpurun_out/r2a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_single -o single -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 20 --warmup 3 --no-ring --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_single.json 2> $GRAFT_REPO_ROOT/$OUT/bench_single.err
cd $GRAFT_REPO_ROOT
python tools/kstats.py $OUT/prof_single > $OUT/kstats_single.txt
find $OUT/prof_single -name "*.csv" ! -name "*kernel_stats*" -delete
cat $OUT/kstats_single.txt
