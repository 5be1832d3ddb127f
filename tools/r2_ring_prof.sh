OUT=${1:-gpurun_out/r2j}
mkdir -p $OUT
python tools/ring_bench.py 1024 2048 1 > $OUT/ring_1ctx.txt 2>&1
python tools/ring_bench.py 1024 4096 4 > $OUT/ring_4ctx.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_ring -o ring -- python3 $GRAFT_REPO_ROOT/tools/ring_bench.py 1024 2048 1 > $GRAFT_REPO_ROOT/$OUT/ring_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $OUT/prof_ring > $OUT/kstats_ring.txt
find $OUT/prof_ring -name "*.csv" ! -name "*kernel_stats*" -delete
cat $OUT/ring_1ctx.txt $OUT/ring_4ctx.txt; head -22 $OUT/kstats_ring.txt
