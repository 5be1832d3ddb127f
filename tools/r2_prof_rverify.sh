OUT=${1:-gpurun_out/r2v}; mkdir -p $OUT
python tools/ring_verify_bench.py > $OUT/rverify.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o rv -- python3 $GRAFT_REPO_ROOT/tools/ring_verify_bench.py > $GRAFT_REPO_ROOT/$OUT/rverify_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $OUT/prof > $OUT/kstats_rv.txt
find $OUT/prof -name "*.csv" ! -name "*kernel_stats*" -delete
cat $OUT/rverify.txt; head -16 $OUT/kstats_rv.txt
