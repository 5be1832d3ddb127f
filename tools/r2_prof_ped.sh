OUT=${1:-gpurun_out/r2p}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_ped -o ped -- python3 $GRAFT_REPO_ROOT/tools/ped_bench.py > $GRAFT_REPO_ROOT/$OUT/ped_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $OUT/prof_ped > $OUT/kstats_ped.txt
find $OUT/prof_ped -name "*.csv" ! -name "*kernel_stats*" -delete
cat $OUT/ped_prof.txt; head -14 $OUT/kstats_ped.txt
