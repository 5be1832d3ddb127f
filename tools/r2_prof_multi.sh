# kernel trace of the default (16-context) bench command
OUT=${1:-gpurun_out/r2g}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_multi -o multi -- python3 $GRAFT_REPO_ROOT/bench.py --no-ring --no-cpu-baseline ${BENCH_ARGS} > $GRAFT_REPO_ROOT/$OUT/bench_multi.json 2> $GRAFT_REPO_ROOT/$OUT/bench_multi.err
cd $GRAFT_REPO_ROOT
python tools/kstats.py $OUT/prof_multi > $OUT/kstats_multi.txt
python - <<EOF
import csv,glob,collections
# busy time: union of kernel intervals, and per-kernel sum, over the timed region (last 480 steps ~ tail of trace)
rows=[]
for f in glob.glob("$OUT/prof_multi/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
if rows:
    t0=rows[0][0]; end=rows[-1][1]
    # union
    busy=0; cs,ce=rows[0][0],rows[0][1]
    for s,e,_ in rows[1:]:
        if s>ce: busy+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    busy+=ce-cs
    print("trace span %.1f ms, union of kernel intervals %.1f ms, sum of durations %.1f ms"%((end-t0)/1e6,busy/1e6,sum(e-s for s,e,_ in rows)/1e6))
EOF
find $OUT/prof_multi -name "*.csv" ! -name "*kernel_stats*" -delete
head -20 $OUT/kstats_multi.txt
