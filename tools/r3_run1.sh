# round 3, first GPU validation: full -m gpu suite, default bench, ubench (Bandersnatch modulus special case), the
# host-starved regime emulated with taskset (2 cores: scalar hashing vs the multi-buffer service), single-context kernel stats
set -x
OUT=gpurun_out/r3a
mkdir -p $OUT
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; lscpu | head -20 >> $OUT/host.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench tools/ubench.hip && timeout 300 /tmp/ubench > $OUT/ubench.txt 2>&1
grep -n "te_madd\|fp_mul<FqBand" $OUT/ubench.txt
# host-starved regime (what 8 ranks on a 16-CPU quota would see): 2 cores for the whole process
for mode in scalar mb; do
  if [ $mode = mb ]; then export AVRF_HASH_THREADS=2; S=20; else unset AVRF_HASH_THREADS; S=4; fi
  timeout 300 taskset -c 0-1 python bench.py --gpus 1 --steps 40 --warmup 10 --streams $S --no-ring --no-cpu-baseline > $OUT/bench_2cores_$mode.json 2> $OUT/bench_2cores_$mode.err; echo "rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_2cores_$mode.json").read().strip().splitlines()[-1]); print("$mode", d["value"]/1e6, "M/s", d["config"]["contexts_per_gpu"], d["host"])
except Exception as e: print("$mode", e)
PY
done
unset AVRF_HASH_THREADS
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_single -o single -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 20 --warmup 3 --no-ring --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_single.json 2> $GRAFT_REPO_ROOT/$OUT/bench_single.err
cd $GRAFT_REPO_ROOT
python tools/kstats.py $OUT/prof_single > $OUT/kstats_single.txt
find $OUT/prof_single -name "*.csv" ! -name "*kernel_stats*" -delete; find $OUT/prof_single -name "*.db" -delete
cat $OUT/kstats_single.txt | head -30
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("HEADLINE", d["value"]/1e6, "M/s", d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["host"])
a=d.get("additional_metrics",{}); print({k:v for k,v in a.items() if not isinstance(v,dict)})
PY
