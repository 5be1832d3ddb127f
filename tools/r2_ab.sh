# A/B of library builds: tools/r2_ab.sh <out_dir> <lib1> <lib2> ...
OUT=$1; shift
mkdir -p $OUT
for L in "$@"; do
  tag=$(basename $L .so)
  AVRF_LIB_PATH=$PWD/$L python bench.py --no-ring --no-cpu-baseline --streams 1 --steps 30 --warmup 3 > $OUT/${tag}_s1.json 2> $OUT/${tag}.err
  AVRF_LIB_PATH=$PWD/$L python bench.py --no-ring --no-cpu-baseline --steps 480 > $OUT/${tag}_s16.json 2>> $OUT/${tag}.err
  python - <<EOF
import json
for k in ("s1","s16"):
    d=json.load(open("$OUT/${tag}_%s.json"%k)); print("$tag",k,"value %.1f M/s"%(d["value"]/1e6),"ms/step %.3f"%d["ms_per_step"],"acc_ms %.3f"%d["roofline"]["kernel_avg_ms"], d["config"]["msm_plan"], {a:round(b) for a,b in d["step_breakdown_us"]["one_context_alone"].items()})
EOF
done
