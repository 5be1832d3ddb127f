#!/bin/bash
# A/B runs of bench.py on one GPU box: bash tools/ab.sh OUTDIR NAME "ARGS" [NAME "ARGS" ...]
#   ARGS are appended to `python bench.py --gpus 1 --no-ring --no-cpu-baseline --no-projection --steps 20 --warmup 5 --min-seconds 1`;
#   a leading `cpus=LIST` in ARGS runs that variant under `taskset -c LIST` (the per-rank CPU share of a multi-GPU node);
#   a leading `lib=PATH` selects another build of the library (AVRF_LIB_PATH); `env=K=V` sets an environment variable for that variant.
# One summary line per variant; the JSON lines are kept under OUTDIR.  (Replaces the one-off tools/r2_*.sh / r3_run*.sh scripts.)
OUT=$1; shift; mkdir -p "$OUT"
B="python bench.py --gpus 1 --no-ring --no-cpu-baseline --no-projection --steps 20 --warmup 5 --min-seconds 1"
while [ $# -ge 2 ]; do
  name=$1; args=$2; shift 2
  pre=""; envs=""
  for w in $args; do
    case $w in
      cpus=*) pre="taskset -c ${w#cpus=}"; args=${args#"$w"};;
      lib=*) envs="$envs AVRF_LIB_PATH=${w#lib=}"; args=${args#"$w"};;
      env=*) envs="$envs ${w#env=}"; args=${args/"$w"/};;
    esac
  done
  env $envs $pre $B $args > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
path, name = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    h = d["host"]; e = d.get("e2e_including_h2d") or {}; r = d["roofline"]
    print(name, round(d["value"] / 1e6, 2), "M/s | e2e", round((e.get("value") or 0) / 1e6, 2), "h2d GB/s", round(e.get("h2d_GB_per_s") or 0, 1),
          "| slots", h["slots_per_rank"], "lanes", h["lanes_per_rank"], "thr", h["host_threads_per_rank"],
          "| host cpu us/step", {k: round(v) for k, v in h["host_cpu_us_per_step"].items()}, "sleeps/step", round(h["host_sleeps_per_step"], 2),
          "| k_accumulate ms alone/contended", round(r["kernel_avg_ms"], 3), round(r["kernel_avg_ms_contended"], 3))
except Exception as ex:
    print(name, "FAILED", ex); print(open(path.replace(".json", ".err")).read()[-1500:])
PY
done
