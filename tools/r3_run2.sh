# round 3: (a) size of the shared host pool for the ring legs, (b) host-starved regime: scalar chains vs multi-buffer service
set -x
OUT=gpurun_out/r3b
mkdir -p $OUT
for T in 16 32 64; do
  AVRF_HOST_THREADS=$T timeout 600 python bench.py --ring-only > $OUT/ring_pool$T.json 2> $OUT/ring_pool$T.err
  python - <<PY
import json
d=json.loads(open("$OUT/ring_pool$T.json").read().strip().splitlines()[-1])
print("pool $T:", {k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,dict) and k!="workload"}, d.get("ring_verify_single_ms_by_entry_point"))
print("   c4:", {k: round(v,1) for k,v in d["configs4_shape"].items() if isinstance(v,float)})
print("   c2:", {k: round(v,1) for k,v in d["configs2_shape"].items() if isinstance(v,float)})
print("   roofline:", d.get("roofline"))
PY
done
for cfg in "0 4" "0 8" "2 20" "2 28"; do
  set -- $cfg
  timeout 300 taskset -c 0-1 python bench.py --gpus 1 --steps 40 --warmup 10 --hash-threads $1 --streams $2 --no-ring --no-cpu-baseline > $OUT/bench_2cores_h$1_s$2.json 2> $OUT/bench_2cores_h$1_s$2.err; echo "rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_2cores_h$1_s$2.json").read().strip().splitlines()[-1]); print("2 cores, hash threads $1, contexts $2:", round(d["value"]/1e6,1), "M/s", d["host"]["weight_hash"])
except Exception as e: print("$cfg", e)
PY
done
