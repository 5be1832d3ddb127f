# A/B: resident k_accumulate<TeCurve> waves per SIMD (141 VGPRs allow 3) capped with AVRF_MSM_OCC
OUT=gpurun_out/r3occ; mkdir -p $OUT
for rep in 1 2; do
  for OCC in 0 2 1; do
    export AVRF_MSM_OCC=$OCC; [ $OCC = 0 ] && unset AVRF_MSM_OCC
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-ring --no-cpu-baseline > $OUT/multi_${OCC}_$rep.json 2>/dev/null
    python bench.py --gpus 1 --streams 1 --steps 20 --warmup 5 --no-ring --no-cpu-baseline > $OUT/single_${OCC}_$rep.json 2>/dev/null
    python - $OUT/multi_${OCC}_$rep.json $OUT/single_${OCC}_$rep.json occ=$OCC <<'P'
import json, sys
m = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[3], "multi %.2f M/s (%.3f ms/step, %d contexts), k_accumulate alone %.4f ms; single context %.2f M/s, device_msm %.0f us" % (
    m["value"] / 1e6, m["ms_per_step"], m["config"]["contexts_per_gpu"], m["roofline"]["kernel_avg_ms"], s["value"] / 1e6, s["step_breakdown_us"]["one_context_alone"]["device_msm"]))
P
  done
done
