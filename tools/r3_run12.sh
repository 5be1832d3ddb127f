# A/B: multiplier with the 2^32 - 1 modulus limbs done by subtract / add (libavrf.so) against a multiply-add for every limb (libavrf_prev.so)
OUT=gpurun_out/r3ones; mkdir -p $OUT
for rep in 1 2; do
  for L in libavrf_prev.so libavrf.so; do
    export AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-ring --no-cpu-baseline > $OUT/multi_${L}_$rep.json 2>/dev/null
    python bench.py --gpus 1 --streams 1 --steps 20 --warmup 5 --no-ring --no-cpu-baseline > $OUT/single_${L}_$rep.json 2>/dev/null
    python - $OUT/multi_${L}_$rep.json $OUT/single_${L}_$rep.json $L <<'P'
import json, sys
m = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[3], "multi %.2f M/s (%.3f ms/step), k_accumulate alone %.4f ms; single context %.2f M/s, device_msm %.0f us" % (
    m["value"] / 1e6, m["ms_per_step"], m["roofline"]["kernel_avg_ms"], s["value"] / 1e6, s["step_breakdown_us"]["one_context_alone"]["device_msm"]))
P
    python tools/ped_bench.py 2>&1 | grep "/s" | sed "s/^/$L  /"
    python tools/ring_bench.py 1024 4096 4 2>&1 | grep "contexts x" | sed "s/^/$L  /"
  done
done
