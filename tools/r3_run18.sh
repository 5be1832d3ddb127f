# A/B: SHA-512 transcript block in the lane's private memory (one store per absorbed word) against the register block with
# 16-way select chains (libavrf_sm.so = suite 0 TUs built with -DAVRF_SHA_BLOCK_IN_MEMORY)
AVRF_LIB_PATH=$PWD/ark_vrf_amd/libavrf_sm.so python -m pytest tests -m gpu -x -q -k "thin or pedersen or tiny or wire or fullsize or vectors or pipeline" 2>&1 | tail -2
for rep in 1 2; do for L in libavrf.so libavrf_sm.so; do
  export AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-ring --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), d['step_breakdown_us']['one_context_alone'])
"
  python tools/ped_bench.py 2>&1 | grep "/s" | grep -v "batch" | sed "s/^/$L  /"
done; done
bash tools/r2_prof_single.sh gpurun_out/r3sm > /dev/null 2>&1; grep "prepare\|terms" gpurun_out/r3sm/kstats_single.txt
