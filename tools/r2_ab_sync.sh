# A/B on one box: host threads spinning (runtime default) vs sleeping in the driver while they wait, over context counts
# usage: r2_ab_sync.sh "<modes>" "<context counts>" [reps]
MODES=${1:-"spin blocking"}; CTXS=${2:-"16 24 32"}; REPS=${3:-2}
for rep in $(seq $REPS); do
for mode in $MODES; do
  for s in $CTXS; do
    python bench.py --host-wait $mode --no-ring --no-cpu-baseline --steps 20 --warmup 5 --streams $s --min-seconds 1.0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['step_breakdown_us']['all_contexts_in_flight']
print('$mode', $s, 'q=${GPU_MAX_HW_QUEUES:-24}', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms; hash', round(a['host_weight_transcript']), 'us')"
  done
done
done
