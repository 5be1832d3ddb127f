# the driver's command (--steps 20 --warmup 5): seven pipelined host threads (default) against one blocking thread per context
for rep in 1 2 3; do
  for cfg in "0 0" "20 20" "21 7"; do
    set -- $cfg
    python bench.py --gpus 1 --streams $1 --host-threads $2 --steps 20 --warmup 5 --no-ring --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('args $1 $2: contexts', d['config']['contexts_per_gpu'], 'threads', d['config']['host_threads_per_gpu'], round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), 'ms/step', d['step_breakdown_us']['all_contexts_in_flight'])
"
  done
done
