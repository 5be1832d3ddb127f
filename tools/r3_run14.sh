# host threads x contexts per thread with the three-call run (bench.py --host-threads): the headline against the thread-per-context form
for Q in 24 48; do
export GPU_MAX_HW_QUEUES=$Q
for cfg in "20 20" "21 7" "22 11" "23 8" "24 8" "28 7" "32 8" "20 10" "20 5" "21 3"; do
  set -- $cfg
  python bench.py --gpus 1 --streams $1 --host-threads $2 --steps 48 --warmup 8 --no-ring --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $Q contexts', d['config']['contexts_per_gpu'], 'threads', d['config']['host_threads_per_gpu'], round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), 'ms/step')
"
done
done
