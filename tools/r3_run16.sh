# host-starved ranks (what N = 4 / 8 ranks on a 16-CPU quota see), emulated with taskset: the multi-buffer hash service with a
# blocking thread per context (host_plan's choice below 6 cores) against scalar chains on pipelined host threads
run() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']/1e6,2), 'M/s; contexts', d['config']['contexts_per_gpu'], 'threads', d['config']['host_threads_per_gpu'], '|', d['host']['weight_hash'][:60])
"; }
for cpus in 0-3 0-1; do
  taskset -c $cpus python bench.py --gpus 1 --steps 20 --warmup 5 --min-seconds 1 --no-ring --no-cpu-baseline 2>/dev/null | run "cpus $cpus default:"
  for cfg in "9 3" "12 4" "6 2" "15 5"; do set -- $cfg
    taskset -c $cpus python bench.py --gpus 1 --hash-threads 0 --streams $1 --host-threads $2 --steps 20 --warmup 5 --min-seconds 1 --no-ring --no-cpu-baseline 2>/dev/null | run "cpus $cpus scalar $1/$2:"
  done
done
