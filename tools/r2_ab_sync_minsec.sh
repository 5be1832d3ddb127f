for rep in 1 2 3; do
for cfg in "--host-wait spin --streams 16 --min-seconds 0.3" "--host-wait blocking --streams 24 --min-seconds 0.3" "--host-wait blocking --streams 24 --min-seconds 1.0" "--host-wait blocking --streams 24 --min-seconds 2.0"; do
  python bench.py --no-ring --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), d['timed_blocks'])"
done; done
