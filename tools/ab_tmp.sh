D="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-ring --no-projection --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["host"]; print(round(d["value"]/1e6,1), "M/s  threads", h["host_threads_per_rank"], "slots", h["slots_per_rank"], "lanes", h["lanes_per_rank"], "hash us/step", round(h["host_cpu_us_per_step"]["hash"]), h.get("cpu_model"), "sleeps", h["host_sleeps_per_step"])'
for i in 1 2; do for t in 6 10 14; do $D --host-threads $t 2>/dev/null | python3 -c "$P"; done; done
