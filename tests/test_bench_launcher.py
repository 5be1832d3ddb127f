"""bench.py's own rank launcher: `python bench.py --gpus N` without WORLD_SIZE must start N ranks itself --
a child torch.distributed.run spawned before the process touches a GPU -- and hand ONE JSON line through.  CPU only: the ranks
run bench.py's --selftest-launch branch (a gloo group instead of the GPU workload)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "AVRF_BENCH_DRY_LAUNCH")}
    env.update(kw)
    return env


def test_launcher_command_line():
    """argv of the child: torch.distributed.run, one node, N processes, rendezvous on 127.0.0.1, then bench.py with the caller's
    own arguments (so every rank parses the same --gpus / --steps / --warmup)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "20", "--warmup", "5"], env=_clean_env(AVRF_BENCH_DRY_LAUNCH="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_launcher_starts_n_ranks_and_prints_one_line():
    """for real at N = 2: two child ranks join a process group, rank 0 prints the one line, the launcher passes it through and
    returns the children's exit code"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launch"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_counted"] == 2 and out["master_addr"] == "127.0.0.1" and out["local_rank"] == "0"
    assert out["slots_per_rank"] >= 1 and out["host_threads_per_rank"] >= 1


def test_under_a_launcher_the_process_is_a_rank():
    """with WORLD_SIZE in the environment (the driver's torch.distributed.run) bench.py must NOT launch again"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = _clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--selftest-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["ranks_counted"] == 1


def test_host_plan_follows_the_quota():
    """slots, lanes, native host threads and the hash group of a rank's pool from the CPU quota divided by the ranks (bench.host_plan)"""
    sys.path.insert(0, ROOT)
    import bench
    p = bench.host_plan(16, 1)
    assert (p["slots"], p["lanes"], p["threads"], p["group"], p["cores"]) == (144, 12, 6, 8, 16.0)   # six threads, three groups of eight transcripts each, twelve lanes (round 6 sweeps)
    p = bench.host_plan(16, 8)                                    # 2 cores per rank: sixteen transcripts hashed together (two interleaved groups), three groups per thread
    assert (p["threads"], p["group"], p["cores"]) == (2, 16, 2.0) and p["slots"] == 96 and p["lanes"] == 12
    assert bench.host_plan(256, 8)["threads"] == 6 and bench.host_plan(256, 8)["group"] == 8
    assert bench.host_plan(16, 2)["threads"] == 6 and bench.host_plan(8, 2)["threads"] == 4 and bench.host_plan(8, 2)["slots"] == 96
    p = bench.host_plan(16, 4)                                    # 4 cores per rank: multi-buffer hashing on four threads
    assert (p["slots"], p["threads"], p["group"], p["lanes"]) == (96, 4, 8, 12)
    p = bench.host_plan(16, 1, group_arg=1)                       # --hash-group 1: scalar chains, four slots and two lanes per thread
    assert (p["slots"], p["lanes"], p["threads"], p["group"]) == (24, 12, 6, 1)
    p = bench.host_plan(16, 1, slots_arg=12)                      # --slots overrides
    assert (p["slots"], p["threads"], p["lanes"]) == (12, 6, 12)
    p = bench.host_plan(16, 1, slots_arg=20, threads_arg=20)
    assert (p["slots"], p["threads"]) == (20, 20)
    p = bench.host_plan(2, 1, group_arg=16)
    assert (p["group"], p["slots"], p["threads"]) == (16, 96, 2)
    for q in (1, 2, 4, 16, 64, 256):
        for w in (1, 2, 4, 8):
            p = bench.host_plan(q, w)
            assert 1 <= p["threads"] <= p["lanes"] <= p["slots"] and p["lanes"] + p["threads"] <= 21     # lanes + ingest streams
    cp = bench.pick_cpus(2)
    assert len(cp) == min(2, len(os.sched_getaffinity(0))) and len(set(cp)) == len(cp)
    assert bench.cpu_quota() >= 1
