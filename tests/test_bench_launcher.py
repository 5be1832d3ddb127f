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
    assert out["contexts_per_rank"] >= 1


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
    """contexts per rank and hashing mode from the CPU quota divided by the ranks (bench.host_plan)"""
    sys.path.insert(0, ROOT)
    import bench
    streams, mb, cores, threads = bench.host_plan(16, 1, 0, 480)
    assert (streams, mb, cores, threads) == (21, 0, 16.0, 7)      # seven host threads keep three contexts each in flight
    streams, mb, cores, threads = bench.host_plan(16, 8, 0, 480)  # 2 cores per rank: the multi-buffer service, 8 lanes per thread
    assert mb == 2 and streams == 20 and cores == 2.0 and threads == 20     # ... with one host thread per context
    assert bench.host_plan(256, 8, 0, 480) == (21, 0, 32.0, 7)
    assert bench.host_plan(16, 2, 0, 480)[::3] == (21, 7) and bench.host_plan(12, 2, 0, 480)[::3] == (18, 6)
    assert bench.host_plan(16, 4, 0, 480) == (12, 0, 4.0, 4)      # 4 cores per rank: still scalar chains, four pipelined threads
    assert bench.host_plan(16, 1, 0, 20)[::3] == (21, 7)                                     # (a block of steps may be smaller than the contexts)
    assert bench.host_plan(16, 1, 12, 480)[::3] == (12, 7)                                   # --streams overrides
    assert bench.host_plan(16, 1, 20, 480, 20)[::3] == (20, 20)                              # --host-threads: one thread per context
    assert all(bench.host_plan(q, w, 0, 480)[0] <= 23 for q in (1, 2, 4, 16, 64, 256) for w in (1, 2, 4, 8))   # 24 contexts halve the rate
    assert bench.cpu_quota() >= 1
