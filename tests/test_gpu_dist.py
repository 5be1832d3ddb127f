"""GPU: one thin BatchVerifier split into shards through the C ABI pieces used by
ark_vrf_amd/dist.py -- P logical shards on the one visible device (the box has a single GPU; RCCL
refuses two ranks on one device), combined on the host -- and the same through dist.sharded_thin_batch_verify
with a 1-rank process group."""
import os
import socket

import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shards", [1, 3, 4])
def test_logical_shards_sum_to_identity(shards):
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.dist import GpuEngine, IDENTITY_XY, shard_range, shard_thin_batch
    n = 501
    good = orc.gen_batch(0, 0, n)
    for tamper in (False, True):
        b = dict(good)
        if tamper:
            p = bytearray(b["proofs"]); p[96 * 77 + 64] ^= 1; b["proofs"] = bytes(p)
        engs = [GpuEngine(nat.Context(0)) for _ in range(shards)]
        cs, ss = [], []
        for r, e in enumerate(engs):
            lo, hi = shard_range(n, r, shards)
            sh = shard_thin_batch(b, lo, hi)
            st, c = e.challenges(sh)
            assert st == 0
            cs.append(c); ss.append(b"".join(sh["proofs"][96 * j + 64: 96 * j + 96] for j in range(sh["n"])))
        seed = engs[0].weight_seed(0, b"".join(cs), b"".join(ss))
        parts = [e.partial(seed, shard_range(n, r, shards)[0]) for r, e in enumerate(engs)]
        total = engs[0].points_sum(0, b"".join(parts))
        assert (total == IDENTITY_XY) == (not tamper)
        # the unsharded verifier agrees
        c0 = engs[0].ctx
        assert c0.thin_batch_stage(nat.Batch(n, b["ios_xy"], b["io_counts"], b["ads"], b["ad_lens"], pks_xy=b["pks_xy"], proofs=b["proofs"])) == 0
        assert c0.thin_batch_run() == (1 if tamper else 0)


def test_sharded_verify_one_rank_group():
    import torch.distributed as dist
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.dist import GpuEngine, sharded_thin_batch_verify
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        b = orc.gen_batch(0, 0, 200)
        eng = GpuEngine(nat.Context(0))
        assert sharded_thin_batch_verify(eng, 0, b, dist) == 0
        p = bytearray(b["proofs"]); p[96 * 5 + 64] ^= 1
        b2 = dict(b); b2["proofs"] = bytes(p)
        assert sharded_thin_batch_verify(eng, 0, b2, dist) == 1
        b3 = dict(b); b3["pks_xy"] = bytes(32) + (1).to_bytes(32, "little") + b["pks_xy"][64:]
        assert sharded_thin_batch_verify(eng, 0, b3, dist) == 2
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("suite,shards", [(0, 1), (0, 3), (1, 4)])
def test_pedersen_logical_shards(suite, shards):
    """pedersen::BatchVerifier (src/pedersen.rs:341-426) split into shards: the partial points add up to the identity for
    a valid batch, not for a tampered one, and the unsharded verifier agrees."""
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.dist import GpuEngine, IDENTITY_XY, shard_pedersen_batch, shard_range
    n = 333
    good = orc.gen_batch(suite, 1, n)
    for tamper in (False, True):
        b = dict(good)
        if tamper:
            p = bytearray(b["proofs"]); p[256 * 200 + 230] ^= 1; b["proofs"] = bytes(p)
        engs = [GpuEngine(nat.Context(suite)) for _ in range(shards)]
        cs, rs = [], []
        for r, e in enumerate(engs):
            lo, hi = shard_range(n, r, shards)
            sh = shard_pedersen_batch(b, lo, hi)
            st, c = e.ped_challenges(sh)
            assert st == 0
            cs.append(c); rs.append(b"".join(sh["proofs"][256 * j + 192: 256 * j + 256] for j in range(sh["n"])))
        seed = engs[0].ped_weight_seed(suite, b"".join(cs), b"".join(rs))
        parts = [e.ped_partial(seed, shard_range(n, r, shards)[0]) for r, e in enumerate(engs)]
        assert (engs[0].points_sum(suite, b"".join(parts)) == IDENTITY_XY) == (not tamper)
        c0 = engs[0].ctx
        assert c0.pedersen_batch_stage(nat.Batch(n, b["ios_xy"], b["io_counts"], b["ads"], b["ad_lens"], proofs=b["proofs"])) == 0
        assert c0.pedersen_batch_run() == (1 if tamper else 0)


def test_sharded_pedersen_verify_one_rank_group():
    import torch.distributed as dist
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.dist import GpuEngine, sharded_pedersen_batch_verify
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        b = orc.gen_batch(0, 1, 150)
        eng = GpuEngine(nat.Context(0))
        assert sharded_pedersen_batch_verify(eng, 0, b, dist) == 0
        p = bytearray(b["proofs"]); p[256 * 9 + 200] ^= 1
        b2 = dict(b); b2["proofs"] = bytes(p)
        assert sharded_pedersen_batch_verify(eng, 0, b2, dist) == 1
        b3 = dict(b); b3["proofs"] = bytes(32) + (1).to_bytes(32, "little") + b["proofs"][64:]     # Yb = identity
        assert sharded_pedersen_batch_verify(eng, 0, b3, dist) == 2
    finally:
        dist.destroy_process_group()


def test_sharded_ring_prove_and_verify_one_rank_group(golden_dir):
    """Ring proofs / verifications split by index over the process group (1 rank here; world 2 on gloo in
    tests/test_dist_gloo.py): same bytes as one call, verdict = worst slice."""
    import json
    import torch.distributed as dist
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.dist import sharded_ring_batch_verify, sharded_ring_prove
    from ark_vrf_amd.ring import RingSetup, ring_batch_verify
    from helpers import xy
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ctx = nat.Context(0)
        setup = RingSetup(ctx, open(os.path.join(golden_dir, "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read(), 8)
        sks = [orc.from_seed(0, bytes([11, i]) + bytes(30)) for i in range(5)]
        key = setup.index([xy(0, pk) for _, pk in sks])
        h = orc.hash_to_curve(0, b"dist-ring")
        idx, bl, ybs = [], [], []
        for j in (4, 0, 2):
            ped, b = orc.pedersen_prove(0, sks[j][0], [(h, orc.vrf_output(0, sks[j][0], h))], b"x")
            idx.append(j); bl.append(b); ybs.append(xy(0, ped[:32]))
        proofs = sharded_ring_prove(key.prove, idx, bl, dist)
        assert proofs == key.prove(idx, bl)
        vf = lambda lo, hi: ring_batch_verify(setup, [key.commitment], None, ybs[lo:hi], proofs[lo:hi])
        assert sharded_ring_batch_verify(vf, 3, dist) == 0
        bad = list(proofs); bad[1] = bad[1][:200] + bytes([bad[1][200] ^ 1]) + bad[1][201:]
        vb = lambda lo, hi: ring_batch_verify(setup, [key.commitment], None, ybs[lo:hi], bad[lo:hi])
        assert sharded_ring_batch_verify(vb, 3, dist) != 0
    finally:
        dist.destroy_process_group()


def test_partial_requires_fresh_challenges():
    """*_batch_partial must not run on challenges of an earlier staging."""
    import ctypes as C
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd.dist import GpuEngine, shard_thin_batch
    b = orc.gen_batch(0, 0, 64)
    eng = GpuEngine(nat.Context(0))
    sh = shard_thin_batch(b, 0, 64)
    st, c = eng.challenges(sh)
    assert st == 0
    seed = eng.weight_seed(0, c, b"".join(b["proofs"][96 * j + 64: 96 * j + 96] for j in range(64)))
    eng.partial(seed, 0)                                                       # fine
    out = (C.c_uint8 * 64)()
    nb = nat.Batch(64, b["ios_xy"], b["io_counts"], b["ads"], b["ad_lens"], pks_xy=b["pks_xy"], proofs=b["proofs"])
    assert eng.ctx.thin_batch_stage(nb) == 0                                   # re-staged: challenges are stale
    assert nat.lib().avrf_thin_batch_partial(eng.ctx._h, nat._u8(seed), C.c_uint64(0), out) == nat.ERR_BAD_ARG
    st, c = eng.challenges(sh)
    assert st == 0 and eng.ctx.thin_verify(nb) == [0] * 64                     # thin_verify stages too
    assert nat.lib().avrf_thin_batch_partial(eng.ctx._h, nat._u8(seed), C.c_uint64(0), out) == nat.ERR_BAD_ARG
    # identity pk: challenges fail -> no partial either
    bad = dict(sh); bad["pks_xy"] = bytes(32) + (1).to_bytes(32, "little") + sh["pks_xy"][64:]
    st, _ = eng.challenges(bad)
    assert st == 2
    assert nat.lib().avrf_thin_batch_partial(eng.ctx._h, nat._u8(seed), C.c_uint64(0), out) == nat.ERR_BAD_ARG


_RCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.distributed as dist
import oracle as orc
from ark_vrf_amd import _native as nat
from ark_vrf_amd.dist import GpuEngine, sharded_thin_batch_verify, sharded_pedersen_batch_verify, sharded_ring_prove
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
b = orc.gen_batch(0, 0, 300)
eng = GpuEngine(nat.Context(0))
assert sharded_thin_batch_verify(eng, 0, b, dist, device="cuda") == 0
p = bytearray(b["proofs"]); p[96 * 5 + 64] ^= 1
b2 = dict(b); b2["proofs"] = bytes(p)
assert sharded_thin_batch_verify(eng, 0, b2, dist, device="cuda") == 1
bp = orc.gen_batch(0, 1, 120)
assert sharded_pedersen_batch_verify(eng, 0, bp, dist, device="cuda") == 0
outs = sharded_ring_prove(lambda idx, bl: [bytes([i]) * 8 for i in idx], [3, 1, 2], [bytes(32)] * 3, dist, device="cuda")
assert outs == [bytes([3]) * 8, bytes([1]) * 8, bytes([2]) * 8]
t = torch.ones(4, device="cuda"); dist.all_reduce(t); assert float(t.sum()) == 4.0
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_path_world_size_one():
    """The collectives of ark_vrf_amd/dist.py on backend "nccl" (= RCCL) with device tensors -- the path an 8-GPU run takes --
    executed for real at world size 1.  Own process: RCCL initialisation stays out of the pytest process."""
    import subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_whole_path_on_rccl_world_size_one():
    """bench.py as a RANK with a process group (AVRF_FORCE_DIST=1: backend nccl at world size 1): the headline loop, the
    split-ONE-batch leg (sharded_thin_batch_verify) and BASELINE configs[3] sharded by index (sharded_ring_prove /
    sharded_ring_batch_verify) -- the code an 8-GPU run executes, at sizes that take seconds."""
    import json, subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(AVRF_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--min-seconds", "0.2",
                        "--items", "4096", "--ring-proofs", "64", "--ring-size", "64", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["host"]["host_cpu_quota"] >= 1
    m = out["multi_gpu"]
    assert m["split_one_batch"]["value"] > 0, m
    assert m["ring_sharded"]["ring_vrf_proofs_per_sec"] > 0 and m["ring_sharded"]["ranks"] == 1, m
    assert out["additional_metrics"]["ring_vrf_batch_verifications_per_sec"] > 0
    # what a scaling run is judged by: the ranks RCCL itself counted, every rank's own figures, the weak-scaling ring leg
    assert m["communicator"]["backend"] == "nccl" and m["communicator"]["ranks_counted_by_an_all_reduce_of_ones"] == 1 and m["communicator"]["world_size_torch"] == 1
    assert len(m["per_rank"]) == 1 and m["per_rank"][0]["value"] > 0 and m["per_rank"][0]["host_cpu_us_per_step"] > 0
    assert m["ring_sharded_weak"].get("ring_vrf_proofs_per_sec", 0) > 0 and "weak" in m["ring_sharded_weak"]["scaling"], m["ring_sharded_weak"]


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` for real: the launcher starts two ranks (torch.distributed.run), which here share the one
    visible GPU and meet over gloo (AVRF_BENCH_SHARE_GPU / AVRF_BENCH_BACKEND -- RCCL refuses two ranks on a device).  Every step
    of the N > 1 flow runs with world size 2: per-rank batches, the barriers and reductions of the timed region, ONE batch split
    over two ranks (challenges and partial points all-gathered), configs[3] sharded by index with the proofs all-gathered and the
    verdicts reduced, one JSON line with n_gpus = 2.  The figures of such a run mean nothing; the verdicts must."""
    import json, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AVRF_BENCH_SHARE_GPU="1", AVRF_BENCH_BACKEND="gloo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--min-seconds", "0.2",
                        "--items", "4096", "--ring-proofs", "96", "--ring-size", "64", "--streams", "2", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["timed_steps_all_gpus"] >= 2 * out["timed_steps_per_gpu"] * 0.5
    m = out["multi_gpu"]
    assert m["split_one_batch"]["value"] > 0, m
    assert m["ring_sharded"]["ranks"] == 2 and m["ring_sharded"]["ring_vrf_proofs_per_sec"] > 0, m
    assert m["ring_sharded"]["ring_vrf_batch_verifications_per_sec"] > 0 and m["ring_sharded"]["ring_vrf_independent_verifications_per_sec"] > 0
    assert m["communicator"]["ranks_counted_by_an_all_reduce_of_ones"] == 2 and sorted(r["rank"] for r in m["per_rank"]) == [0, 1]
    assert all(r["value"] > 0 for r in m["per_rank"]) and m["ring_sharded_weak"].get("ranks") == 2 and m["ring_sharded_weak"]["ring_vrf_proofs_per_sec"] > 0, m["ring_sharded_weak"]
