"""Multi-GPU orchestration: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm;
"gloo" in the CPU tests).

Two modes:

* independent batches / proofs: every rank owns whole units, no data-path collective -- that is what
  bench.py measures (`shard_range` + one all_reduce of verdicts);
* ONE thin::BatchVerifier split over the ranks (`sharded_thin_batch_verify`): the weights of
  src/thin.rs:274-289 are squeezed from a transcript over ALL items' (c_j, s_j), so the per-item
  challenges are all-gathered, every rank derives the same weight seed, computes the MSM of its own
  shard's terms, and the P partial points (64 bytes each) are all-gathered and added on the host.
  RCCL has no elliptic-curve reduction operator, hence all-gather + local sum rather than all-reduce.

The engine object hides who computes: `GpuEngine` drives libavrf.so; the CPU tests pass an
oracle-backed stand-in with the same three methods.
"""
import ctypes as C

IDENTITY_XY = bytes(32) + (1).to_bytes(32, "little")


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) of n units for `rank` of `world`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_thin_batch(b, lo, hi):
    """Slice a packed thin batch dict (C-ABI layout, see oracle.gen_batch / bench.make_batch)."""
    io_pre, ad_pre = [0], [0]
    for c in b["io_counts"]:
        io_pre.append(io_pre[-1] + c)
    for a in b["ad_lens"]:
        ad_pre.append(ad_pre[-1] + a)
    return dict(n=hi - lo, pks_xy=b["pks_xy"][64 * lo: 64 * hi], ios_xy=b["ios_xy"][128 * io_pre[lo]: 128 * io_pre[hi]],
                io_counts=b["io_counts"][lo:hi], ads=b["ads"][ad_pre[lo]: ad_pre[hi]], ad_lens=b["ad_lens"][lo:hi],
                proofs=b["proofs"][96 * lo: 96 * hi])


def shard_pedersen_batch(b, lo, hi):
    """Same for a packed Pedersen batch (proofs 256 bytes per item, no public keys)."""
    io_pre, ad_pre = [0], [0]
    for c in b["io_counts"]:
        io_pre.append(io_pre[-1] + c)
    for a in b["ad_lens"]:
        ad_pre.append(ad_pre[-1] + a)
    return dict(n=hi - lo, ios_xy=b["ios_xy"][128 * io_pre[lo]: 128 * io_pre[hi]], io_counts=b["io_counts"][lo:hi],
                ads=b["ads"][ad_pre[lo]: ad_pre[hi]], ad_lens=b["ad_lens"][lo:hi], proofs=b["proofs"][256 * lo: 256 * hi])


class GpuEngine:
    """libavrf.so-backed engine for one rank (one context = one GPU stream)."""

    def __init__(self, ctx):
        from . import _native as nat
        self.ctx, self.nat = ctx, nat

    # ---- pedersen::BatchVerifier pieces (src/pedersen.rs:276-293,341-426)
    def ped_challenges(self, shard):
        nat = self.nat
        b = nat.Batch(shard["n"], shard["ios_xy"], shard["io_counts"], shard["ads"], shard["ad_lens"], proofs=shard["proofs"])
        st = self.ctx.pedersen_batch_stage(b)
        if st != 0:
            return st, b""
        out = (C.c_uint8 * max(1, 16 * shard["n"]))()
        st = nat.lib().avrf_pedersen_batch_challenges(self.ctx._h, out)
        return st, bytes(out)[: 16 * shard["n"]]

    def ped_weight_seed(self, suite, c_all, resp_all):
        seed = (C.c_uint8 * 64)()
        st = self.nat.lib().avrf_batch_weight_seed(int(suite), 1, C.c_size_t(len(resp_all) // 64), self.nat._u8(c_all), self.nat._u8(resp_all), seed)
        assert st == 0
        return bytes(seed)

    def ped_partial(self, seed, first_index):
        out = (C.c_uint8 * 64)()
        st = self.nat.lib().avrf_pedersen_batch_partial(self.ctx._h, self.nat._u8(seed), C.c_uint64(first_index), out)
        assert st == 0, st
        return bytes(out)

    def challenges(self, shard):
        nat = self.nat
        b = nat.Batch(shard["n"], shard["ios_xy"], shard["io_counts"], shard["ads"], shard["ad_lens"],
                      pks_xy=shard["pks_xy"], proofs=shard["proofs"])
        st = self.ctx.thin_batch_stage(b)
        if st != 0:
            return st, b""
        out = (C.c_uint8 * max(1, 16 * shard["n"]))()
        st = nat.lib().avrf_thin_batch_challenges(self.ctx._h, out)
        return st, bytes(out)[: 16 * shard["n"]]

    def weight_seed(self, suite, c_all, s_all):
        seed = (C.c_uint8 * 64)()
        st = self.nat.lib().avrf_batch_weight_seed(int(suite), 0, C.c_size_t(len(s_all) // 32), self.nat._u8(c_all), self.nat._u8(s_all), seed)
        assert st == 0
        return bytes(seed)

    def partial(self, seed, first_index):
        out = (C.c_uint8 * 64)()
        st = self.nat.lib().avrf_thin_batch_partial(self.ctx._h, self.nat._u8(seed), C.c_uint64(first_index), out)
        assert st == 0, st
        return bytes(out)

    def points_sum(self, suite, pts):
        out = (C.c_uint8 * 64)()
        st = self.nat.lib().avrf_points_sum(int(suite), C.c_size_t(len(pts) // 64), self.nat._u8(pts), out)
        assert st == 0, st
        return bytes(out)


def _all_gather_bytes(dist, group, data, device):
    """all-gather of variable-length byte strings (lengths first, then padded payloads)."""
    import torch
    world = dist.get_world_size(group)
    ln = torch.tensor([len(data)], dtype=torch.int64, device=device)
    lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lens, ln, group=group)
    lens = [int(x.item()) for x in lens]
    cap = max(max(lens), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if data:
        buf[: len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    outs = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return [bytes(o[:l].cpu().numpy().tobytes()) for o, l in zip(outs, lens)]


def sharded_thin_batch_verify(engine, suite, batch, dist, group=None, device="cpu"):
    """thin::BatchVerifier::verify (src/thin.rs:257-325) of ONE batch split by items over the ranks of
    `group`.  Every rank passes the same full `batch` dict (or at least its own shard's bytes) and gets
    the same status: 0 Ok, 1 VerificationFailure, 2 InvalidData."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = batch["n"]
    if n == 0:
        return 0                                                   # src/thin.rs:262-264
    lo, hi = shard_range(n, rank, world)
    shard = shard_thin_batch(batch, lo, hi)
    st, c_mine = engine.challenges(shard)
    # InvalidData anywhere rejects the batch before any equation (src/thin.rs:266-271)
    flag = torch.tensor([1 if st != 0 else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        return 2
    s_mine = b"".join(shard["proofs"][96 * j + 64: 96 * j + 96] for j in range(shard["n"]))
    parts = _all_gather_bytes(dist, group, c_mine + s_mine, device)   # exchange step 1: challenges + responses
    c_all = b"".join(p[: len(p) // 3] for p in parts)                # 16 bytes c + 32 bytes s per item
    s_all = b"".join(p[len(p) // 3:] for p in parts)
    seed = engine.weight_seed(suite, c_all, s_all)                    # sequential hash, every rank the same
    mine = engine.partial(seed, lo)
    pts = _all_gather_bytes(dist, group, mine, device)               # exchange step 2: P partial points
    total = engine.points_sum(suite, b"".join(pts))
    return 0 if total == IDENTITY_XY else 1


def sharded_pedersen_batch_verify(engine, suite, batch, dist, group=None, device="cpu"):
    """pedersen::BatchVerifier::verify (src/pedersen.rs:341-426) of ONE batch split by items over the ranks: same two
    exchange steps as the Thin form (challenges + responses s || sb, then the partial points)."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = batch["n"]
    if n == 0:
        return 0                                                   # src/pedersen.rs:343-345
    lo, hi = shard_range(n, rank, world)
    shard = shard_pedersen_batch(batch, lo, hi)
    st, c_mine = engine.ped_challenges(shard)
    flag = torch.tensor([1 if st != 0 else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        return 2                                                   # src/pedersen.rs:348-353
    r_mine = b"".join(shard["proofs"][256 * j + 192: 256 * j + 256] for j in range(shard["n"]))
    parts = _all_gather_bytes(dist, group, c_mine + r_mine, device)   # 16 bytes c + 64 bytes (s || sb) per item
    c_all = b"".join(p[: len(p) // 5] for p in parts)
    r_all = b"".join(p[len(p) // 5:] for p in parts)
    seed = engine.ped_weight_seed(suite, c_all, r_all)
    mine = engine.ped_partial(seed, lo)
    pts = _all_gather_bytes(dist, group, mine, device)
    total = engine.points_sum(suite, b"".join(pts))
    return 0 if total == IDENTITY_XY else 1


# ---- Ring VRF: proofs are independent units -- static partition by proof index, per-ring state
# (SRS tables, prover key) replicated on every rank, one gather of the proof bytes / one reduction of the verdicts.

def sharded_ring_prove(prove_fn, key_indices, blindings, dist, group=None, device="cpu"):
    """ring::Prover::prove's ring half for a list of (key index, blinding) split over the ranks of `group`.
    prove_fn(indices, blindings) -> list of proof bytes for this rank's slice (RingKey.prove on a GPU rank).
    Every rank returns the complete list, in input order."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = len(key_indices)
    lo, hi = shard_range(n, rank, world)
    mine = prove_fn(list(key_indices[lo:hi]), list(blindings[lo:hi])) if hi > lo else []
    parts = _all_gather_bytes(dist, group, b"".join(mine), device)   # the only exchange: the proof bytes
    out = []
    for r, p in enumerate(parts):
        rlo, rhi = shard_range(n, r, world)
        if rhi > rlo:
            plen = len(p) // (rhi - rlo)
            out += [p[i * plen: (i + 1) * plen] for i in range(rhi - rlo)]
    return out


def sharded_ring_batch_verify(verify_fn, n_items, dist, group=None, device="cpu"):
    """ring::BatchVerifier over items split by index: verify_fn(lo, hi) -> status of this rank's slice (0 Ok,
    1 VerificationFailure, 2 InvalidData); the batch's status is the worst one (every slice is its own pairing check)."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_items, rank, world)
    st = verify_fn(lo, hi) if hi > lo else 0
    t = torch.tensor([st], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())
