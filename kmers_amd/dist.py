"""Multi-GPU layer of the k-mer scan: one process per GPU, reads sharded embarrassingly.

The reference has no distributed code at all: k-mers never span reads
(CanonicalKmerIterator is per slice, src/naive_impl/canonical_kmer_iterator.rs:72-83), so rank g
simply owns a contiguous range of reads and NO data-path collective is needed for the reduce /
materialise passes.  The only exchange steps are
  * combining the per-rank 32-byte summaries (wrapping add / xor), and
  * the optional bucket histogram: one all-reduce(sum, int64) of 2^b counters (RCCL over xGMI on
    GPUs; `backend="nccl"` IS RCCL on ROCm).
Everything here works on CPU tensors with the gloo backend too (that is how it is tested without GPUs).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

M64 = (1 << 64) - 1


def shard_range(n_reads: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced partition of reads: returns (first_read, n_reads_of_rank)."""
    base, extra = divmod(n_reads, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def rccl_rank_count(device=None, group=None) -> int:
    """Ranks that really take part in a collective: all-reduce(sum) of a one.  bench.py asserts it equals the
    number of GPUs it claims (a process group that silently has one rank would otherwise measure one GPU)."""
    if not dist.is_initialized():
        return 1
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    return int(one.item())


def _split(v: int) -> tuple[int, int]:
    return v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF


def combine_summaries(local: dict[str, int], group=None, device=None) -> dict[str, int]:
    """All-reduce a per-rank summary {n_valid, sum_canon, xor_hash, sum_fw} (python ints, u64).

    Sums are wrapping mod 2^64.  They are exchanged as 32-bit halves held in int64 lanes so the
    all-reduce itself can never overflow a signed lane, then recombined with carry on every rank.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(local)
    adds = ("n_valid", "sum_canon", "sum_fw")
    parts = []
    for k in adds:
        parts.extend(_split(local.get(k, 0) & M64))
    t_add = torch.tensor(parts, dtype=torch.int64, device=device)
    t_xor = torch.tensor([local.get("xor_hash", 0) & 0x7FFFFFFFFFFFFFFF, (local.get("xor_hash", 0) >> 63) & 1],
                         dtype=torch.int64, device=device)
    dist.all_reduce(t_add, op=dist.ReduceOp.SUM, group=group)
    # RCCL/NCCL has no BXOR reduction: gather the 16 bytes of every rank and fold locally
    gathered = [torch.empty_like(t_xor) for _ in range(dist.get_world_size(group))]
    dist.all_gather(gathered, t_xor, group=group)
    for i, g in enumerate(gathered):
        t_xor = g.clone() if i == 0 else torch.bitwise_xor(t_xor, g)
    out = {}
    vals = [int(x) for x in t_add.cpu().tolist()]
    for i, k in enumerate(adds):
        lo, hi = vals[2 * i], vals[2 * i + 1]
        out[k] = (lo + (hi << 32)) & M64
    x = [int(v) for v in t_xor.cpu().tolist()]
    out["xor_hash"] = (x[0] | ((x[1] & 1) << 63)) & M64
    return out


def allreduce_histogram(counts: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of per-rank bucket counters (int64, 2^b entries): ONE collective, the only real
    exchange step of the path (BASELINE configs[4]).  On GPUs this is ncclAllReduce(sum, int64) via
    RCCL over xGMI; 8 MiB at b=20, latency-bound."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


class ShardedScanner:
    """Per-rank driver: owns a kmers_amd.api.Context on its GPU, scans its shard, combines summaries."""

    def __init__(self, ctx, rank: int | None = None, world: int | None = None, group=None):
        self.ctx = ctx
        self.group = group
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)

    def canonical_reduce(self, local_bases, n_local_reads, read_len, k, hasher=0, hasher_k=0, flags=0) -> dict[str, int]:
        s = self.ctx.canonical_reduce(local_bases, n_local_reads, read_len, k, hasher, hasher_k, flags)
        local = {"n_valid": s.n_valid, "sum_canon": s.sum_canon, "xor_hash": s.xor_hash, "sum_fw": s.sum_fw}
        return combine_summaries(local, self.group, device=self.ctx.device)

    def histogram(self, local_bases, n_local_reads, read_len, k, hasher, hasher_k, log2_buckets) -> torch.Tensor:
        counts = self.ctx.histogram(local_bases, n_local_reads, read_len, k, hasher, hasher_k, log2_buckets)
        return allreduce_histogram(counts, self.group)
