"""world_size-2 gloo tests (CPU) of the N>1 path: shard ranges, summary combine, histogram all-reduce.
The per-shard compute is stood in by the CPU oracle (test infrastructure); what is under test is the
host-side sharding / combination logic of kmers_amd.dist, identical for RCCL on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_reads, L, k, b, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kmers_amd import dist as kd
    from oracle import oracle

    start, cnt = kd.shard_range(n_reads, rank, world)
    # rank r owns stream bytes [start*L, (start+cnt)*L) of the deterministic synthetic stream
    host = oracle.gen_reads(0x6B6D6572735F7631, start * L, cnt * L)
    host = host.copy()
    host[::997] = ord("N")  # some dirty reads too
    s = oracle.canonical_reduce(host, cnt, L, k, hasher_k=k)
    local = {"n_valid": s.n_valid, "sum_canon": s.sum_canon, "xor_hash": s.xor_hash, "sum_fw": s.sum_fw}
    tot = kd.combine_summaries(local)
    counts = torch.from_numpy(oracle.histogram(host, cnt, L, k, k, b).view(np.int64).copy())
    kd.allreduce_histogram(counts)
    if rank == 0:
        out_q.put((tot, counts.numpy().view(np.uint64).copy(), local))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_reads", [1001, 64])
def test_two_rank_combine_matches_single_process(n_reads):
    from kmers_amd import dist as kd
    from oracle import oracle

    world, L, k, b = 2, 150, 31, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, L, k, b, q)) for r in range(world)]
    for p in procs:
        p.start()
    tot, counts, local0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference over the same (identically dirtied) shards
    exp = {"n_valid": 0, "sum_canon": 0, "xor_hash": 0, "sum_fw": 0}
    exp_counts = np.zeros(1 << b, np.uint64)
    covered = 0
    for r in range(world):
        start, cnt = kd.shard_range(n_reads, r, world)
        assert start == covered
        covered += cnt
        host = oracle.gen_reads(0x6B6D6572735F7631, start * L, cnt * L).copy()
        host[::997] = ord("N")
        s = oracle.canonical_reduce(host, cnt, L, k, hasher_k=k)
        exp["n_valid"] += s.n_valid
        exp["sum_canon"] = (exp["sum_canon"] + s.sum_canon) & kd.M64
        exp["sum_fw"] = (exp["sum_fw"] + s.sum_fw) & kd.M64
        exp["xor_hash"] ^= s.xor_hash
        exp_counts += oracle.histogram(host, cnt, L, k, k, b)
    assert covered == n_reads
    assert tot == exp
    assert (counts == exp_counts).all()
    assert tot["n_valid"] >= local0["n_valid"]


def test_shard_range_properties():
    from kmers_amd import dist as kd

    for n in (0, 1, 7, 8, 9, 1000, 10**9 + 7):
        for world in (1, 2, 3, 8):
            spans = [kd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_combine_wraps_mod_2_64_single_process():
    from kmers_amd import dist as kd

    # without a process group the local summary is returned unchanged
    loc = {"n_valid": 5, "sum_canon": kd.M64, "xor_hash": 1 << 63, "sum_fw": 7}
    assert kd.combine_summaries(loc) == loc


# ---- kmers_amd.api.Comm: rank 0's id travels with its status byte (a failure on rank 0 must reach every rank)

def _comm_id_worker(rank, world, port, status0, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace

    from kmers_amd import _lib
    from kmers_amd.api import Comm

    ident = bytes(range(128))
    raw = Comm._torch_broadcast(SimpleNamespace(ctx=None), (ident + bytes([status0])) if rank == 0 else None)
    out_q.put((rank, len(raw), raw[:_lib.COMM_ID_BYTES] == ident, raw[_lib.COMM_ID_BYTES]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("status0", [0, 3])
def test_comm_id_broadcast_carries_rank0_status(status0):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_id_worker, args=(r, world, port, status0, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [g[0] for g in got] == [0, 1]
    for _, n, same_id, st in got:
        assert n == 129 and same_id and st == status0
