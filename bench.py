#!/usr/bin/env python3
"""bench.py -- canonical k-mers/s at k=31 on 150 bp reads (BASELINE.json metric), MI355X.

One "step" = one pass of the hot path (kmx_canonical_reduce: encode + sliding window + reverse
complement + canonical min + wrapping-sum reduce) over this rank's shard of synthetic reads,
already resident in HBM.  N=1 workload = BASELINE configs[1]: 1e8 x 150 bp reads, k=31
(15.0 GB in, 1.2e10 canonical k-mers per step).  N>1: reads shard embarrassingly, one process
per GPU, the same 1e8 reads per GPU (weak scaling), no data-path collective; torch.distributed
(RCCL) is used only for the barrier / max-over-ranks timing and the checksum combine.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured-copy ceiling


def usable_cores() -> int:
    """host threads this process may really use: affinity mask capped by the cgroup cpu quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(host_sample, n_reads, L, k, seconds_target=20.0):
    """Time the CPU oracle (plain-C port of the reference's naive_impl streaming iterator,
    canonical_kmer_iterator.rs:42-116) on this box's host cores: 1 thread and all threads."""
    import numpy as np
    from oracle import oracle

    lib = oracle.lib(native=True)
    cores = usable_cores()

    def run(nthreads, reads_each, reps=1):
        outs = [0] * nthreads

        def work(i):
            lo = (i * reads_each) % max(n_reads - reads_each + 1, 1)
            for _ in range(reps):   # the sample is re-scanned when it is smaller than the time budget
                outs[i] += oracle.canonical_reduce(host_sample[lo * L:(lo + reads_each) * L], reads_each, L, k, native_lib=lib).n_valid

        ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        return sum(outs) / dt, dt

    # calibrate each leg on a small slice, then size its sample for ~seconds_target/2 of wall time
    per = L - k + 1
    rate1, _ = run(1, min(n_reads, 20_000))
    want_1t = max(20_000, rate1 * (seconds_target / 2) / per)
    reads_1t = int(min(n_reads, want_1t))
    reps_1t = max(1, int(round(want_1t / reads_1t)))
    rate1, dt1 = run(1, reads_1t, reps_1t)
    rate_mt, _ = run(cores, min(n_reads, 20_000))
    want_mt = max(20_000, rate_mt / cores * (seconds_target / 2) / per)
    reads_mt = int(min(n_reads // cores if n_reads >= cores * 20_000 else n_reads, want_mt))
    reps_mt = max(1, int(round(want_mt / reads_mt)))
    rate_mt, dt_mt = run(cores, reads_mt, reps_mt)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": rate_mt, "unit": "canonical k-mers/s", "cores": cores, "kind": "port",
        "value_1thread": rate1,
        "sample": f"{cores} threads x {reads_mt} reads x {reps_mt} passes x {L} bp (same synthetic stream as the GPU run), k={k}, "
                  f"{dt_mt:.1f}s; 1 thread x {reads_1t} reads x {reps_1t} passes {dt1:.1f}s; oracle = plain-C port of "
                  f"naive_impl CanonicalKmerIterator, gcc -O3 -march=native",
        "cpu_model": model,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads-per-gpu", type=int, default=100_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("--hash", action="store_true", help="also fold the LexHasher(k) word hash (BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--packed", action="store_true",
                    help="reads held as a 2-bit SeqVector (kmx_seqvec_canonical_reduce): 0.25 B per base from HBM (SURVEY 8f row f1; not the metric)")
    ap.add_argument("--histogram", type=int, default=0, metavar="LOG2_BUCKETS",
                    help="also time the bucket histogram + RCCL all-reduce (BASELINE configs[4]); reported in 'histogram'")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the kmx path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from kmers_amd import _lib
    from kmers_amd.api import SEED_DEFAULT, Context

    ctx = Context(local_rank)
    L, k, n = args.read_len, args.k, args.reads_per_gpu
    nbytes = n * L
    hasher, hk = (_lib.HASH_LEX, k) if args.hash else (_lib.HASH_NONE, 0)

    # synthetic reads generated on the device; rank r owns stream bytes [r*nbytes, (r+1)*nbytes)
    bases = ctx.gen_reads(nbytes, SEED_DEFAULT, rank * nbytes)
    out = ctx.empty(4, torch.int64)
    torch.cuda.synchronize()

    two_word = k > 32   # [u64;2] k-mers (BASELINE configs[2], k=63): kmx_canonical_reduce2, build-defined order/hash
    if two_word:
        import ctypes as C

        out = ctx.empty(5, torch.int64)
        rd = ctx._reads(bases, n, L, None)

    words = None
    if args.packed:
        if two_word:
            raise SystemExit("--packed: single-word k-mers only")
        words = ctx.seqvec_from_bytes(bases)
        torch.cuda.synchronize()

    def step():
        if words is not None:
            ctx.seqvec_canonical_reduce(words, n, L, k, hasher, hk, 0, out=out, sync=False)
        elif two_word:
            ctx._ck(ctx.lib.kmx_canonical_reduce2(ctx._h, C.byref(rd), k, int(args.hash), C.c_void_p(out.data_ptr())))
        else:
            ctx.canonical_reduce_async(bases, n, L, k, hasher, hk, 0, out=out)

    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(ctx.stream)
        step()
        b.record(ctx.stream)
    barrier()
    elapsed = time.perf_counter() - t0

    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    summ = out.cpu().numpy().view(np.uint64)
    n_valid, sum_canon = int(summ[0]), int(summ[1])

    from kmers_amd import dist as kd

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())
    tot = kd.combine_summaries({"n_valid": n_valid, "sum_canon": sum_canon, "xor_hash": int(summ[2]) if not two_word else 0, "sum_fw": 0},
                               device=ctx.device)   # wrapping add / xor of the per-shard summaries
    total_kmers_per_step = tot["n_valid"]

    hist_info = None
    if args.histogram:
        b = args.histogram
        counts = torch.zeros(1 << b, dtype=torch.int64, device=ctx.device)
        ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b, counts=counts)   # warm-up
        counts.zero_()
        barrier()
        t0 = time.perf_counter()
        ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b, counts=counts)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        kd.allreduce_histogram(counts)       # the path's only real collective: all-reduce(sum, int64) over RCCL/xGMI
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hist_info = {"log2_buckets": b, "scan_ms": (t1 - t0) * 1e3, "allreduce_ms": (t2 - t1) * 1e3,
                     "total_count": int(counts.sum().item()), "expect": total_kmers_per_step}

    if rank == 0:
        value = total_kmers_per_step * args.steps / elapsed_max
        # algorithmic bytes per launch: L bytes read per read (L/4 when the reads are 2-bit packed), writes negligible
        algo_bytes = float(nbytes) / (4.0 if args.packed else 1.0)
        achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9
        # parity spot-check against the CPU oracle on the head of this rank's shard (outside the timed region)
        from oracle import oracle

        n_chk = min(n, 200_000)
        host = bases[: n_chk * L].cpu().numpy()
        if two_word:
            o = oracle.canonical_reduce2(host, n_chk, L, k, with_hash=args.hash)
            g = ctx.canonical_reduce2(bases[: n_chk * L], n_chk, L, k, with_hash=args.hash)
            parity = tuple(getattr(g, f) for f, _ in g._fields_) == tuple(getattr(o, f) for f, _ in o._fields_)
        else:
            o = oracle.canonical_reduce(host, n_chk, L, k, hasher_k=hk)
            g = ctx.canonical_reduce(bases[: n_chk * L], n_chk, L, k, hasher, hk, 0)
            parity = (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash if args.hash else 0)
        parity = parity and n_valid == n * max(L - k + 1, 0)
        traffic = None
        try:   # HBM bytes per launch from the committed PMC passes (same kernel, same workload only)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pt = json.load(f)
            if (k, L, n, args.hash, args.packed) == (31, 150, 100_000_000, False, False):
                traffic = pt["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        res = {
            "metric": "canonical k-mers/sec at k=31, 150 bp reads; HBM GB/s vs peak",
            "value": value,
            "unit": "canonical k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"k={k} {'canonicalize from a 2-bit SeqVector' if args.packed else 'encode+canonicalize'}{'+lex-hash' if args.hash else ''} (reduce mode), "
                            f"{n} x {L} bp synthetic reads per GPU" + ("" if args.packed else " (BASELINE configs[1])"),
                "reads_per_gpu": n, "read_len": L, "k": k, "parallelism": f"shard{world}",
                "bytes_per_gpu": nbytes,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "kernel": ("kmx::scan_bitsliced_kernel<%d,%d,*>" % (k, 10 if L <= 160 else 16)) if (13 <= k <= 31 or (33 <= k <= 63 and k % 2 == 1 and L <= 160)) and L <= 256 else ("kmx::scan_uniform_kernel" if k <= 31 else "kmx::reduce2_generic_kernel"),
                "avg_kernel_ms": avg_kernel_ms, "min_kernel_ms": min(kernel_ms),
                "median_kernel_ms": sorted(kernel_ms)[len(kernel_ms) // 2], "algorithmic_bytes_per_launch": algo_bytes,
                "frac_of_measured_copy_ceiling_6290": achieved / 6290.0,
            },
            "parity_vs_oracle": "ok" if parity else "MISMATCH",
            "checksum": f"{tot['sum_canon']:#018x}",
        }
        if hist_info is not None:
            res["histogram"] = hist_info
        if world == 1 and not args.no_cpu_baseline and not two_word:
            n_s = min(n, 4_000_000)
            res["cpu_baseline"] = cpu_baseline(bases[: n_s * L].cpu().numpy(), n_s, L, k)
        print(json.dumps(res), flush=True)
        if not parity:
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
