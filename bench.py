#!/usr/bin/env python3
"""bench.py -- canonical k-mers/s at k=31 on 150 bp reads (BASELINE.json metric), MI355X.

One "step" = one pass of the hot path over this rank's shard of synthetic reads, already resident in HBM:
kmx_canonical_reduce (encode + sliding window + reverse complement + canonical min + wrapping-sum reduce), or, for
--config 4, kmx_histogram + the RCCL all-reduce of the bucket counters.

  --config 1    (default) BASELINE configs[1]: k=31, 1e8 x 150 bp per GPU (15.0 GB in, 1.2e10 canonical k-mers per step)
  --config 2    BASELINE configs[2]: k=21 or k=63 (-k 21 | -k 63; [u64;1] vs [u64;2] storage), 1e8 x 150 bp
  --config 3    BASELINE configs[3]: k=31 + LexHasher word hash folded in, 1.25e8 reads per GPU (1e9 over 8 GPUs)
  --config 4    BASELINE configs[4]: k=31 bucket histogram (2^20 buckets) per GPU, then all-reduce over RCCL/xGMI,
                scan and all-reduce timed separately

N > 1: reads shard embarrassingly, one process per GPU, the same number of reads per GPU (weak scaling), no data-path
collective; torch.distributed (backend "nccl" = RCCL) carries the barrier, the max-over-ranks timing and the checksum
combine; the histogram all-reduce goes through libkmx's own RCCL communicator (kmx_comm_*, include/kmx.h) when it can be
created, through torch.distributed otherwise (stated in the line).  A bare `python bench.py --gpus N` (no torchrun)
starts the N ranks itself as fresh child processes; under torchrun (WORLD_SIZE set) it is one of the ranks.
Every line states `rccl_ranks` = the number of ranks a real all-reduce saw, and the run fails if that is not --gpus.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
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


# ------------------------------------------------------------------------------------------------ CPU baseline

def cpu_baseline(host_sample, n_reads, L, k, seconds_target=24.0):
    """Time the CPU oracle (plain-C port of the reference's naive_impl path) on this box's host cores.
    Three consumer shapes, each at 1 thread and at all usable threads:
      (ii)  streaming CanonicalKmerIterator + canonical word sum (canonical_kmer_iterator.rs:42-116) -- `value`
      (i)   per-window Kmer::from + sum == compute_naive (benches/simple_benchmark.rs:14-22)
      (iii) (ii) + hash_one(LexHasherState::new(k)) folded in (hash.rs:10-20,60-71)"""
    import ctypes as C

    import numpy as np
    from oracle import oracle

    lib = oracle.lib(native=True)
    cores = usable_cores()
    per = L - k + 1

    def run_iter(hk):
        def one(lo, reads_each):
            return oracle.canonical_reduce(host_sample[lo * L:(lo + reads_each) * L], reads_each, L, k, hasher_k=hk, native_lib=lib).n_valid
        return one

    def run_naive(lo, reads_each):   # compute_naive over the same bytes taken as one string (the bench's own shape: one long string)
        seg = host_sample[lo * L:(lo + reads_each) * L]
        out = C.c_uint64()
        lib.kmo_compute_naive(seg.ctypes.data_as(C.POINTER(C.c_uint8)), seg.size, k, C.byref(out))
        return seg.size - k + 1

    def timed(fn, nthreads, reads_each, reps=1):
        outs = [0] * nthreads

        def work(i):
            lo = (i * reads_each) % max(n_reads - reads_each + 1, 1)
            for _ in range(reps):   # the sample is re-scanned when it is smaller than the time budget
                outs[i] += fn(lo, reads_each)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        return sum(outs) / dt, dt

    def leg(fn, nthreads, seconds):
        """calibrate on a small slice, then size the sample for ~`seconds` of wall time"""
        rate, _ = timed(fn, nthreads, min(n_reads, 20_000))
        want = max(20_000, rate / nthreads * seconds / per)
        reads = int(min(n_reads // nthreads if n_reads >= nthreads * 20_000 else n_reads, want))
        reps = max(1, int(round(want / reads)))
        rate, dt = timed(fn, nthreads, reads, reps)
        return rate, dt, reads, reps

    s = seconds_target / 8.0
    it0, itk = run_iter(0), run_iter(k)
    r_mt, dt_mt, reads_mt, reps_mt = leg(it0, cores, 2 * s)
    r_1t, dt_1t, reads_1t, reps_1t = leg(it0, 1, 2 * s)
    n_mt, dtn_mt, _, _ = leg(run_naive, cores, s)
    n_1t, dtn_1t, _, _ = leg(run_naive, 1, s)
    h_mt, dth_mt, _, _ = leg(itk, cores, s)
    h_1t, dth_1t, _, _ = leg(itk, 1, s)
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
        "value": r_mt, "unit": "canonical k-mers/s", "cores": cores, "kind": "port",
        "value_1thread": r_1t,
        "variants": {
            "streaming_iterator": {"all_threads": r_mt, "one_thread": r_1t, "ref": "canonical_kmer_iterator.rs:42-116"},
            "per_window_compute_naive": {"all_threads": n_mt, "one_thread": n_1t, "unit": "k-mers/s (not canonicalised)",
                                         "ref": "benches/simple_benchmark.rs:14-22"},
            "streaming_iterator_lex_hash": {"all_threads": h_mt, "one_thread": h_1t, "ref": "hash.rs:10-20,60-71"},
        },
        "sample": f"{cores} threads x {reads_mt} reads x {reps_mt} passes x {L} bp (same synthetic stream as the GPU run), k={k}, "
                  f"{dt_mt:.1f}s; 1 thread x {reads_1t} reads x {reps_1t} passes {dt_1t:.1f}s; per-window and lex-hash variants "
                  f"{dtn_mt + dtn_1t + dth_mt + dth_1t:.1f}s more; oracle = plain-C port of naive_impl, gcc -O3 -march=native",
        "cpu_model": model,
    }


def cpu_baseline2(host_sample, n_reads, L, k, with_hash, seconds_target=12.0):
    """[u64;2] k-mers (k = 33..64, BASELINE configs[2]): the oracle's kmo_canonical_reduce2 -- the same iterator walk as
    canonical_kmer_iterator.rs:42-70 on 128-bit words (build-defined, as the GPU path: the reference has no two-word
    naive_impl) -- at all usable threads and at one."""
    from oracle import oracle

    oracle.lib()
    cores = usable_cores()
    per = L - k + 1

    def timed(nthreads, reads_each, reps):
        outs = [0] * nthreads

        def work(i):
            lo = (i * reads_each) % max(n_reads - reads_each + 1, 1)
            for _ in range(reps):
                outs[i] += oracle.canonical_reduce2(host_sample[lo * L:(lo + reads_each) * L], reads_each, L, k, with_hash=with_hash).n_valid

        ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        return sum(outs) / dt, dt

    def leg(nthreads, seconds):
        rate, _ = timed(nthreads, min(n_reads, 20_000), 1)
        want = max(20_000, rate / nthreads * seconds / per)
        reads = int(min(n_reads // nthreads if n_reads >= nthreads * 20_000 else n_reads, want))
        reps = max(1, int(round(want / reads)))
        rate, dt = timed(nthreads, reads, reps)
        return rate, dt, reads, reps

    r_mt, dt_mt, reads_mt, reps_mt = leg(cores, seconds_target * 0.6)
    r_1t, dt_1t, reads_1t, reps_1t = leg(1, seconds_target * 0.4)
    return {
        "value": r_mt, "unit": "canonical k-mers/s", "cores": cores, "kind": "port", "value_1thread": r_1t,
        "sample": f"{cores} threads x {reads_mt} reads x {reps_mt} passes x {L} bp (same synthetic stream as the GPU run), k={k} ([u64;2]), "
                  f"{dt_mt:.1f}s; 1 thread x {reads_1t} reads x {reps_1t} passes {dt_1t:.1f}s; oracle kmo_canonical_reduce2, gcc -O3",
    }


# ------------------------------------------------------------------------------------------------ launch of N ranks

def passthrough_args(argv):
    return [a for a in argv]


def spawn_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without torchrun: start N fresh child processes, one per GPU.  This parent never
    touches the GPU (and never execs): it only waits.  Rank 0's stdout is ours, so its JSON line is the output."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "KMX_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = set(range(args.gpus))
    deadline = time.monotonic() + float(os.environ.get("KMX_BENCH_SPAWN_TIMEOUT_S", "3600"))   # a hung rank must not hang the caller
    while alive:
        if time.monotonic() > deadline:
            sys.stderr.write("bench.py: ranks still running at the wall-clock limit; stopping them\n")
            for o in alive:
                procs[o].kill()      # exactly the processes started above
            return 5
        for r in list(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0 and rc == 0:
                rc = c
                sys.stderr.write(f"bench.py: rank {r} exited with {c}; stopping the other ranks\n")
                for o in alive:
                    procs[o].terminate()   # exactly the processes started above
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------ HBM traffic (PMC)

def measure_traffic(argv, log=sys.stderr):
    """HBM bytes per launch of the scan kernel from the TCC counters, measured in THIS run: two child processes of
    this script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need separate passes, and no tracing flag is mixed
    in), each launching the calibration stream (kmx_calib_stream_read: a known byte count in the scan's own access
    pattern) and the scan.  Read correction = known bytes / counted bytes of the calibration kernel in the same pass
    (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of a wide coalesced read).  None if rocprofv3 is absent
    or a pass fails -- the bench line then carries traffic: null."""
    import csv
    import glob

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof:
        return None, "rocprofv3 not found"
    res = {}
    tmp = tempfile.mkdtemp(prefix="kmx_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "t", "--",
                   sys.executable, os.path.abspath(__file__), *argv, "--pmc-child"]
            # own session: on a timeout the profiler AND the python child under it are stopped (a surviving child would keep
            # its 15 GB of HBM and its launches next to the timed region of this run)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                _, err_txt = pr.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, 9)    # the group this call created, nothing else
                except OSError:
                    pass
                pr.wait()
                return None, f"rocprofv3 --pmc {ctr}: timed out, pass stopped"

            class _R:
                returncode, stderr = pr.returncode, err_txt
            r = _R
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} exited with {r.returncode}: {r.stderr[-300:]}"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {ctr}: no counter csv"
            per = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != ctr:
                        continue
                    name = row["Kernel_Name"]
                    key = "calib" if "calib_stream_read_kernel" in name else ("scan" if ("scan_" in name or "reduce" in name or "hist" in name or "sweep_flagged" in name) else None)
                    if key:
                        per.setdefault(key, {}).setdefault(name, []).append(float(row["Counter_Value"]))
            res[ctr] = per
        return res, None
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def traffic_from_counters(res, algo_read_bytes, n_step_launch_sets):
    """counter csv digest -> bytes per step.  Counter values are KiB.  All kernels of a step are summed
    (the bit-sliced scan + the kernel that rolls the reads it blanked out; the passes of the partitioned histogram)."""
    fetch, write = res["FETCH_SIZE"], res["WRITE_SIZE"]
    if "calib" not in fetch or "scan" not in fetch:
        return None
    calib_counts = [v for vs in fetch["calib"].values() for v in vs]
    calib_kib = sum(calib_counts) / len(calib_counts)
    corr = algo_read_bytes / (calib_kib * 1024.0)
    fetch_kib = sum(sum(vs) for vs in fetch["scan"].values()) / n_step_launch_sets
    write_kib = sum(sum(vs) for vs in write.get("scan", {}).values()) / n_step_launch_sets
    return {
        "bytes_per_step": fetch_kib * 1024.0 * corr + write_kib * 1024.0,
        "FETCH_SIZE_KiB_per_step": fetch_kib, "WRITE_SIZE_KiB_per_step": write_kib,
        "read_correction": corr,
        "read_correction_from": f"kmx_calib_stream_read over the same buffer in the same pass: {algo_read_bytes:.0f} B known / {calib_kib * 1024.0:.0f} B counted",
        "write_correction": 1.0,
    }


# ------------------------------------------------------------------------------------------------ the bench itself

def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-steps", type=int, default=30,
                    help="untimed steps of the same workload run before the W warm-up steps (the device settles after the calibration kernel)")
    ap.add_argument("--config", default="1", choices=["1", "2", "3", "4"], help="BASELINE.json configs[N] (see the module docstring)")
    ap.add_argument("--reads-per-gpu", type=int, default=None)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-k", type=int, default=None)
    ap.add_argument("--hash", action="store_true", help="also fold the LexHasher(k) word hash (BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=24.0, help="CPU work the cpu_baseline leg is sized for (rank 0, after the timed region)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic")
    ap.add_argument("--sustain-steps", type=int, default=1000,
                    help="after the timed region: this many more back-to-back steps, timed as one interval (the package reaches its "
                         "power cap within ~1 s, so a 20-step window flatters the kernel); reported in 'sustained'")
    ap.add_argument("--packed", action="store_true",
                    help="reads held as a 2-bit SeqVector (kmx_seqvec_canonical_reduce): 0.25 B per base from HBM (SURVEY 8f row f1; not the metric)")
    ap.add_argument("--histogram", type=int, default=0, metavar="LOG2_BUCKETS",
                    help="config 1-3: additionally time one bucket histogram + all-reduce after the timed region")
    ap.add_argument("--dist-single", action="store_true", help="create the nccl process group even for one rank (exercises RCCL on a 1-GPU box)")
    ap.add_argument("--spawn-selftest", action="store_true", help=argparse.SUPPRESS)   # tests: ranks only rendezvous (gloo, no GPU) and count themselves
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)        # internal: the run under rocprofv3 --pmc
    args = ap.parse_args(argv)
    if args.config == "2":
        args.k = args.k or 21
        if args.k not in (21, 63):
            ap.error("--config 2 is k=21 or k=63")
    if args.config in ("3", "4"):
        args.k = args.k or 31
        args.reads_per_gpu = args.reads_per_gpu or 125_000_000
        args.hash = args.hash or args.config == "3"
    args.k = args.k or 31
    args.reads_per_gpu = args.reads_per_gpu or 100_000_000
    if args.config == "4" and not args.histogram:
        args.histogram = 20
    return args


def selftest_worker(args):
    """ranks rendezvous over gloo and count themselves (CPU only: the launch logic, not the GPU path)"""
    import torch.distributed as dist

    from kmers_amd import dist as kd

    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ranks = kd.rccl_rank_count()
    ok = ranks == args.gpus == world
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rccl_ranks": ranks, "backend": "gloo",
                          "spawned_by_bench": os.environ.get("KMX_BENCH_SPAWNED") == "1"}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 4


def worker(args, traffic_raw=None, traffic_err=None):
    # stdout carries ONE line, the JSON of rank 0.  Native libraries write there as well (RCCL prints a version banner
    # when a communicator is created, flushed at exit -- after our line): everything but that line goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: n_gpus would be misreported")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the kmx path has no CPU fallback")
    # KMX_BENCH_TEST_SHARED_GPU=1 (tests only, stated in the line): every rank runs on cuda:0 and the process group is gloo --
    # the whole N > 1 control flow (barriers, summary combine, histogram exchange and its fallback, verdict broadcast) on a
    # box with ONE GPU.  RCCL refuses two ranks on one device, so this is the only way to run that code on the GPU pool.
    shared_gpu = os.environ.get("KMX_BENCH_TEST_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.dist_single:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    coll_dev = None if shared_gpu else torch.device("cuda", local_rank)   # where the small control-plane tensors live

    from kmers_amd import _lib
    from kmers_amd import dist as kd
    from kmers_amd.api import SEED_DEFAULT, Comm, Context

    ctx = Context(local_rank)
    # ranks that a real RCCL all-reduce sees: must be what the line will claim as n_gpus
    rccl_ranks = kd.rccl_rank_count(coll_dev)
    if rccl_ranks != args.gpus:
        sys.stderr.write(f"bench.py: all-reduce saw {rccl_ranks} ranks, --gpus {args.gpus}\n")
        if dist is not None:
            dist.destroy_process_group()
        sys.exit(4)

    L, k, n = args.read_len, args.k, args.reads_per_gpu
    nbytes = n * L
    hasher, hk = (_lib.HASH_LEX, k) if args.hash else (_lib.HASH_NONE, 0)
    hist_mode = args.config == "4"

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

    # the histogram exchange: libkmx's own RCCL communicator (the C ABI a Rust / C++ host would use), torch.distributed otherwise
    comm, collective = None, "none (1 rank)"
    counts = None
    if args.histogram:
        counts = torch.zeros(1 << args.histogram, dtype=torch.int64, device=ctx.device)
        if dist is not None:
            try:
                if shared_gpu:
                    raise RuntimeError("ranks share one GPU (test mode): RCCL cannot build a communicator")
                comm = Comm(ctx, world, rank)
                collective = "kmx_histogram_allreduce: ncclAllReduce(ncclUint64, ncclSum) on libkmx's RCCL communicator"
            except Exception as e:   # noqa: BLE001 -- keep the run alive: the fallback is RCCL as well
                comm = None
                collective = f"torch.distributed all_reduce (nccl = RCCL); kmx_comm_create failed: {e}"
            flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # all ranks take the same route
            if int(flag.item()) == 0 and comm is not None:
                comm.close()
                comm = None
                collective = "torch.distributed all_reduce (nccl = RCCL); kmx_comm_create failed on another rank"

    def hist_scan():
        ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, args.histogram, counts=counts)

    def hist_exchange():
        if comm is not None:
            comm.histogram_allreduce(counts)
        elif dist is not None and shared_gpu:
            host = counts.cpu()
            kd.allreduce_histogram(host)
            counts.copy_(host)
        elif dist is not None:
            with torch.cuda.stream(ctx.stream):
                kd.allreduce_histogram(counts)

    ev_mid = []

    def step():
        if hist_mode:
            counts.zero_()
            hist_scan()
            e = torch.cuda.Event(enable_timing=True)
            e.record(ctx.stream)
            ev_mid.append(e)
            hist_exchange()
        elif words is not None:
            ctx.seqvec_canonical_reduce(words, n, L, k, hasher, hk, 0, out=out, sync=False)
        elif two_word:
            ctx._ck(ctx.lib.kmx_canonical_reduce2(ctx._h, C.byref(rd), k, int(args.hash), C.c_void_p(out.data_ptr())))
        else:
            ctx.canonical_reduce_async(bases, n, L, k, hasher, hk, 0, out=out)

    if args.pmc_child:   # under rocprofv3 --pmc: the calibration kernel (known bytes) and the step, nothing else
        cal = ctx.empty(1, torch.int64)
        for _ in range(3):
            ctx.calib_stream_read(bases, out=cal)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        return 0

    # Same-run ceiling of the HBM read stream (kmx_calib_stream_read: the scan's load shape, no compute), measured BEFORE the
    # timed region: it is part of every line, and ~70 ms of streaming also brings the device out of its idle power state --
    # from cold the first ten launches of any kernel run up to 40 % slow (tools/step_times.py), which W = 5 warm-up steps
    # alone do not cover.
    cal = ctx.empty(1, torch.int64)
    for _ in range(5):
        ctx.calib_stream_read(bases, out=cal)
    cevs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(25)]
    with torch.cuda.stream(ctx.stream):
        for a, b in cevs:
            a.record(ctx.stream)
            ctx.calib_stream_read(bases, out=cal)
            b.record(ctx.stream)
    torch.cuda.synchronize()
    stream_ms = sorted(a.elapsed_time(b) for a, b in cevs)[len(cevs) // 2]
    stream_gbps = float(nbytes) / (stream_ms * 1e-3) / 1e9

    # The read-only calibration kernel draws less power than the scan; the first ~15 scan launches after it run ~1 % slower
    # than the ones that follow (profiles/r03_step_times.txt), W = 5 alone ends in the middle of that.  These settle steps
    # are the same workload, untimed and reported ("settle_steps") -- the 1000-step "sustained" figure is the cross-check.
    with torch.cuda.stream(ctx.stream):
        for _ in range(args.settle_steps):
            step()
        for _ in range(args.warmup):
            step()
    ev_mid.clear()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(ctx.stream):
        for a, b in evs:
            a.record(ctx.stream)
            step()
            b.record(ctx.stream)
    barrier()
    elapsed = time.perf_counter() - t0

    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    scan_ms = [a.elapsed_time(m) for (a, _), m in zip(evs, ev_mid)] if hist_mode else kernel_ms
    xchg_ms = [m.elapsed_time(b) for (_, b), m in zip(evs, ev_mid)] if hist_mode else []
    avg_scan_ms = sum(scan_ms) / len(scan_ms)

    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    per_rank_s = [elapsed]
    if dist is not None:
        # every rank's own wall time of the K steps and its kernel average: the line's value uses the MAX, the spread says
        # how evenly the shards ran (scaling efficiency falls out of one run)
        mine = torch.tensor([elapsed, avg_kernel_ms], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_s = [float(x[0].item()) for x in allr]
        per_rank_kernel_ms = [float(x[1].item()) for x in allr]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    else:
        per_rank_kernel_ms = [avg_kernel_ms]
    elapsed_max = float(t.item())

    summary_combine_ms = None
    if hist_mode:
        total_count = int(counts.sum().item())            # after the all-reduce: k-mers of ALL ranks
        total_kmers_per_step = total_count
        tot = {"sum_canon": int(counts[: 1 << 10].sum().item())}
        n_valid_local = n * max(L - k + 1, 0)
    else:
        summ = out.cpu().numpy().view(np.uint64)
        n_valid_local, sum_canon = int(summ[0]), int(summ[1])
        t_comb = time.perf_counter()
        tot = kd.combine_summaries({"n_valid": n_valid_local, "sum_canon": sum_canon, "xor_hash": int(summ[2]) if not two_word else 0, "sum_fw": 0},
                                   device=coll_dev)   # wrapping add / xor of the per-shard summaries
        summary_combine_ms = (time.perf_counter() - t_comb) * 1e3   # the only exchange of configs[1..3]: 32 bytes per rank, once per job
        total_kmers_per_step = tot["n_valid"]

    # ---- outside the timed region: sustained run, same-run read ceiling, optional histogram, parity, CPU baseline
    sustained = None
    if args.sustain_steps > 0:
        ev_mid.clear()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        with torch.cuda.stream(ctx.stream):
            a.record(ctx.stream)
            for _ in range(args.sustain_steps):
                step()
            b.record(ctx.stream)
        barrier()
        ev_mid.clear()
        sus_ms = a.elapsed_time(b) / args.sustain_steps
        sustained = {"steps": args.sustain_steps, "ms_per_step": sus_ms}

    hist_info = None
    if args.histogram and not hist_mode:
        b_ = args.histogram
        hist_scan()   # warm-up (grows the work buffer)
        counts.zero_()
        barrier()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(ctx.stream):
            e0.record(ctx.stream)
            hist_scan()
            e1.record(ctx.stream)
            hist_exchange()
            e2.record(ctx.stream)
        torch.cuda.synchronize()
        hist_info = {"log2_buckets": b_, "scan_ms": e0.elapsed_time(e1), "allreduce_ms": e1.elapsed_time(e2),
                     "total_count": int(counts.sum().item()), "expect": total_kmers_per_step, "collective": collective}

    parity = True
    res = None
    if rank == 0:
        from oracle import oracle

        value = total_kmers_per_step * args.steps / elapsed_max
        # algorithmic bytes per launch: L bytes read per read (L/4 when the reads are 2-bit packed), writes negligible
        algo_bytes = float(nbytes) / (4.0 if args.packed else 1.0)
        achieved = algo_bytes / (avg_scan_ms * 1e-3) / 1e9
        # parity spot-check against the CPU oracle on the head of this rank's shard (outside the timed region)
        n_chk = min(n, 200_000)
        host = bases[: n_chk * L].cpu().numpy()
        if hist_mode:
            n_h = min(n, 20_000)
            o = oracle.histogram(host[: n_h * L], n_h, L, k, k, args.histogram)
            g = ctx.histogram(bases[: n_h * L], n_h, L, k, _lib.HASH_LEX, k, args.histogram).cpu().numpy().view(np.uint64)
            parity = bool((g == o).all()) and total_count == world * n * max(L - k + 1, 0)
        elif two_word:
            o = oracle.canonical_reduce2(host, n_chk, L, k, with_hash=args.hash)
            g = ctx.canonical_reduce2(bases[: n_chk * L], n_chk, L, k, with_hash=args.hash)
            parity = tuple(getattr(g, f) for f, _ in g._fields_) == tuple(getattr(o, f) for f, _ in o._fields_)
        else:
            o = oracle.canonical_reduce(host, n_chk, L, k, hasher_k=hk)
            g = ctx.canonical_reduce(bases[: n_chk * L], n_chk, L, k, hasher, hk, 0)
            parity = (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash if args.hash else 0)
        parity = parity and n_valid_local == n * max(L - k + 1, 0)
        traffic, traffic_detail = None, traffic_err
        if traffic_raw is not None:
            td = traffic_from_counters(traffic_raw, float(nbytes), 3)
            if td is not None:
                traffic, traffic_detail = td["bytes_per_step"], td
        bs_kernel = (13 <= k <= 31 or 33 <= k <= 64) and L <= 256
        seg_kernel = (13 <= k <= 31 or 33 <= k <= 64) and L > 256 and not args.packed
        if hist_mode:
            kernel_name = "kmx::scan_uniform_kernel<SinkHist*> (partition pass + per-partition tables)"
        elif bs_kernel:
            # the frame (packed dwords per read) launch_bs_any picks: 5 / 7 / 8 / 10 / 13 / 16 words for reads of up to 80 / 112 / 128 / 160 / 208 / 256 bases
            frame = 10 if (k > 32 or args.packed) and L <= 160 else 16 if args.packed else (13 if L <= 208 else 16) if k > 32 else \
                next(nw for nw, lmax in ((5, 80), (7, 112), (8, 128), (10, 160), (13, 208), (16, 256)) if L <= lmax)
            kernel_name = "kmx::scan_bitsliced_kernel<%d,%d,*>" % (k, frame)
        elif seg_kernel:
            # reads above 256 bases: equal overlapping segments on the uniform kernel (bs_seg_plan in kmx_bitslice_kernel.h)
            wr, t10 = L - k + 1, min(128, 160 - k)
            t_max = t10 if 32 < k <= 49 else min(192, 208 - k)
            n_seg = -(-wr // t_max)
            t_seg = -(-wr // n_seg)
            kernel_name = "kmx::scan_bitsliced_kernel<%d,%d,*,SEG> (%d segments of %d windows per read)" % (k, 10 if t_seg <= t10 else 13, n_seg, t_seg)
        else:
            kernel_name = "kmx::scan_uniform_kernel" if k <= 31 else "kmx::reduce2_generic_kernel"
        cfg_names = {"1": "BASELINE configs[1]", "2": "BASELINE configs[2]", "3": "BASELINE configs[3]", "4": "BASELINE configs[4]"}
        what = ("distinct-k-mer bucket histogram (2^%d buckets, LexHasher) + RCCL all-reduce" % args.histogram) if hist_mode else \
               ("canonicalize from a 2-bit SeqVector" if args.packed else "encode+canonicalize") + ("+lex-hash" if args.hash else "") + " (reduce mode)"
        res = {
            "metric": "canonical k-mers/sec at k=31, 150 bp reads; HBM GB/s vs peak",
            "value": value,
            "unit": "canonical k-mers/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": args.settle_steps,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "per_rank": {"ms_per_step_min": min(per_rank_s) / args.steps * 1e3, "ms_per_step_max": max(per_rank_s) / args.steps * 1e3,
                         "kernel_ms_min": min(per_rank_kernel_ms), "kernel_ms_max": max(per_rank_kernel_ms),
                         "summary_combine_ms": summary_combine_ms},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"k={k} {what}, {n} x {L} bp synthetic reads per GPU" + ("" if args.packed else f" ({cfg_names[args.config]})"),
                "reads_per_gpu": n, "read_len": L, "k": k, "parallelism": f"shard{world}",
                "bytes_per_gpu": nbytes, **({"TEST_MODE": "KMX_BENCH_TEST_SHARED_GPU: all ranks on cuda:0, gloo process group -- not a multi-GPU measurement"} if shared_gpu else {}), "launched_by": "bench.py (self-spawned ranks)" if os.environ.get("KMX_BENCH_SPAWNED") == "1" else ("torchrun" if world > 1 else "single process"),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "kernel": kernel_name,
                "avg_kernel_ms": avg_scan_ms, "min_kernel_ms": min(scan_ms),
                "median_kernel_ms": sorted(scan_ms)[len(scan_ms) // 2], "algorithmic_bytes_per_launch": algo_bytes,
                "frac_of_measured_copy_ceiling_6290": achieved / 6290.0,
                "same_run_stream_read_GBps": stream_gbps, "frac_of_same_run_stream_read": achieved / stream_gbps,
                "traffic_detail": traffic_detail,
            },
            "parity_vs_oracle": "ok" if parity else "MISMATCH",
            "checksum": f"{tot['sum_canon']:#018x}",
        }
        if sustained is not None:
            sustained["achieved_GBps"] = algo_bytes / (sustained["ms_per_step"] * 1e-3) / 1e9 if not hist_mode else None
            sustained["frac"] = sustained["achieved_GBps"] / HBM_PEAK_GBPS if not hist_mode else None
            sustained["value"] = total_kmers_per_step / (sustained["ms_per_step"] * 1e-3)
            res["sustained"] = sustained
        if hist_mode:
            res["histogram"] = {"log2_buckets": args.histogram, "scan_ms": avg_scan_ms, "allreduce_ms": sum(xchg_ms) / len(xchg_ms),
                                "total_count": total_count, "expect": world * n * max(L - k + 1, 0), "collective": collective}
        elif hist_info is not None:
            res["histogram"] = hist_info
        if not args.no_cpu_baseline:
            # rank 0 only, after the timed region, at every N (the other ranks wait at the verdict broadcast below)
            n_s = min(n, 4_000_000)
            sample = bases[: n_s * L].cpu().numpy()
            res["cpu_baseline"] = (cpu_baseline2(sample, n_s, L, k, args.hash, args.cpu_baseline_seconds / 2.0) if two_word
                                   else cpu_baseline(sample, n_s, L, k, args.cpu_baseline_seconds))
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    # every rank learns the verdict and leaves together (a lone sys.exit on rank 0 would strand the others in a collective)
    if dist is not None:
        v = torch.tensor([1 if parity else 0], dtype=torch.int64, device=coll_dev)
        dist.broadcast(v, src=0)
        parity = bool(int(v.item()))
        if comm is not None:
            comm.close()
        dist.barrier()
        dist.destroy_process_group()
    return 0 if parity else 3


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    have_world = "WORLD_SIZE" in os.environ
    if not have_world and args.gpus > 1:
        sys.exit(spawn_ranks(args, argv))       # before anything touches the GPU; children are fresh processes
    if args.spawn_selftest:
        sys.exit(selftest_worker(args))
    traffic_raw, traffic_err = None, "not measured (--no-traffic, N > 1, or a child of another run)"
    if not have_world and args.gpus == 1 and not args.no_traffic and not args.pmc_child and not args.dist_single:
        # two short child runs under rocprofv3 --pmc, BEFORE this process initialises the GPU
        traffic_raw, traffic_err = measure_traffic([a for a in argv if a not in ("--no-cpu-baseline",)] + ["--no-cpu-baseline", "--no-traffic"])
    sys.exit(worker(args, traffic_raw, traffic_err))


if __name__ == "__main__":
    main()
