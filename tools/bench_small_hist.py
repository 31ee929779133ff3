"""dev tool: kmx_histogram (2^20 buckets) on small batches -- what a call costs beside its kernels (uniform 150-bp reads, k = 31)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _timing  # noqa: F401
from kmers_amd.api import Context

ctx = Context(0)
k, L = 31, 150
for n in (100_000, 1_000_000, 4_000_000, 16_000_000):
    bases = ctx.gen_reads(L * n)
    f = lambda: ctx.histogram(bases, n, L, k, 1, k, 20)
    for _ in range(12):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"n = {n:>9}: call {ts[len(ts)//2]:8.1f} us (best {ts[0]:8.1f});  at 14.6 ms per 1e8 reads: {n * 14.6e-5:7.1f} us")
