"""dev tool: host-side time of kmx_histogram calls (is the GPU waiting for the host?)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmers_amd.api import Context
from kmers_amd import _lib
ctx = Context(0)
for n in (100_000_000, 125_000_000):
    L, k, b = 150, 31, 20
    bases = ctx.gen_reads(n * L)
    counts = torch.zeros(1 << b, dtype=torch.int64, device=ctx.device)
    torch.cuda.synchronize()
    for rep in range(5):
        t0 = time.perf_counter()
        ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b, counts=counts)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n={n} call {rep}: host {1e3*(t1-t0):9.2f} ms, then sync {1e3*(t2-t1):9.2f} ms", flush=True)
    del bases
