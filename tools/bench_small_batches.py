"""dev tool: kmx_canonical_reduce on small batches -- what a call costs beside its kernel (uniform 150-bp reads, k = 31; whole call incl. the summary's way back:
kmx_canonical_reduce + a copy of the summary, and kmx_canonical_reduce_host, which returns it)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _timing  # noqa: F401
from kmers_amd.api import Context
from kmers_amd import _lib

ctx = Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sizes = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (10_000, 100_000, 1_000_000, 4_000_000, 16_000_000)
for n in sizes:
    bases = ctx.gen_reads(L * n)
    ideal = n * L / 6.3e12 * 1e6
    want = None
    for name, f in (("reduce + copy back", lambda: ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, 0)),
                    ("reduce_host       ", lambda: ctx.canonical_reduce_host(bases, n, L, k, _lib.HASH_LEX, k, 0))):
        for _ in range(20):
            got = f()
        key = (got.n_valid, got.sum_canon, got.xor_hash, got.sum_fw)
        want = want or key
        assert key == want, (key, want)
        torch.cuda.synchronize()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
        ts.sort()
        print(f"n = {n:>9}  {name}: call {ts[len(ts)//2]:8.1f} us (best {ts[0]:8.1f});  the bytes at 6.3 TB/s: {ideal:7.1f} us")
