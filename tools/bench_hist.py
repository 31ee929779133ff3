"""dev tool: time kmx_histogram (k=31, Lex hasher) at several table sizes on synthetic reads and check the total count"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _timing import warm
from kmers_amd.api import Context
from kmers_amd import _lib

ctx = Context(0)
L, k = 150, 31
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [10, 14, 16, 20, 21, 22]
bases = ctx.gen_reads(n * L)
tot = n * (L - k + 1)
for b in bs:
    counts = torch.zeros(1 << b, dtype=torch.int64, device="cuda")
    ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b, counts=counts)
    torch.cuda.synchronize()
    ok = int(counts.sum().item()) == tot
    warm(lambda: ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b, counts=counts), at_least=6)
    ts = []
    for _ in range(3):
        counts.zero_()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b, counts=counts)
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e))
    ms = sorted(ts)[1]
    print(f"b={b:2d}  {ms:9.3f} ms  {tot/ms/1e6:8.1f} G k-mers/s  {n*L/ms/1e6:7.0f} GB/s in  total {'ok' if ok else 'WRONG'}  max bucket {int(counts.max().item())}")
