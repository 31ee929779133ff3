"""dev tool: kmx_canonical_reduce on ragged reads (offsets array): tiled word-domain kernel vs what the lengths allow"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _timing import warm
from kmers_amd.api import Context

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
rng = np.random.default_rng(1)
print(f"== k = {k}, {n} reads")
for name, lens, hint in (("all 150 (hint 160)", np.full(n, 150), 160), ("all 150 (hint 150)", np.full(n, 150), 150), ("all 150 (no hint)", np.full(n, 150), 0),
                         ("150 with 2% trimmed to 36..149 (hint 150)", np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150), 150),
                         ("uniform 100..160 (hint 160)", rng.integers(100, 161, n), 160),
                         ("150 with 2% trimmed to 36..149 (hint 160)", np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150), 160),
                         ("all 100 (hint 100)", np.full(n, 100), 100), ("uniform 50..100 (hint 100)", rng.integers(50, 101, n), 100),
                         ("uniform 50..100 (hint 160)", rng.integers(50, 101, n), 160)):
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(offsets[-1])
    bases = ctx.gen_reads(total)
    d_off = ctx.to_device(offsets)
    out = ctx.canonical_reduce(bases, n, hint, k, offsets=d_off)
    exp = int(np.maximum(lens - k + 1, 0).sum())
    warm(lambda: ctx.canonical_reduce_async(bases, n, hint, k, 0, 0, 0, d_off))
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.canonical_reduce_async(bases, n, hint, k, 0, 0, 0, d_off); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    print(f"{name:46s} {ms:8.3f} ms  {total/ms/1e6:7.0f} GB/s  {exp/ms/1e6:8.1f} G k-mers/s  n_valid {'ok' if out.n_valid == exp else 'WRONG'}")
    if os.environ.get("HIST"):
        b = int(os.environ["HIST"])
        cnt = ctx.histogram(bases, n, hint, k, 1, k, b, offsets=d_off)
        warm(lambda: ctx.histogram(bases, n, hint, k, 1, k, b, offsets=d_off, counts=cnt), at_least=6)
        ts = []
        for _ in range(3):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ctx.histogram(bases, n, hint, k, 1, k, b, offsets=d_off, counts=cnt); e.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(e))
        cnt.zero_()   # (one call's table: the warm-up and the timed calls accumulated into it)
        ctx.histogram(bases, n, hint, k, 1, k, b, offsets=d_off, counts=cnt)
        print(f"    histogram 2^{b}: {sorted(ts)[1]:8.3f} ms   total {'ok' if int(cnt.sum().item()) == exp else 'WRONG'}")
        del cnt
    del bases, d_off
