"""dev tool: kmx_canonical_reduce on LONG ragged reads (an offsets array, lengths in the thousands): the segment path (a length bound
above 256: kmx_segments.hip) against the per-read path (no bound), and uniform reads of the same mean length"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _timing import warm
from kmers_amd.api import Context

ctx = Context(0)
total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 6_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
rng = np.random.default_rng(2)
print(f"== k = {k}, ~{total / 1e9:.1f} G bases per batch")
for name, lo, hi in (("300..1000", 300, 1000), ("1000..20000", 1000, 20000), ("10000..50000", 10000, 50000)):
    n = int(total / ((lo + hi) / 2))
    lens = rng.integers(lo, hi + 1, n)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    tb = int(offsets[-1])
    bases = ctx.gen_reads(tb)
    d_off = ctx.to_device(offsets)
    exp = int(np.maximum(lens - k + 1, 0).sum())
    for label, hint, reps in (("segments (bound 2^20)", 1 << 20, 5), ("per-read path (no bound)", 0, 1)):
        if k > 32:     # two-word k: kmx_canonical_reduce2 (it synchronises; the time includes the summary's way back)
            call = lambda: ctx.canonical_reduce2(bases, n, hint, k, with_hash=False, offsets=d_off)
        else:
            call = lambda: ctx.canonical_reduce_async(bases, n, hint, k, 0, 0, 0, d_off)
        out = ctx.canonical_reduce2(bases, n, hint, k, offsets=d_off) if k > 32 else ctx.canonical_reduce(bases, n, hint, k, offsets=d_off)
        if reps > 1:
            warm(call, at_least=3)
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); call(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = sorted(ts)[len(ts) // 2]
        print(f"lengths {name:14s} {n:9d} reads  {label:26s} {ms:9.3f} ms  {tb/ms/1e6:7.0f} GB/s = {tb/ms/1e6/8000:.3f} of 8 TB/s   n_valid {'ok' if out.n_valid == exp else 'WRONG'}")
    del bases, d_off
