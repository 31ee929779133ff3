"""dev tool: kmx_canonical_reduce (k=31, 150 bp) when a fraction of the reads holds an N"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _timing import warm
from kmers_amd.api import Context

ctx = Context(0)
n, L, k = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 150, 31
FRACS = [float(x) for x in os.environ["FRACS"].split(",")] if os.environ.get("FRACS") else (0.0, 0.001, 0.005, 0.02, 0.1)
for frac in FRACS:
    bases = ctx.gen_reads(n * L)
    nd = int(n * frac)
    if nd:
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        rd = torch.randperm(n, device="cuda", generator=g)[:nd]
        pos = torch.randint(0, L, (nd,), device="cuda", generator=g)
        bases[rd * L + pos] = ord("N")
    out = ctx.canonical_reduce(bases, n, L, k)
    warm(lambda: ctx.canonical_reduce_async(bases, n, L, k))
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.canonical_reduce_async(bases, n, L, k); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    print(f"{100*frac:5.1f} % of the reads hold an N: {ms:7.3f} ms  {n*L/ms/1e6:6.0f} GB/s   n_valid {out.n_valid}")
    if os.environ.get("HIST"):   # the word-domain kernel (bucket histogram, 2^HIST buckets) on the same input
        hb = int(os.environ["HIST"])
        cnt = ctx.histogram(bases, n, L, k, 1, k, hb)
        warm(lambda: ctx.histogram(bases, n, L, k, 1, k, hb, counts=cnt), at_least=6)
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ctx.histogram(bases, n, L, k, 1, k, hb, counts=cnt); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        cnt.zero_()   # (the warm-up and the timed calls accumulate into it: the check is of ONE call)
        ctx.histogram(bases, n, L, k, 1, k, hb, counts=cnt)
        print(f"        histogram 2^{hb}: {sorted(ts)[1]:7.3f} ms   total {'ok' if int(cnt.sum().item()) == out.n_valid else 'WRONG'}")
        del cnt
    del bases
# the same through the ragged layout (an offsets array, frame 160)
if os.environ.get("FRACS"):
    sys.exit(0)
import numpy as np
d_off = ctx.to_device((np.arange(n + 1, dtype=np.uint64) * np.uint64(L)))
for frac in (0.0, 0.005, 0.02):
    bases = ctx.gen_reads(n * L)
    nd = int(n * frac)
    if nd:
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        rd = torch.randperm(n, device="cuda", generator=g)[:nd]
        pos = torch.randint(0, L, (nd,), device="cuda", generator=g)
        bases[rd * L + pos] = ord("N")
    out = ctx.canonical_reduce(bases, n, 160, k, offsets=d_off)
    warm(lambda: ctx.canonical_reduce_async(bases, n, 160, k, 0, 0, 0, d_off))
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.canonical_reduce_async(bases, n, 160, k, 0, 0, 0, d_off); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    print(f"ragged, {100*frac:5.1f} % of the reads hold an N: {ms:7.3f} ms  {n*L/ms/1e6:6.0f} GB/s   n_valid {out.n_valid}")
    del bases
