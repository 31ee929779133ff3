"""dev tool: ONE shape of reads behind offsets, a few calls of kmx_canonical_reduce -- for counter passes (rocprofv3 --pmc ... -- python3 tools/ragged_one.py SHAPE [n] [k])
SHAPE: trim2 (150 bases, 2 % trimmed to 36..149, bound 150) | one (150 bases, ONE read trimmed: every tile but one uniform, the ragged scan all the same) |
       mix (100..160, bound 160) | short (50..100, bound 100)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _timing  # noqa: F401
from kmers_amd.api import Context

shape = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
rng = np.random.default_rng(1)
if shape == "trim2":
    lens, hint = np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150), 150
elif shape == "one":
    lens, hint = np.full(n, 150), 150
    lens[n // 2] = 100
elif shape == "mix":
    lens, hint = rng.integers(100, 161, n), 160
else:
    lens, hint = rng.integers(50, 101, n), 100
ctx = Context(0)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
bases = ctx.gen_reads(int(offsets[-1]))
d_off = ctx.to_device(offsets)
for _ in range(6):
    ctx.canonical_reduce_async(bases, n, hint, k, 0, 0, 0, d_off)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ctx.canonical_reduce_async(bases, n, hint, k, 0, 0, 0, d_off); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(f"{shape}: {n} reads, {n // 64} tiles, {sorted(ts)[2]:.3f} ms per call, 11 calls")
