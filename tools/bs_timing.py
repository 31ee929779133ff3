"""dev tool: the per-phase cycle breakdown of the bit-sliced kernel, from a build with the instrumentation patch:
    python tools/dev_variant.py bstiming --only kmx_bitslice.hip --patch tools/patches/bs_timing.patch
    KMX_DEV_LIB=bstiming python tools/bs_timing.py [n_reads] [reps] [ragged]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devlib
devlib.from_env()
import numpy as np, torch, ctypes as C
from kmers_amd.api import Context
from kmers_amd import _lib
ctx = Context(0)
n, L, k = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 150, 31
bases = ctx.gen_reads(n * L)
NWV = 4096
out = torch.zeros(8 + NWV * 16, dtype=torch.int64, device="cuda")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else ""
ragged = mode in ("ragged", "trim")   # ragged: the same reads behind an offsets array (frame 160); trim: 2 % of the reads trimmed to 36..149 bases, bound 150
hint = 160
d_off = ctx.to_device((np.arange(n + 1, dtype=np.uint64) * np.uint64(L))) if ragged else None
if mode == "trim":
    rng = np.random.default_rng(1)
    lens = np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = ctx.gen_reads(int(offs[-1]))
    d_off = ctx.to_device(offs)
    hint = 150
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record()
    if ragged:
        ctx.canonical_reduce_async(bases, n, hint, k, 0, 0, 0, d_off, out=out)
    else:
        ctx.canonical_reduce_async(bases, n, L, k, out=out)
    b.record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
print("kernel ms per launch (events): min %.3f median %.3f max %.3f -> %.0f GB/s at median" % (min(ms), sorted(ms)[len(ms)//2], max(ms), n * L / sorted(ms)[len(ms)//2] / 1e6))
v = out.cpu().numpy().view(np.uint64)
d = v[8:8 + NWV * 16].reshape(NWV, 16).astype(np.float64)
raw5 = v[8:8 + NWV * 16].reshape(NWV, 16)[:, 5]
start_abs = (raw5 >> np.uint64(20)).astype(np.float64)
d[:, 5] = (raw5 & np.uint64((1 << 20) - 1)).astype(np.float64)
keep = d[:, 5] > 0
start_abs = start_abs[keep]
d = d[keep]
st = (start_abs - start_abs.min()) / 1e5
en = st + d[:, 7] / 1e5
print("waves", len(d), "start ms: min %.3f p50 %.3f p90 %.3f max %.3f | end ms: min %.3f p50 %.3f max %.3f" % (st.min(), np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en), en.max()))
print("late starters (>0.1 ms):", int((st > 0.1).sum()))
print("per-wave lifetime ms (first 64 waves): min %.3f max %.3f" % (d[:,7].min()/1e5, d[:,7].max()/1e5))
tiles = d[:, 5]
names = ["A0 wait loads", "A encode+lds", "ticket take/issue", "C tail: plane stores, totals, fence", "D pass 2", "", "", "", "D pass 1 + late rows", "B realign", "B ragged: read ends + validity planes", "C transposes"]
print("tiles per wave", tiles.mean())
q = np.percentile(tiles, [0, 10, 50, 90, 100])
print("tiles per wave: min %d p10 %d median %d p90 %d max %d  (the ticket queue lets a wave that is served more often take more tiles)" % tuple(q))
tot = 0
for i, nme in enumerate(names):
    if not nme:
        continue
    c = (d[:, i] / tiles).mean()
    tot += c
    print(f"{nme:40s} {c:10.0f} cycles/tile")
print("total", tot)
clk = (d[:, 6] / (d[:, 7] / 100e6)).mean() / 1e9
print(f"effective shader clock during the kernel: {clk:.3f} GHz (cycle counter vs 100 MHz wall clock); wave lifetime {d[:,7].mean()/100e6*1e3:.3f} ms")
