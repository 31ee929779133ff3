"""dev tool: build libkmx with -DKMX_BS_TIMING and print the per-phase cycle breakdown of the bit-sliced kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from kmers_amd.api import Context
from kmers_amd import _lib
ctx = Context(0)
n, L, k = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 150, 31
bases = ctx.gen_reads(n * L)
NWV = 4096
out = torch.zeros(8 + NWV * 8, dtype=torch.int64, device="cuda")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record()
    ctx.canonical_reduce_async(bases, n, L, k, out=out)
    b.record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
print("kernel ms per launch (events): min %.3f median %.3f max %.3f -> %.0f GB/s at median" % (min(ms), sorted(ms)[len(ms)//2], max(ms), n * L / sorted(ms)[len(ms)//2] / 1e6))
v = out.cpu().numpy().view(np.uint64)
d = v[8:8 + NWV * 8].reshape(NWV, 8).astype(np.float64)
d = d[d[:, 5] > 0]
st = (d[:, 4] - d[:, 4].min()) / 1e5
en = st + d[:, 7] / 1e5
print("waves", len(d), "start ms: min %.3f p50 %.3f p90 %.3f max %.3f | end ms: min %.3f p50 %.3f max %.3f" % (st.min(), np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en), en.max()))
print("late starters (>0.1 ms):", int((st > 0.1).sum()))
print("per-wave lifetime ms (first 64 waves): min %.3f max %.3f" % (d[:,7].min()/1e5, d[:,7].max()/1e5))
tiles = d[:, 5]
names = ["A1 wait loads", "A2 encode+lds", "B realign", "C transpose", "D main loop"]
print("tiles per wave", tiles.mean())
tot = 0
for i, nme in enumerate(names):
    c = (d[:, i] / tiles).mean()
    tot += c
    print(f"{nme:16s} {c:10.0f} cycles/tile")
print("total", tot)
clk = (d[:, 6] / (d[:, 7] / 100e6)).mean() / 1e9
print(f"effective shader clock during the kernel: {clk:.3f} GHz (cycle counter vs 100 MHz wall clock); wave lifetime {d[:,7].mean()/100e6*1e3:.3f} ms")
# --- finish time vs placement
full = v[8:8 + NWV * 8].reshape(NWV, 8).astype(np.float64)
ok = full[:, 5] > 0
wid = np.nonzero(ok)[0]
en_all = (full[ok, 4] - full[ok, 4].min() + full[ok, 7]) / 1e5
blk = wid // 4
print("by XCD (block%8): ", " ".join("%.2f" % en_all[blk % 8 == x].mean() for x in range(8)))
print("by wave-in-block: ", " ".join("%.2f" % en_all[wid % 4 == x].mean() for x in range(4)))
nb = blk.max() + 1
print("by block third:   ", " ".join("%.2f" % en_all[(blk >= nb * i // 3) & (blk < nb * (i + 1) // 3)].mean() for i in range(3)))
print("per-block spread within block (max-min) mean: %.3f" % np.mean([en_all[blk == b].max() - en_all[blk == b].min() for b in range(0, nb, 7)]))
h, e = np.histogram(en_all, bins=12)
print("hist", list(h), ["%.2f" % x for x in e])
