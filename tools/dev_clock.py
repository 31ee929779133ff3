"""dev tool: where a small batch's scan kernel spends its time.  Needs the `clock` development build
(python tools/dev_variant.py clock --only kmx_bitslice.hip --patch tools/_patches/clock.patch), whose headline kernel leaves seven
s_memrealtime stamps (100 MHz) per wave behind the summary:
    0 entry  1 before the ticket  2 ticket in, first rows asked for  3 first tile packed, next rows asked for  4 tile loop left
    5 sums folded (before the block's barrier)  6 end
    python tools/dev_clock.py [n_reads,...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import devlib
devlib.use("clock")
from kmers_amd.api import Context, _ptr
from kmers_amd import _lib

ctx = Context(0)
k, L = 31, 150
sizes = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (10_000, 100_000, 1_000_000, 4_000_000)
for n in sizes:
    bases = ctx.gen_reads(L * n)
    n_waves = 4 * min(256 * 3, ((n + 63) // 64 + 3) // 4)
    out = torch.zeros(64 + 8 * n_waves + 64, dtype=torch.int64, device=ctx.device)
    r = ctx._reads(bases, n, L, None)
    for rep in range(6):
        out.zero_()
        torch.cuda.synchronize()
        ctx._ck(ctx.lib.kmx_canonical_reduce(ctx._h, C.byref(r), k, _lib.HASH_LEX, k, 0, _ptr(out)))
        ctx.synchronize()
        torch.cuda.synchronize()
    v = out.cpu().numpy().view(np.uint64)[64:64 + 8 * n_waves].reshape(n_waves, 8).astype(np.int64)
    v = v[v[:, 0] != 0]
    t0 = v[:, 0].min()
    # stamps in time order: 0 entry, 2 first rows asked for + ticket in, 3 first tile packed, 4 loop left, 7 totals gathered, 1 accumulators folded, 5 sums ready, 6 end
    order = [0, 2, 3, 4, 7, 1, 5, 6]
    names = {0: "entry", 2: "ticket in", 3: "first tile packed", 4: "loop left", 7: "plane totals gathered", 1: "accumulators folded", 5: "sums ready", 6: "end"}
    us = (v[:, :8] - t0) / 100.0
    print(f"n = {n}: {len(v)} waves stamped; microseconds from the first wave's entry (min / median / max over the waves)")
    for i in order:
        c = us[:, i]
        c = c[c > -1e6]
        print(f"   {i} {names[i]:24s} {c.min():7.2f} {np.median(c):7.2f} {c.max():7.2f}")
    d = np.diff(us[:, order], axis=1)
    print("   per wave, stage to stage (median): " + "  ".join(f"{np.median(d[:, i]):.2f}" for i in range(len(order) - 1)))
