"""dev tool: kmx_canonical_reduce2 (two-word k) on reads behind an offsets array -- uniform behind offsets (the device-side gate
picks the uniform kernel), truly ragged (the ragged bit-sliced kernel in the 10-word frame), a bound above the frame (lane per read)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ctypes as C
import numpy as np, torch
from _timing import warm
from kmers_amd.api import Context, _ptr

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(1)


def t(f):
    warm(f); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[2]


out = ctx.empty(5, torch.int64)
print(f"== kmx_canonical_reduce2 with the hash fold, {n} reads behind an offsets array")
for name, lens, hint in (("all 150, bound 150", np.full(n, 150), 150),
                         ("2 % trimmed to 36..149, bound 150", np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150), 150),
                         ("100..160 mix, bound 160", rng.integers(100, 161, n), 160),
                         ("2 % trimmed, no bound", np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150), 0)):
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(offsets[-1])
    bases = ctx.gen_reads(total)
    d_off = ctx.to_device(offsets)
    for k in (63, 47, 33):
        r = ctx._reads(bases, n, hint, d_off)
        ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_reduce2(ctx._h, C.byref(r), k, 1, _ptr(out))))
        exp = int(np.maximum(lens - k + 1, 0).sum())
        got = int(out.cpu().numpy().view(np.uint64)[0])
        print(f"k={k}  {name:36s} {ms:8.3f} ms  {total / ms / 1e6:7.0f} GB/s = {total / ms / 8e9:.3f} of the roofline  n_valid {'ok' if got == exp else 'WRONG'}")
    del bases, d_off
