"""dev tool: kmx_canonical_windows (canonical words only) on reads of which a share holds an N -- what a dirty tile costs the materialise path"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from _timing import warm
from kmers_amd.api import Context, _ptr

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L, k = 150, 31
W = L - k + 1
rng = np.random.default_rng(1)
clean = ctx.gen_reads(n * L)
canon = ctx.empty(n * W, torch.int64)
for frac in (0.0, 0.001, 0.005, 0.02, 0.10):
    bases = clean.clone()
    if frac:
        reads = np.nonzero(rng.random(n) < frac)[0]
        idx = torch.from_numpy(reads * L + rng.integers(0, L, len(reads))).to(bases.device)
        bases[idx] = ord("N")
    r = ctx._reads(bases, n, L, None)
    f = lambda: ctx._ck(ctx.lib.kmx_canonical_windows(ctx._h, C.byref(r), None, k, None, None, _ptr(canon), None))
    warm(f)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    print(f"{100 * frac:5.1f} % of the reads hold an N: canon only {ms:8.3f} ms = {8 * n * W / ms / 1e6:6.0f} GB/s written")
