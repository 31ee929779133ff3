"""dev tool: time the non-headline entry points (materialise, histogram, [u64;2], ragged) on synthetic reads"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmers_amd.api import Context
from kmers_amd import _lib

ctx = Context(0)
L = 150
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
bases = ctx.gen_reads(n * L)

def timeit(name, fn, bytes_in, kmers, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[len(ts) // 2]
    print(f"{name:46s} {ms:9.3f} ms  {kmers/ms/1e6:8.1f} G k-mers/s  {bytes_in/ms/1e6:8.0f} GB/s in")

for k in (31, 21):
    W = L - k + 1
    tot = n * W
    canon = ctx.empty(tot, torch.int64); fw = ctx.empty(tot, torch.int64); rc = ctx.empty(tot, torch.int64); fl = ctx.empty(tot, torch.uint8)
    def win(c, f, r, g):
        rd = ctx._reads(bases, n, L, None)
        import ctypes as C
        ctx._ck(ctx.lib.kmx_canonical_windows(ctx._h, C.byref(rd), None, k, f.data_ptr() if f is not None else None, r.data_ptr() if r is not None else None, c.data_ptr() if c is not None else None, g.data_ptr() if g is not None else None))
    timeit(f"windows k={k} canon only (8 B/k-mer out)", lambda: win(canon, None, None, None), n * L, tot)
    timeit(f"windows k={k} canon+flags", lambda: win(canon, None, None, fl), n * L, tot)
    timeit(f"windows k={k} fw+rc+canon+flags (25 B/k-mer out)", lambda: win(canon, fw, rc, fl), n * L, tot)
    del canon, fw, rc, fl
    for b in (12, 20, 26):
        counts = torch.zeros(1 << b, dtype=torch.int64, device="cuda")
        timeit(f"histogram k={k} lex b={b}", lambda: ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b, counts=counts), n * L, tot)
k = 63
timeit("reduce2 k=63 ([u64;2], bit-sliced)", lambda: ctx.canonical_reduce2(bases, n, L, k, True), n * L, n * (L - k + 1))
off = ctx.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
timeit("reduce k=31 ragged offsets (bit-sliced)", lambda: ctx.canonical_reduce(bases, n, 0, 31, offsets=off), n * L, n * 120)
