"""dev tool: kmx_histogram (k=31, Lex hasher, 2^20 and 2^12 buckets) and kmx_canonical_windows when 0.5 % of the reads hold an N"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from kmers_amd.api import Context, _ptr
from kmers_amd import _lib

ctx = Context(0)
L, k = 150, 31
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
def t(f, reps=3):
    f(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for frac in (0.0, 0.005):
    bases = ctx.gen_reads(n * L)
    nd = int(n * frac)
    if nd:
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        rd = torch.randperm(n, device="cuda", generator=g)[:nd]
        pos = torch.randint(0, L, (nd,), device="cuda", generator=g)
        bases[rd * L + pos] = ord("N")
    for b in (20, 12):
        counts = torch.zeros(1 << b, dtype=torch.int64, device="cuda")
        ms = t(lambda: ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b, counts=counts))
        print(f"{100*frac:4.1f} % dirty reads: histogram b={b}: {ms:8.3f} ms")
    canon = ctx.empty(n * (L - k + 1), torch.int64)
    r = ctx._reads(bases, n, L, None)
    ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows(ctx._h, C.byref(r), None, k, None, None, _ptr(canon), None)))
    print(f"{100*frac:4.1f} % dirty reads: canonical words materialised: {ms:8.3f} ms")
    del bases, canon
