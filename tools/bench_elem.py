"""dev tool: the element-wise word calls (revcomp / canonical / hash of u64 k-mer words) against a plain copy of the same arrays"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from _timing import warm
from kmers_amd.api import Context, _ptr
from kmers_amd import _lib

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
k = 31
words = torch.randint(0, 2**62, (n,), dtype=torch.int64, device=ctx.device)
out = torch.empty_like(words)
isc = ctx.empty(n, torch.uint8)
def t(f, reps=5):
    warm(f); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ctx.stream); f(); b.record(ctx.stream); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
with torch.cuda.stream(ctx.stream):
    ms = t(lambda: out.copy_(words))
    print(f"torch copy            {ms:7.3f} ms  {16*n/ms/1e6:6.0f} GB/s")
    ms = t(lambda: ctx._ck(ctx.lib.kmx_revcomp_words(ctx._h, _ptr(words), n, k, _ptr(out))))
    print(f"kmx_revcomp_words     {ms:7.3f} ms  {16*n/ms/1e6:6.0f} GB/s")
    ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_words(ctx._h, _ptr(words), n, k, _ptr(out), None)))
    print(f"kmx_canonical_words   {ms:7.3f} ms  {16*n/ms/1e6:6.0f} GB/s  (words only)")
    ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_words(ctx._h, _ptr(words), n, k, _ptr(out), _ptr(isc))))
    print(f"kmx_canonical_words   {ms:7.3f} ms  {17*n/ms/1e6:6.0f} GB/s  (words + flags)")
    ms = t(lambda: ctx._ck(ctx.lib.kmx_hash_words(ctx._h, _ptr(words), n, _lib.HASH_LEX, k, _ptr(out))))
    print(f"kmx_hash_words (lex)  {ms:7.3f} ms  {16*n/ms/1e6:6.0f} GB/s")
