"""dev tool: time kmx_seqvec_minimizers / kmx_minimizer_words on synthetic reads"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _timing import warm
from kmers_amd.api import Context
from kmers_amd import _lib

ctx = Context(0)
L = 150
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
bases = ctx.gen_reads(n * L)
words = ctx.seqvec_from_bytes(bases)
for k, w in ((31, 15), (21, 11), (31, 21)):
    tot = n * (L - k + 1)
    warm(lambda: ctx.seqvec_minimizers(words, n, L, k, w, _lib.HASH_LEX, w), at_least=8)
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.seqvec_minimizers(words, n, L, k, w, _lib.HASH_LEX, w); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[1]
    print(f"seqvec_minimizers k={k} w={w}: {ms:8.3f} ms  {tot/ms/1e6:7.1f} G k-mers/s  ({12*tot/ms/1e6:6.0f} GB/s written)")
# round 6: the same iterator over the ASCII reads themselves (kmx_minimizers), uniform and behind offsets
import numpy as np
for k, w in ((31, 15), (21, 11)):
    tot = n * (L - k + 1)
    f = lambda: ctx.minimizers(bases, n, L, k, w, _lib.HASH_LEX, w, check=False)
    warm(f, at_least=8)
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[1]
    print(f"minimizers (ASCII, uniform) k={k} w={w}: {ms:8.3f} ms  {tot/ms/1e6:7.1f} G k-mers/s  ({12*tot/ms/1e6:6.0f} GB/s written)")
rng = np.random.default_rng(1)
lens = np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150).astype(np.int64)
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
k, w = 31, 15
wins = np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))]).astype(np.uint64)
rb = ctx.gen_reads(int(offs[-1]))
d_off, d_win = ctx.to_device(offs), ctx.to_device(wins)
tot = int(wins[-1])
f = lambda: ctx.minimizers(rb, n, 150, k, w, _lib.HASH_LEX, w, offsets=d_off, win_offsets=d_win, check=False)
warm(f, at_least=8)
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = sorted(ts)[1]
print(f"minimizers (ASCII, 2 % trimmed behind offsets) k={k} w={w}: {ms:8.3f} ms  {tot/ms/1e6:7.1f} G k-mers/s  ({12*tot/ms/1e6:6.0f} GB/s written)")
