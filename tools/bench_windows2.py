"""dev tool: kmx_canonical_windows2 ([u64;2] k-mers materialised) against the single-word materialise and a plain fill"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from _timing import warm
from kmers_amd.api import Context, _ptr

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bases = ctx.gen_reads(n * L)
def t(f, reps=5):
    warm(f); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
r = ctx._reads(bases, n, L, None)
k1 = 31
tot1 = n * (L - k1 + 1)
c1 = ctx.empty(tot1, torch.int64)
ms1 = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows(ctx._h, C.byref(r), None, k1, None, None, _ptr(c1), None)))
print(f"k={k1} single word, canon only: {ms1:8.3f} ms = {8*tot1/ms1/1e6:6.0f} GB/s written, {tot1/ms1/1e6:6.1f} G k-mers/s")
del c1
for k in (33, 47, 63, 64):
    W = L - k + 1
    tot = n * W
    canon = ctx.empty(2 * tot, torch.int64)
    ms_fill = t(lambda: canon.fill_(7))
    ms_c = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows2(ctx._h, C.byref(r), None, k, None, None, _ptr(canon), None)))
    print(f"k={k}: canon only {ms_c:8.3f} ms = {16*tot/ms_c/1e6:6.0f} GB/s written, {tot/ms_c/1e6:6.1f} G k-mers/s   (fill of the same array {ms_fill:7.3f} ms = {16*tot/ms_fill/1e6:6.0f} GB/s)")
    fw, rc, fl = ctx.empty(2 * tot, torch.int64), ctx.empty(2 * tot, torch.int64), ctx.empty(tot, torch.uint8)
    ms_a = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows2(ctx._h, C.byref(r), None, k, _ptr(fw), _ptr(rc), _ptr(canon), _ptr(fl))))
    print(f"k={k}: fw+rc+canon+flags {ms_a:8.3f} ms = {49*tot/ms_a/1e6:6.0f} GB/s written")
    del canon, fw, rc, fl
# ragged reads (offsets + win_offsets; round 4: the tiled kernel with per-lane geometry)
import numpy as np
k = 63
for name, lens, hint in (("all 150, bound 150", np.full(n, 150), 150), ("2 % trimmed to 70..149, bound 150", np.where(np.random.default_rng(1).random(n) < 0.02, np.random.default_rng(2).integers(70, 150, n), 150), 150),
                         ("100..160 mix, bound 160", np.random.default_rng(3).integers(100, 161, n), 160)):
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    wo = np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))]).astype(np.uint64)
    rb = ctx.gen_reads(int(off[-1]))
    d_off, d_wo = ctx.to_device(off), ctx.to_device(wo)
    tot = int(wo[-1])
    canon = ctx.empty(2 * tot, torch.int64)
    rr = ctx._reads(rb, n, hint, d_off)
    ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows2(ctx._h, C.byref(rr), _ptr(d_wo), k, None, None, _ptr(canon), None)))
    print(f"k={k} ragged {name}: canon only {ms:8.3f} ms = {16*tot/ms/1e6:6.0f} GB/s written, {tot/ms/1e6:6.1f} G k-mers/s")
    del rb, canon, d_off, d_wo
# round 6: uniform reads of which 2 % hold an N, canon only (a dirty tile stays on the tiled path; the sweep zeroes the spoiled slots)
rng6 = np.random.default_rng(6)
dirty = ctx.gen_reads(n * L)
sel = np.nonzero(rng6.random(n) < 0.02)[0]
dirty[torch.from_numpy(sel * L + rng6.integers(0, L, len(sel))).to(dirty.device)] = ord("N")
rd = ctx._reads(dirty, n, L, None)
k = 63
tot = n * (L - k + 1)
canon = ctx.empty(2 * tot, torch.int64)
ms_d = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows2(ctx._h, C.byref(rd), None, k, None, None, _ptr(canon), None)))
print(f"k={k}, 2 % of the reads hold an N: canon only {ms_d:8.3f} ms = {16*tot/ms_d/1e6:6.0f} GB/s written")
