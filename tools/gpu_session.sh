# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7g; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "windows" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 - <<'PY'
import sys, torch, ctypes as C
sys.path.insert(0, 'tools')
from _timing import warm
from kmers_amd.api import Context, _ptr
ctx = Context(0)
def t(f):
    warm(f); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[2]
for L, n in ((300, 5_000_000), (1000, 1_500_000), (10000, 150_000)):
    k = 31
    bases = ctx.gen_reads(n * L)
    tot = n * (L - k + 1)
    canon = ctx.empty(tot, torch.int64)
    r = ctx._reads(bases, n, L, None)
    ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows(ctx._h, C.byref(r), None, k, None, None, _ptr(canon), None)))
    print(f"k={k} L={L}: canon only {ms:8.3f} ms = {8*tot/ms/1e6:6.0f} GB/s written, {tot/ms/1e6:6.1f} G k-mers/s")
    fw, rc, fl = ctx.empty(tot, torch.int64), ctx.empty(tot, torch.int64), ctx.empty(tot, torch.uint8)
    ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows(ctx._h, C.byref(r), None, k, _ptr(fw), _ptr(rc), _ptr(canon), _ptr(fl))))
    print(f"k={k} L={L}: fw+rc+canon+flags {ms:8.3f} ms = {25*tot/ms/1e6:6.0f} GB/s written")
    del fw, rc, fl
    b1 = ctx.gen_reads(n * L + 16)[1:1 + n * L]      # (a base that is not 16-byte aligned: the lane-per-read kernel, what every such call took before)
    r1 = ctx._reads(b1, n, L, None)
    ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows(ctx._h, C.byref(r1), None, k, None, None, _ptr(canon), None)))
    print(f"k={k} L={L}: canon only, lane-per-read kernel {ms:8.3f} ms = {8*tot/ms/1e6:6.0f} GB/s written")
    del b1
    del bases, canon
PY
