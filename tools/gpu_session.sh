# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7s; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "reduce2 or fuzz" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 - <<'PY'
import numpy as np, torch, sys, ctypes as C
sys.path.insert(0, 'tools')
from _timing import warm
from kmers_amd.api import Context, _ptr
ctx = Context(0)
n = 40_000_000
def t(f):
    warm(f); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[2]
out = ctx.empty(5, torch.int64)
lens = np.full(n, 250)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
total = int(offsets[-1]); bases = ctx.gen_reads(total); d_off = ctx.to_device(offsets)
for k in (63, 33):
    r = ctx._reads(bases, n, 250, d_off)
    ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_reduce2(ctx._h, C.byref(r), k, 1, _ptr(out))))
    print(f"k={k} 4e7 reads  all 250, bound 250   {ms:8.3f} ms  {total / ms / 1e6:7.0f} GB/s = {total / ms / 8e9:.3f} of the roofline")
PY
