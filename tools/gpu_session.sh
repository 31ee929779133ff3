# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r6w; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 - <<'PY' | tee $O/reduce2_offsets.txt
import numpy as np, torch, sys
sys.path.insert(0, 'tools')
from _timing import warm
from kmers_amd.api import Context
ctx = Context(0)
n, L = 20_000_000, 150
bases = ctx.gen_reads(n * L)
off = ctx.to_device((np.arange(n + 1, dtype=np.uint64) * L))
lens = np.full(n, L); lens[::50] = 120
off2 = ctx.to_device(np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64))
def t(f):
    warm(f); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[2]
import ctypes as C
from kmers_amd.api import _ptr
out = ctx.empty(5, torch.int64)
for k in (63, 33):
    for name, o, hint in (("no offsets", None, L), ("offsets, all 150, bound 150", off, L), ("offsets, all 150, no bound", off, 0), ("offsets, 2 % trimmed, bound 150", off2, L)):
        r = ctx._reads(bases, n, hint, o)
        ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_reduce2(ctx._h, C.byref(r), k, 1, _ptr(out))))
        print(f"k={k} 2e7 x 150 bp  {name:34s} {ms:8.3f} ms  {n * L / ms / 1e6:7.0f} GB/s")
PY
