# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7n; mkdir -p $O
cd $GRAFT_REPO_ROOT
export KMX_LIB_VARIANT=wpa
timeout 1200 python -m pytest tests -x -q -m gpu -k "windows" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in default wpa; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  echo "[$v]"; python3 tools/bench_windows.py 1500000 1000 2>/dev/null | grep "flags"
  python3 - <<'PY'
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, 'tools')
from _timing import warm
from kmers_amd.api import Context, _ptr
ctx = Context(0)
n, k = 20_000_000, 31
lens = np.where(np.random.default_rng(1).random(n) < 0.02, np.random.default_rng(2).integers(36, 150, n), 150)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
wo = np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))]).astype(np.uint64)
rb = ctx.gen_reads(int(off[-1])); d_off, d_wo = ctx.to_device(off), ctx.to_device(wo)
tot = int(wo[-1])
fw, rc, cn, fl = (ctx.empty(tot, torch.int64) for _ in range(3)) , None, None, None
a = [ctx.empty(tot, torch.int64) for _ in range(3)]; f = ctx.empty(tot, torch.uint8)
r = ctx._reads(rb, n, 150, d_off)
def t(fn):
    warm(fn); ts = []
    for _ in range(5):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record(); fn(); y.record(); torch.cuda.synchronize(); ts.append(x.elapsed_time(y))
    return sorted(ts)[2]
ms = t(lambda: ctx._ck(ctx.lib.kmx_canonical_windows(ctx._h, C.byref(r), _ptr(d_wo), k, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(f))))
print(f"ragged 2 % trimmed, 2e7 reads: fw+rc+canon+flags {ms:8.3f} ms = {25*tot/ms/1e6:6.0f} GB/s written")
PY
done | tee $O/wpa.txt
