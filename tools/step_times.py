import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from kmers_amd.api import Context
ctx = Context(0)
n, L, k = 100_000_000, 150, 31
bases = ctx.gen_reads(n * L)
out = ctx.empty(4, torch.int64)
torch.cuda.synchronize()
def run(nsteps, tag):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]
    for a, b in evs:
        a.record(); ctx.canonical_reduce_async(bases, n, L, k, 0, 0, 0, out=out); b.record()
    torch.cuda.synchronize()
    print(tag, " ".join("%.3f" % a.elapsed_time(b) for a, b in evs))
run(30, "cold:")
time.sleep(2.0)
run(30, "after 2 s idle:")
cal = ctx.empty(1, torch.int64)
time.sleep(2.0)
for _ in range(40): ctx.calib_stream_read(bases, out=cal)
run(30, "after 40 stream reads:")
