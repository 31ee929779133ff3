"""dev tool: per-step times of the headline scan in bench.py's order (stream-read calibration, W warm-up steps, barrier, K timed steps)"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from kmers_amd.api import Context
ctx = Context(0)
n, L, k = 100_000_000, 150, 31
bases = ctx.gen_reads(n * L)
out = ctx.empty(4, torch.int64)
cal = ctx.empty(1, torch.int64)
torch.cuda.synchronize()
def run(nsteps, tag):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]
    with torch.cuda.stream(ctx.stream):
        for a, b in evs:
            a.record(ctx.stream); ctx.canonical_reduce_async(bases, n, L, k, 0, 0, 0, out=out); b.record(ctx.stream)
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    print(tag, "avg %.3f |" % (sum(ts) / len(ts)), " ".join("%.3f" % t for t in ts))
for rep in range(3):
    for _ in range(30): ctx.calib_stream_read(bases, out=cal)
    torch.cuda.synchronize()
    with torch.cuda.stream(ctx.stream):
        for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5): ctx.canonical_reduce_async(bases, n, L, k, 0, 0, 0, out=out)
    torch.cuda.synchronize()
    run(20, "timed 20:")
    run(20, "next 20: ")
    run(200, "next 200:")
