#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6g
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r6g/pytest_full.txt 2>&1
tail -5 gpurun_out/r6g/pytest_full.txt
python3 - > gpurun_out/r6g/ragged_small_k.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from kmers_amd.api import Context
ctx = Context(0)
n = 20_000_000
rng = np.random.default_rng(1)
lens = np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
bases = ctx.gen_reads(int(offsets[-1])); d_off = ctx.to_device(offsets)
for k in (8, 9, 12, 13, 31):
    for _ in range(5): ctx.canonical_reduce_async(bases, n, 150, k, 0, 0, 0, d_off)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.canonical_reduce_async(bases, n, 150, k, 0, 0, 0, d_off); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    print(f"k = {k:2d}: 2e7 reads of 150 bases, 2 % trimmed, behind offsets: {ms:.3f} ms = {int(offsets[-1]) / ms / 1e6:.0f} GB/s = {int(offsets[-1]) / ms / 8e9:.3f} of the roofline")
PY
