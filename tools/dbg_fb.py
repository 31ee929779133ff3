import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kmers_amd.api import Context
ctx = Context(0)
n = 2_000_000
for lens in (np.full(n, 150), np.random.default_rng(1).integers(100, 161, n)):
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = ctx.gen_reads(int(off[-1]))
    out = ctx.canonical_reduce(bases, n, 160, 31, offsets=ctx.to_device(off))
    v = out.n_valid
    print("tiles", n // 64, "bad-encode", (v >> 40) & 0xFFF, "not-fits", v >> 52, "n", v & ((1 << 40) - 1))
