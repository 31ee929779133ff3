"""Development only (round 5): localise a wrong sum_canon of the k-mer scan against the oracle.

    [KMX_DEV_LIB=NAME] python tools/dev_bisect_sum.py [L=1000] [n=150000] [k=31] [mode=uniform|ragged2L|trim]

Runs the reduce on n reads, compares every field with the oracle, prints the wrapping difference of the sums (its bit pattern says
which term of the closed form is off: a multiple of MASK[k] = the popcount of the masks, a single 2^(2t+b) = one class counter,
...), repeats the call (is the wrong value stable?), then halves the read range until the smallest failing prefix / the failing
64-read tiles are found.  Works from any tree that has kmers_amd/ and oracle/ (run from that tree's root)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
try:
    import devlib

    devlib.from_env()
except ImportError:
    pass
from kmers_amd import _lib  # noqa: E402
from kmers_amd.api import Context  # noqa: E402
from oracle import oracle  # noqa: E402

M64 = (1 << 64) - 1
args = dict(a.split("=") for a in sys.argv[1:])
L = int(args.get("L", 1000))
n = int(args.get("n", 150000))
k = int(args.get("k", 31))
mode = args.get("mode", "uniform")
fb = int(args.get("fb", L))   # first byte of the synthetic stream (bench.py: 0)
ctx = Context(0)
bases = ctx.gen_reads(n * L, first_byte=fb)
host = bases.cpu().numpy()


def layout(lo, hi):
    """(device bases, n_reads, read_len argument, offsets or None, host slice, oracle offsets or None) of the reads [lo, hi)"""
    if mode == "uniform":
        return bases[lo * L:hi * L], hi - lo, L, None, host[lo * L:hi * L], None
    if mode == "ragged2L":     # pairs of reads as one ragged read of 2 L bases behind an offsets array, bound 1 << 16
        m = (hi - lo) // 2
        off = (np.arange(m + 1, dtype=np.uint64) * np.uint64(2 * L))
        return bases[lo * L:(lo + 2 * m) * L], m, 1 << 16, off, host[lo * L:(lo + 2 * m) * L], off
    if mode == "trim":         # every 50th read loses its last 1..100 bases (bound L)
        rng = np.random.default_rng(lo * 7919 + hi)
        lens = np.full(hi - lo, L, dtype=np.uint64)
        idx = np.arange(0, hi - lo, 50)
        lens[idx] = L - rng.integers(1, min(100, L - k), size=len(idx)).astype(np.uint64)
        # reads stay where they are (gaps behind the trimmed ones are not allowed: pack them)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        src = host[lo * L:hi * L].reshape(hi - lo, L)
        packed = np.concatenate([src[i, :int(lens[i])] for i in range(hi - lo)]) if (hi - lo) <= 4096 else None
        if packed is None:
            keep = (np.arange(L)[None, :] < lens[:, None])
            packed = src[keep]
        return ctx.to_device(packed), hi - lo, L, off, packed, off
    raise SystemExit(mode)


def run(lo, hi, reps=1):
    d, m, rl, off, h, ooff = layout(lo, hi)
    doff = ctx.to_device(off) if off is not None else None
    gs = []
    for _ in range(reps):
        g = ctx.canonical_reduce(d, m, rl, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=doff)
        gs.append((g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw))
    if ooff is None:
        o = oracle.canonical_reduce(h, m, rl, k, hasher_k=k)
    else:
        o = oracle.canonical_reduce(h, m, 0, k, hasher_k=k, offsets=ooff)
    return gs, (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)


def show(tag, g, o):
    names = ("n_valid", "sum_canon", "xor_hash", "sum_fw")
    bad = [nm for nm, a, b in zip(names, g, o) if a != b]
    print(f"{tag}: {'ok' if not bad else 'MISMATCH ' + ','.join(bad)}")
    if bad:
        d = (g[1] - o[1]) & M64
        mask = (1 << (2 * k)) - 1
        print(f"   sum_canon gpu {g[1]:#018x} oracle {o[1]:#018x} diff {d:#018x} (-diff {(-d) & M64:#018x})")
        for sign, v in (("+", d), ("-", (-d) & M64)):
            if v % mask == 0 and v // mask < (1 << 40):
                print(f"   diff = {sign}{v // mask} x MASK[k]  (popcount-of-masks term / window count)")
            if v & (v - 1) == 0:
                print(f"   diff = {sign}2^{v.bit_length() - 1}")
        print(f"   n_valid diff {g[0] - o[0]}, xor diff {g[2] ^ o[2]:#x}, sum_fw diff {(g[3] - o[3]) & M64:#x}")
    return not bad


gs, o = run(0, n, reps=4)
print(f"# L={L} n={n} k={k} mode={mode} fb={fb}; 4 calls: {'identical' if len(set(gs)) == 1 else 'DIFFER between calls: ' + str(gs)}")
ok = show("whole", gs[0], o)
if ok and len(set(gs)) == 1:
    print("nothing to bisect")
    ctx.close()
    sys.exit(0)
# smallest failing prefix (multiples of 64 reads)
lo, hi = 0, n
while hi - lo > 64:
    mid = lo + ((hi - lo) // 2 + 63) // 64 * 64
    if mid >= hi:
        break
    g, oo = run(0, mid, reps=2)
    bad = any(x != oo for x in g)
    print(f"  prefix [0,{mid}): {'FAILS' if bad else 'ok'}")
    if bad:
        hi = mid
    else:
        lo = mid
print(f"# smallest failing prefix ends in ({lo},{hi}]")
# disjoint blocks: which fail on their own?
for blk in (64 * 1024, 64 * 64, 64):
    bad_blocks = []
    rng_hi = min(n, hi + blk)
    for a in range(0, rng_hi, blk):
        b = min(a + blk, n)
        g, oo = run(a, b)
        if g[0] != oo:
            bad_blocks.append(a)
        if len(bad_blocks) >= 8:
            break
    print(f"  blocks of {blk} reads failing on their own (first 8 up to read {rng_hi}): {bad_blocks}")
    if not bad_blocks:
        break
ctx.close()
