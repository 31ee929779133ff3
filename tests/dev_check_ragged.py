"""dev check (not collected by pytest): ragged-read summaries of the GPU path against the oracle, case by case"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kmers_amd.api import Context
from kmers_amd import _lib
from oracle import oracle as orc
ctx = Context(0)
rng = np.random.default_rng(1)
k = 31
def run(name, lens, hint=160):
    lens = np.asarray(lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(off[-1]))]
    o = orc.canonical_reduce(host, len(lens), 0, k, hasher_k=k, offsets=off)
    g = ctx.canonical_reduce(ctx.to_device(host), len(lens), hint, k, _lib.HASH_LEX, k, 0, offsets=ctx.to_device(off))
    print(f"{name:34s} n {g.n_valid == o.n_valid} sum {g.sum_canon == o.sum_canon} xor {g.xor_hash == o.xor_hash}   diff {(g.sum_canon - o.sum_canon) & (2**64-1):#x}")
run("all 150 x64", [150]*64)
run("all 150 x640", [150]*640)
run("all 144 x64 (16-aligned)", [144]*64)
run("all 140 x64", [140]*64)
run("all 160 x64", [160]*64)
run("150 except one 149", [150]*10+[149]+[150]*53)
run("150 except one 100", [150]*10+[100]+[150]*53)
run("150 except one 20", [150]*10+[20]+[150]*53)
run("random 100..160 x64", rng.integers(100,161,64))
run("random 31..160 x640", rng.integers(31,161,640))
run("all 150 x64 hint 150", [150]*64, 150)
pad = [150] * 64
run("mixed tile + pad", list(rng.integers(100,161,64)) + pad)
run("tile with len<k and 0 + pad", [150]*5 + [0, 1, 30, 31, 32] + [150]*54 + pad)
run("tile all short (<k) + pad", [20]*64 + pad)
run("tile with one 200 (frame 160) + pad", [150]*20 + [200] + [150]*43 + pad)
run("random 0..160 x6400", rng.integers(0,161,6400))
run("random 31..250 x6400 hint 256", rng.integers(31,251,6400), 256)
run("random 31..250 x6400 hint 0", rng.integers(31,251,6400), 0)
k = 21
run("k=21 random 0..160 x6400", rng.integers(0,161,6400))
run("k=21 random 21..100 x6400 hint 100", rng.integers(21,101,6400), 100)
