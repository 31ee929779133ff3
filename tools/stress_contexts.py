"""dev tool: context life cycle and host threads (see tools/README.md)"""
import sys, os, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from kmers_amd.api import Context
from kmers_amd import _lib
free0 = torch.cuda.mem_get_info()[0]
for i in range(30):
    c = Context(0)
    b = c.gen_reads(2_000_000 * 150)
    s = c.canonical_reduce(b, 2_000_000, 150, 31, _lib.HASH_LEX, 31, 0)
    h = c.histogram(b, 2_000_000, 150, 31, 1, 31, 20)
    assert int(h.sum().item()) == s.n_valid
    del b, h
    c.close()
torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("free before/after 30 contexts (MB):", free0 >> 20, free1 >> 20)
assert free0 - free1 < (512 << 20)
# two contexts on two host threads
res = {}
def work(tag):
    c = Context(0)
    b = c.gen_reads(5_000_000 * 150, first_byte=tag)
    out = []
    for _ in range(20):
        out.append(c.canonical_reduce(b, 5_000_000, 150, 31, _lib.HASH_LEX, 31, 0).sum_canon)
        h = c.histogram(b, 5_000_000, 150, 31, 1, 31, 16)
        out.append(int(h.sum().item()))
    res[tag] = out
    c.close()
ts = [threading.Thread(target=work, args=(t,)) for t in (1, 2, 3)]
[t.start() for t in ts]; [t.join() for t in ts]
for t, o in res.items():
    assert len(set(o[0::2])) == 1 and len(set(o[1::2])) == 1, t
print("threads ok", {t: o[:2] for t, o in res.items()})
