"""dev tool: time kmx_fastx_parse on a synthetic FASTQ / FASTA image (a 64k-record block tiled on the device)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, ctypes as C
from _timing import warm
from kmers_amd.api import Context, _ptr
from fastx_cases import fastq_text, fasta_text

ctx = Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(1)
for name, block in (("fastq 100..150 bp", fastq_text(rng, 65536, 100, 150)), ("fastq 150 bp", fastq_text(rng, 65536, fixed=150)),
                    ("fasta 10..50 kbp, 80 columns", fasta_text(rng, 400, 10000, 50000, width=80, blank=0))):
    text = ctx.to_device(block).repeat(reps)
    n = text.numel()
    nr, nb = C.c_uint64(0), C.c_uint64(0)
    ctx._ck(ctx.lib.kmx_fastx_parse(ctx._h, _ptr(text), n, 0, None, None, 0, C.byref(nr), C.byref(nb)))
    bases = ctx.empty(nb.value, torch.uint8)
    offsets = ctx.empty(nr.value + 1, torch.int64)
    warm(lambda: ctx._ck(ctx.lib.kmx_fastx_parse(ctx._h, _ptr(text), n, 0, _ptr(bases), _ptr(offsets), nr.value, C.byref(nr), C.byref(nb))))
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx._ck(ctx.lib.kmx_fastx_parse(ctx._h, _ptr(text), n, 0, _ptr(bases), _ptr(offsets), nr.value, C.byref(nr), C.byref(nb)))
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    # the tiled image must give the first block's reads `reps` times over (parity itself: tests/test_gpu_fastx.py)
    off = offsets.cpu().numpy().view(np.uint64)
    per = nr.value // reps
    assert nr.value % reps == 0 and nb.value % reps == 0 and int(off[-1]) == nb.value
    r0 = (reps - 1) * per
    assert np.array_equal(off[r0:] - off[r0], off[:per + 1])
    assert torch.equal(bases[-(nb.value // reps):], bases[:nb.value // reps])
    print(f"{name:32s} text {n/1e9:6.2f} GB  reads {nr.value:10d}  bases {nb.value/1e9:6.2f} GB  {ms:8.3f} ms  text {n/ms/1e6:7.0f} GB/s  (2x text + bases: {(2*n+nb.value)/ms/1e6:7.0f} GB/s)")
    del text, bases, offsets
