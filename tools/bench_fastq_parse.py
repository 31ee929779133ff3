"""dev tool: kmx_fastx_parse alone on a FASTQ image in HBM (150-bp records, 5.5 GB by default), n timed calls -- short enough to sit
   under rocprofv3 (--kernel-trace --stats, or one --pmc group per run: tools/pmc_fastq.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import _timing  # noqa: F401  (KMX_DEV_LIB=NAME: a development build)
from kmers_amd.api import Context

ctx = Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 256          # x 65536 reads
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
trimmed = "trimmed" in sys.argv[3:]
bound = 65536 * reps if "onepass" in sys.argv[3:] else None      # single call, buffers for the bound (with tools/patches/fastq_one_pass.patch: FASTQ in one pass)
rng = np.random.default_rng(7)
lens = np.full(65536, 150)
if trimmed:
    lens = np.where(rng.random(65536) < 0.02, rng.integers(36, 150, 65536), 150)
recs = []
for i, ln in enumerate(lens):
    seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(ln)))
    recs.append(b"@SRR000000.%d %d/1\n" % (i, i) + seq + b"\n+\n" + b"I" * int(ln) + b"\n")
block = np.frombuffer(b"".join(recs), dtype=np.uint8)
text = ctx.to_device(np.ascontiguousarray(block)).repeat(reps)
for _ in range(3):
    bases, offsets = ctx.fastx_parse(text, 1, bound)
torch.cuda.synchronize()
ts = []
for _ in range(calls):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); bases, offsets = ctx.fastx_parse(text, 1, bound); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
print(f"fastq {'trimmed' if trimmed else 'untrimmed'} ({'one call, buffers for a bound' if bound else 'counts + emit'}): {text.numel()/1e9:.2f} GB of text, {offsets.numel()-1} reads, {bases.numel()/1e9:.2f} GB of bases: "
      f"parse median {ts[len(ts)//2]:.3f} ms, best {ts[0]:.3f} ms ({text.numel()/ts[len(ts)//2]/1e6:.0f} GB/s of text)")
