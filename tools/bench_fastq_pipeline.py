"""dev tool: a FASTQ file image in HBM -> k-mer results, stage by stage (everything on the device; the image is the only input):
   kmx_fastx_parse -> kmx_reads_length_range -> kmx_canonical_reduce (uniform kernels if every read has the same length,
   else the ragged ones with the tight length bound) -> kmx_histogram (2^20 buckets).  Two inputs: untrimmed 150 bp
   reads with an N in 0.5 % of them, and the same reads with 2 % trimmed to 36..149 bp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from _timing import warm
from kmers_amd.api import Context
from kmers_amd import _lib

ctx = Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 256          # x 65536 reads
k = 31
rng = np.random.default_rng(7)


def timed(f, n=3):
    warm(f)
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2], r


def make_block(lens):
    recs = []
    for i, ln in enumerate(lens):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(ln)))
        recs.append(b"@SRR000000.%d %d/1\n" % (i, i) + seq + b"\n+\n" + b"I" * int(ln) + b"\n")
    return np.frombuffer(b"".join(recs), dtype=np.uint8)


for name, lens in (("150 bp, untrimmed", np.full(65536, 150)),
                   ("150 bp, 2 % trimmed to 36..149", np.where(rng.random(65536) < 0.02, rng.integers(36, 150, 65536), 150))):
    block = make_block(lens)
    if isinstance(block, (bytes, bytearray)):
        block = np.frombuffer(bytes(block), dtype=np.uint8)
    text = ctx.to_device(np.ascontiguousarray(block)).repeat(reps)
    # an N in 0.5 % of the reads: anywhere on a sequence line is fine for the timing; put them into the parsed bases instead
    ms_parse, (bases, offsets) = timed(lambda: ctx.fastx_parse(text))
    n_reads = offsets.numel() - 1
    nd = n_reads // 200
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    rd = torch.randperm(n_reads, device="cuda", generator=g)[:nd]
    pos = offsets[rd] + 5
    bases[pos] = ord("N")
    ms_range, (mn, mx) = timed(lambda: ctx.reads_length_range(offsets))
    if mn == mx:
        ms_scan, s = timed(lambda: ctx.canonical_reduce(bases, n_reads, mx, k, _lib.HASH_LEX, k, 0))
        ms_hist, h = timed(lambda: ctx.histogram(bases, n_reads, mx, k, 1, k, 20))
        path = "uniform kernels"
    else:
        ms_scan, s = timed(lambda: ctx.canonical_reduce(bases, n_reads, mx, k, _lib.HASH_LEX, k, 0, offsets=offsets))
        ms_hist, h = timed(lambda: ctx.histogram(bases, n_reads, mx, k, 1, k, 20, offsets=offsets))
        path = "ragged kernels, bound %d" % mx
    # (round 3) straight from the parser's offsets: kmx_canonical_reduce checks on the device whether the reads are uniform
    ms_gate, sg = timed(lambda: ctx.canonical_reduce(bases, n_reads, mx, k, _lib.HASH_LEX, k, 0, offsets=offsets))
    assert (sg.n_valid, sg.sum_canon, sg.xor_hash) == (s.n_valid, s.sum_canon, s.xor_hash)
    nb = bases.numel()
    print(f"{name}: {text.numel()/1e9:.2f} GB of FASTQ, {n_reads} reads, {nb/1e9:.2f} GB of bases, lengths {mn}..{mx} ({path})")
    print(f"    parse {ms_parse:7.2f} ms ({text.numel()/ms_parse/1e6:5.0f} GB/s of text)   length range {ms_range:5.2f} ms   "
          f"reduce k={k} {ms_scan:6.2f} ms ({nb/ms_scan/1e6:5.0f} GB/s of bases, {s.n_valid/ms_scan/1e6:6.0f} G k-mers/s)   "
          f"histogram 2^20 {ms_hist:6.2f} ms\n"
          f"    reduce called with the offsets array as it is (uniform / ragged decided on the device): {ms_gate:6.2f} ms;  "
          f"parse + that = {ms_parse + ms_gate:6.2f} ms")
    del text, bases, offsets, h
