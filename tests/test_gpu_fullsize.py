"""Parity at BASELINE.json's full sizes.

Two kinds of checks: (1) the WHOLE input against the multi-threaded CPU oracle (SURVEY 8(d) "Parity at scale": the device buffer
is streamed to the host slab by slab, every usable core scans its share of a slab with the oracle -- ctypes releases the GIL --
and the slab summaries are combined: wrapping add / xor / bucket-wise add), one GPU call against it; (2) size-independent
properties: exact window counts, shard linearity (checksum of checksums), exact accounting of injected invalid bytes, the
kernels against each other."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_FULL = int(os.environ.get("KMX_TEST_FULL_READS", 100_000_000))  # BASELINE configs[1]: 1e8 x 150 bp
L = 150
M64 = 2**64 - 1


@pytest.fixture(scope="module")
def ctx():
    from kmers_amd.api import Context

    c = Context()
    yield c
    c.close()


@pytest.fixture(scope="module")
def big(ctx):
    return ctx.gen_reads(N_FULL * L)


def _usable_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def _oracle_over_device_buffer(buf, n_reads, read_len, per_part, combine, slab_reads=6_000_000):
    """the oracle over ALL reads of a device buffer: D2H in slabs of `slab_reads` reads, each slab cut into one part per core,
    `per_part(host_bytes, n)` on a thread per part, the results folded with `combine(acc, part)` (acc starts as None)"""
    from concurrent.futures import ThreadPoolExecutor

    cores = _usable_cores()
    acc = None
    with ThreadPoolExecutor(max_workers=cores) as ex:
        for lo in range(0, n_reads, slab_reads):
            hi = min(n_reads, lo + slab_reads)
            host = buf[lo * read_len:hi * read_len].cpu().numpy()
            m = hi - lo
            step = (m + cores - 1) // cores
            cuts = [(a, min(m, a + step)) for a in range(0, m, step)]
            for part in ex.map(lambda ab: per_part(host[ab[0] * read_len:ab[1] * read_len], ab[1] - ab[0]), cuts):
                acc = part if acc is None else combine(acc, part)
    return acc


@pytest.mark.parametrize("k", [31, 21])
def test_full_input_against_the_oracle(ctx, orc, big, k):
    """BASELINE configs[1] / [2] (k = 31, 21) and the per-GPU work of configs[3] (k = 31 + the LexHasher word hash folded in):
    ONE whole-buffer GPU call == the oracle over all 1e8 reads -- n_valid, the wrapping sum of the canonical words, the xor of
    the Lex hashes and the sum of the forward words (canonical_kmer_iterator.rs:42-70, canonical_kmer.rs:113-119, hash.rs:60-71)"""
    from kmers_amd import _lib

    g = ctx.canonical_reduce(big, N_FULL, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)

    def part(host, n):
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        return (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)

    def comb(a, b):
        return (a[0] + b[0], (a[1] + b[1]) & M64, a[2] ^ b[2], (a[3] + b[3]) & M64)

    o = _oracle_over_device_buffer(big, N_FULL, L, part, comb)
    assert (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw) == o
    # the headline call itself (no hash, no forward sum: what bench.py times) returns the same count and sum
    h = ctx.canonical_reduce(big, N_FULL, L, k)
    assert (h.n_valid, h.sum_canon) == o[:2]


def test_full_input_two_word_k_against_the_oracle(ctx, orc, big):
    """BASELINE configs[2], k = 63 ([u64;2] storage): one GPU call == the oracle over all 1e8 reads (low / high word sums and the
    128-bit xor fold; the order and the hash of two-word k-mers are build-defined)"""
    k = 63
    g = ctx.canonical_reduce2(big, N_FULL, L, k, with_hash=True)

    def part(host, n):
        o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
        return (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)

    def comb(a, b):
        return (a[0] + b[0], (a[1] + b[1]) & M64, (a[2] + b[2]) & M64, a[3] ^ b[3], a[4] ^ b[4])

    o = _oracle_over_device_buffer(big, N_FULL, L, part, comb)
    assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == o


def test_full_input_histogram_against_the_oracle(ctx, orc):
    """BASELINE configs[4] at its own per-GPU size: 1.25e8 reads, k = 31, 2^20 buckets of the Lex hash -- every bucket of ONE GPU
    call == the oracle's histogram of all reads"""
    n = int(os.environ.get("KMX_TEST_HIST_READS", 125_000_000))
    k, b = 31, 20
    buf = ctx.gen_reads(n * L)
    g = ctx.histogram(buf, n, L, k, 1, k, b).cpu().numpy().astype(np.uint64)
    o = _oracle_over_device_buffer(buf, n, L, lambda host, m: orc.histogram(host, m, L, k, k, b).astype(np.uint64),
                                   lambda a, c: a + c, slab_reads=8_000_000)
    assert int(o.sum()) == n * (L - k + 1)
    assert np.array_equal(g, o)


@pytest.mark.parametrize("k", [31, 21])
def test_full_size_counts_linearity_and_prefix(ctx, orc, big, k):
    from kmers_amd import _lib

    whole = ctx.canonical_reduce(big, N_FULL, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    assert whole.n_valid == N_FULL * (L - k + 1)          # clean synthetic input: every window is valid
    # shard linearity: summaries of disjoint read ranges combine (wrapping add / xor) to the whole
    cuts = [0, N_FULL // 3 + 5, N_FULL // 2 + 64 * 7 + 1, N_FULL]
    acc = dict(n=0, s=0, x=0, f=0)
    for a, b in zip(cuts, cuts[1:]):
        part = ctx.canonical_reduce(big[a * L:b * L], b - a, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        acc["n"] += part.n_valid
        acc["s"] = (acc["s"] + part.sum_canon) & M64
        acc["x"] ^= part.xor_hash
        acc["f"] = (acc["f"] + part.sum_fw) & M64
    assert (acc["n"], acc["s"], acc["x"], acc["f"]) == (whole.n_valid, whole.sum_canon, whole.xor_hash, whole.sum_fw)
    # oracle-checked prefix (1e6 reads = 1.2e8 k-mers)
    n_chk = min(N_FULL, 1_000_000)
    host = big[: n_chk * L].cpu().numpy()
    o = orc.canonical_reduce(host, n_chk, L, k, hasher_k=k)
    g = ctx.canonical_reduce(big[: n_chk * L], n_chk, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    assert (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw) == (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)
    # the generic (reference-shaped) kernel and the fast kernel agree on a mid-size slice
    n_mid = min(N_FULL, 3_000_000)
    off = ctx.to_device(np.arange(n_mid + 1, dtype=np.uint64) * np.uint64(L))
    gg = ctx.canonical_reduce(big[: n_mid * L], n_mid, 0, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=off)
    gf = ctx.canonical_reduce(big[: n_mid * L], n_mid, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    assert (gg.n_valid, gg.sum_canon, gg.xor_hash, gg.sum_fw) == (gf.n_valid, gf.sum_canon, gf.xor_hash, gf.sum_fw)


@pytest.mark.parametrize("k", [31, 21])
def test_full_size_half_a_percent_of_dirty_reads(ctx, orc, big, k):
    """5e5 reads with an N (27 % of the tiles hold one): the scan marks them and sweep_flagged_kernel (kmx_sweep.hip) takes the windows
    that hold the N back out.  Window accounting, shard linearity with the dirt in, the ragged kernel on the same bytes, an
    oracle-checked prefix, and the array of masks back to all-zero (a second call gives the same)."""
    import torch
    from kmers_amd import _lib

    rng = np.random.default_rng(k)
    nd = N_FULL // 200
    reads = rng.choice(N_FULL, size=nd, replace=False)
    offs = rng.integers(0, L, size=nd)
    idx = torch.from_numpy((reads.astype(np.int64) * L + offs)).to(big.device)
    saved = big[idx].clone()
    big[idx] = ord("N")
    try:
        whole = ctx.canonical_reduce(big, N_FULL, L, k, _lib.HASH_LEX, k, 0)
        p = offs.astype(np.int64)
        killed = int((np.minimum(p, L - k) - np.maximum(0, p - k + 1) + 1).sum())
        assert whole.n_valid == N_FULL * (L - k + 1) - killed
        again = ctx.canonical_reduce(big, N_FULL, L, k, _lib.HASH_LEX, k, 0)
        assert (again.n_valid, again.sum_canon, again.xor_hash) == (whole.n_valid, whole.sum_canon, whole.xor_hash)
        cuts = [0, N_FULL // 4 + 3, N_FULL // 2 + 64 * 5 + 1, N_FULL]
        n = s = x = 0
        for a, b in zip(cuts, cuts[1:]):
            part = ctx.canonical_reduce(big[a * L:b * L], b - a, L, k, _lib.HASH_LEX, k, 0)
            n += part.n_valid
            s = (s + part.sum_canon) & M64
            x ^= part.xor_hash
        assert (n, s, x) == (whole.n_valid, whole.sum_canon, whole.xor_hash)
        n_mid = min(N_FULL, 20_000_000)   # the ragged kernel (its own blanking path) on the same bytes
        off = ctx.to_device(np.arange(n_mid + 1, dtype=np.uint64) * np.uint64(L))
        gr = ctx.canonical_reduce(big[: n_mid * L], n_mid, L, k, _lib.HASH_LEX, k, 0, offsets=off)
        gu = ctx.canonical_reduce(big[: n_mid * L], n_mid, L, k, _lib.HASH_LEX, k, 0)
        assert (gr.n_valid, gr.sum_canon, gr.xor_hash) == (gu.n_valid, gu.sum_canon, gu.xor_hash)
        n_chk = min(N_FULL, 1_000_000)
        o = orc.canonical_reduce(big[: n_chk * L].cpu().numpy(), n_chk, L, k, hasher_k=k)
        g = ctx.canonical_reduce(big[: n_chk * L], n_chk, L, k, _lib.HASH_LEX, k, 0)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)
    finally:
        big[idx] = saved


@pytest.mark.parametrize("k", [31, 63])
def test_full_input_two_percent_dirty_against_the_oracle(ctx, orc, big, k):
    """Round 6 (the full-size dirty test checked the oracle on a prefix only): 2e6 of the 1e8 reads get invalid bytes -- one
    N, two bytes, a run, both ends, the whole read; 73 % of the tiles hold one -- and ONE whole-buffer GPU call equals the oracle over
    ALL reads: the scan's marks and the sweep's subtraction at full size, single- and two-word k
    (canonical_kmer_iterator.rs:50-66)."""
    import torch
    from kmers_amd import _lib

    rng = np.random.default_rng(600 + k)
    nd = N_FULL // 50
    reads = np.sort(rng.choice(N_FULL, size=nd, replace=False)).astype(np.int64)
    kind = np.arange(nd) % 5
    pos = rng.integers(0, L, size=nd)
    idx = [reads * L + pos]                                                   # every dirty read: one byte
    two = kind == 1
    idx.append(reads[two] * L + rng.integers(0, L, size=int(two.sum())))       # a second one
    run = np.nonzero(kind == 2)[0]
    for j in range(1, 24):                                                     # a run of up to 24
        sel = run[pos[run] + j < L]
        idx.append(reads[sel] * L + pos[sel] + j)
    ends = kind == 3
    idx.append(reads[ends] * L)
    idx.append(reads[ends] * L + (L - 1))
    whole = np.nonzero(kind == 4)[0][:20000]
    idx.append((reads[whole][:, None] * L + np.arange(L)[None, :]).ravel())
    idx = torch.from_numpy(np.unique(np.concatenate(idx))).to(big.device)
    saved = big[idx].clone()
    big[idx] = ord("N")
    try:
        if k <= 31:
            g = ctx.canonical_reduce(big, N_FULL, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
            got = (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw)

            def part(host, n):
                o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
                return (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)

            def comb(a, b):
                return (a[0] + b[0], (a[1] + b[1]) & M64, a[2] ^ b[2], (a[3] + b[3]) & M64)
        else:
            g = ctx.canonical_reduce2(big, N_FULL, L, k, with_hash=True)
            got = (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi)

            def part(host, n):
                o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
                return (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)

            def comb(a, b):
                return (a[0] + b[0], (a[1] + b[1]) & M64, (a[2] + b[2]) & M64, a[3] ^ b[3], a[4] ^ b[4])

        assert got == _oracle_over_device_buffer(big, N_FULL, L, part, comb)
    finally:
        big[idx] = saved


def test_full_size_invalid_byte_accounting(ctx, big):
    """an 'N' at read offset p kills exactly min(p, L-k) - max(0, p-k+1) + 1 windows of that read"""
    import torch

    k = 31
    clean = ctx.canonical_reduce(big, N_FULL, L, k)
    rng = np.random.default_rng(5)
    reads = rng.choice(N_FULL, size=1000, replace=False)
    offs = rng.integers(0, L, size=1000)
    idx = torch.from_numpy((reads.astype(np.int64) * L + offs)).to(big.device)
    saved = big[idx].clone()
    big[idx] = ord("N")
    try:
        dirty = ctx.canonical_reduce(big, N_FULL, L, k)
    finally:
        big[idx] = saved
    killed = sum(min(int(p), L - k) - max(0, int(p) - k + 1) + 1 for p in offs)
    assert dirty.n_valid == clean.n_valid - killed
    again = ctx.canonical_reduce(big, N_FULL, L, k)
    assert (again.n_valid, again.sum_canon) == (clean.n_valid, clean.sum_canon)  # restored, idempotent


@pytest.mark.parametrize("k", [63, 64, 47])
def test_full_size_two_word_k(ctx, orc, big, k):
    """BASELINE configs[2]: k=63 ([u64;2] storage, build-defined order) at the full 1e8 reads -- exact count, shard
    linearity (the word sums and the xor fold of two halves combine to the whole), oracle-checked prefix; and an even k
    and a 3-waves/SIMD k on the same terms"""
    n = N_FULL
    g = ctx.canonical_reduce2(big[: n * L], n, L, k, with_hash=True)
    assert g.n_valid == n * (L - k + 1)
    h = n // 2 + 37
    a = ctx.canonical_reduce2(big[: h * L], h, L, k, with_hash=True)
    b = ctx.canonical_reduce2(big[h * L: n * L], n - h, L, k, with_hash=True)
    M = (1 << 64) - 1   # (kmx_summary2: the low and the high words are summed separately, each wrapping)
    assert (g.n_valid, g.sum_lo, g.sum_hi) == (a.n_valid + b.n_valid, (a.sum_lo + b.sum_lo) & M, (a.sum_hi + b.sum_hi) & M)
    assert (g.xor_lo, g.xor_hi) == (a.xor_lo ^ b.xor_lo, a.xor_hi ^ b.xor_hi)
    n_chk = min(n, 300_000)
    o = orc.canonical_reduce2(big[: n_chk * L].cpu().numpy(), n_chk, L, k, with_hash=True)
    gp = ctx.canonical_reduce2(big[: n_chk * L], n_chk, L, k, with_hash=True)
    assert tuple(getattr(gp, f) for f, _ in gp._fields_) == tuple(getattr(o, f) for f, _ in o._fields_)


def test_full_size_ragged_and_long_reads_agree_with_the_uniform_scan(ctx, big):
    """the same 15 GB three ways: 1e8 uniform reads; the same reads behind an offsets array (ragged bit-sliced kernel);
    and, re-cut as 1.5e6 reads of 10 000 bases (overlapping segments on the ragged kernel), the windows that do not
    straddle an old read boundary -- checked through exact counts and the identity n_valid(long) = n_valid + (k-1)(n - n_long)"""
    from kmers_amd import _lib

    n, k = N_FULL, 31
    a = ctx.canonical_reduce(big, n, L, k, _lib.HASH_LEX, k, 0)
    off = ctx.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    r = ctx.canonical_reduce(big, n, 160, k, _lib.HASH_LEX, k, 0, offsets=off)
    assert (r.n_valid, r.sum_canon, r.xor_hash) == (a.n_valid, a.sum_canon, a.xor_hash)
    del off
    Ll = 10_000
    nl = n * L // Ll
    g = ctx.canonical_reduce(big[: nl * Ll], nl, Ll, k, _lib.HASH_NONE, 0, 0)
    assert g.n_valid == nl * (Ll - k + 1)
    h = nl // 3
    p, q = ctx.canonical_reduce(big[: h * Ll], h, Ll, k), ctx.canonical_reduce(big[h * Ll: nl * Ll], nl - h, Ll, k)
    assert (g.n_valid, g.sum_canon) == (p.n_valid + q.n_valid, (p.sum_canon + q.sum_canon) & ((1 << 64) - 1))


@pytest.mark.parametrize("k", [31, 27, 17])
def test_full_size_three_paths_agree(ctx, big, k):
    """ASCII bit-sliced kernel, ragged word-domain kernel (same reads through an offsets array) and the 2-bit packed
    bit-sliced kernel (k = 31; lane-per-read kernel otherwise) produce the same summary on the first 2e7 reads"""
    from kmers_amd import _lib

    n = min(N_FULL, 20_000_000)
    view = big[: n * L]
    a = ctx.canonical_reduce(view, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    off = ctx.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    r = ctx.canonical_reduce(view, n, 160, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=off)
    assert (r.n_valid, r.sum_canon, r.xor_hash, r.sum_fw) == (a.n_valid, a.sum_canon, a.xor_hash, a.sum_fw)
    words = ctx.seqvec_from_bytes(view)
    p = ctx.seqvec_canonical_reduce(words, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    assert (p.n_valid, p.sum_canon, p.xor_hash, p.sum_fw) == (a.n_valid, a.sum_canon, a.xor_hash, a.sum_fw)
    assert ctx.seqvec_to_bytes(words, 4096).cpu().numpy().tobytes() == view[:4096].cpu().numpy().tobytes()


def test_full_size_ragged_histogram_matches_uniform(ctx, big):
    """2^20 buckets: the partitioned histogram through the offsets layout (batched and per-window slot paths) counts what the
    uniform layout counts, bucket by bucket"""
    import torch

    n = min(N_FULL, 30_000_000)
    k, b = 31, 20
    off = ctx.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    hu = ctx.histogram(big[: n * L], n, L, k, 1, k, b)
    hr = ctx.histogram(big[: n * L], n, L, k, 1, k, b, offsets=off)
    hr0 = ctx.histogram(big[: n * L], n, 0, k, 1, k, b, offsets=off)
    assert int(hu.sum().item()) == n * (L - k + 1)
    assert torch.equal(hu, hr) and torch.equal(hu, hr0)


@pytest.mark.parametrize("b", [12, 20, 22])
def test_full_size_histogram_totals_and_linearity(ctx, big, b):
    """the LDS-table and the partitioned histogram count every k-mer exactly once: totals, and two halves add up to the whole"""
    import torch
    from kmers_amd import _lib

    n, k = min(N_FULL, 30_000_000), 31
    whole = ctx.histogram(big[: n * L], n, L, k, _lib.HASH_LEX, k, b)
    assert int(whole.sum().item()) == n * (L - k + 1)
    h = n // 2 + 33
    parts = ctx.histogram(big[: h * L], h, L, k, _lib.HASH_LEX, k, b)
    parts = ctx.histogram(big[h * L: n * L], n - h, L, k, _lib.HASH_LEX, k, b, counts=parts)
    assert torch.equal(parts, whole)


@pytest.mark.parametrize("k", [31, 21, 27])
def test_full_size_ragged_offsets_past_4gib(ctx, big, k):
    """the same 15 GB behind an offsets array (offsets run past 2^31 and 2^32: the tile bounds are 64-bit end to end)
    give the uniform-layout summary -- bit-sliced ragged kernel (k = 31, 21) and word-domain ragged kernel (k = 27)"""
    import torch

    from kmers_amd import _lib

    uni = ctx.canonical_reduce(big, N_FULL, L, k, _lib.HASH_LEX, k)
    offsets = torch.arange(N_FULL + 1, dtype=torch.int64, device=big.device) * L
    for hint in (160, 0):
        rag = ctx.canonical_reduce(big, N_FULL, hint, k, _lib.HASH_LEX, k, 0, offsets=offsets)
        assert (rag.n_valid, rag.sum_canon, rag.xor_hash) == (uni.n_valid, uni.sum_canon, uni.xor_hash)
    # reads of two lengths: dropping the last 30 bases of every other read = the summary of the two strided halves
    lens = torch.full((N_FULL,), L, dtype=torch.int64, device=big.device)
    lens[1::2] = L - 30
    # (a ragged batch is contiguous: compare on a compacted copy of a prefix, built pair by pair -- 150 + 120 bases)
    n = min(N_FULL, 20_000_000) & ~1
    view = big[:n * L].view(n, L)
    packed = torch.cat([view[0::2], view[1::2, :L - 30]], dim=1).contiguous().view(-1)
    off = torch.zeros(n + 1, dtype=torch.int64, device=big.device)
    off[1:] = torch.cumsum(lens[:n], 0)
    rag = ctx.canonical_reduce(packed, n, 160, k, _lib.HASH_LEX, k, 0, offsets=off)
    a = ctx.canonical_reduce(view[0::2].contiguous().view(-1), (n + 1) // 2, L, k, _lib.HASH_LEX, k)
    b = ctx.canonical_reduce(view[1::2, :L - 30].contiguous().view(-1), n // 2, L - 30, k, _lib.HASH_LEX, k)
    assert rag.n_valid == a.n_valid + b.n_valid
    assert rag.sum_canon == (a.sum_canon + b.sum_canon) & M64
    assert rag.xor_hash == a.xor_hash ^ b.xor_hash


def test_full_size_fastq_image_roundtrip(ctx):
    """a multi-GB FASTQ image built on the device from the synthetic reads: kmx_fastx_parse must give the reads back
    (size-independent property: format -> parse is the identity on bases and offsets)"""
    import torch

    n = min(N_FULL, 16_000_000)
    bases = ctx.gen_reads(n * L).view(n, L)
    hdr = torch.tensor(list(b"@r\n"), dtype=torch.uint8, device=bases.device)
    mid = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=bases.device)
    rec = torch.empty((n, 3 + L + 3 + L + 1), dtype=torch.uint8, device=bases.device)   # "@r\n" seq "\n+\n" qual "\n"
    rec[:, 0:3] = hdr
    rec[:, 3:3 + L] = bases
    rec[:, 3 + L:6 + L] = mid
    rec[:, 6 + L:6 + 2 * L] = ord("@")       # a quality string made of header characters
    rec[:, 6 + 2 * L] = ord("\n")
    text = rec.view(-1)
    assert text.numel() > 2**32
    out, offsets = ctx.fastx_parse(text)
    assert offsets.numel() == n + 1
    assert torch.equal(offsets, torch.arange(n + 1, dtype=torch.int64, device=bases.device) * L)
    assert torch.equal(out, bases.view(-1))


def test_full_size_chunked_histogram_keeps_its_work_buffer(ctx, big):
    """1e8 x 150 bp at k = 13 need more id space than an eighth of the device memory: the reads go in chunks.  The context's work
    buffer must be allocated once and then stay -- a request a little above the budget used to raise the next call's budget, and
    every call re-allocated 36 GB (1.1 s of host time per call; round 3)"""
    from kmers_amd import _lib

    k, b = 13, 20
    n0 = ctx.work_buffer_info()[1]
    first = ctx.histogram(big, N_FULL, L, k, _lib.HASH_LEX, k, b)
    held, n1 = ctx.work_buffer_info()
    n_first = n1 - n0
    again = None
    for _ in range(3):
        again = ctx.histogram(big, N_FULL, L, k, _lib.HASH_LEX, k, b)
    n_later = ctx.work_buffer_info()[1] - n1
    assert held > 0
    assert n_first <= 1 and n_later == 0
    assert int(first.sum().item()) == N_FULL * (L - k + 1)
    import torch

    assert torch.equal(first, again)


@pytest.mark.parametrize("k,two_word", [(31, False), (21, False), (47, True), (63, True)])
def test_full_size_identical_reads_through_the_accumulator_folds(ctx, orc, k, two_word):
    """6e7 copies of ONE read (round 4): every tile adds the same 0/1 pattern from all 64 reads to the fp32 accumulators of
    pass 2 -- the largest per-entry counts there are -- and every wave runs ~300 tiles, i.e. across at least one fold into the
    64-bit class sums (every 256 tiles).  The summary is the one read's, times the number of reads (wrapping); xor folds of an
    even number of equal hashes cancel."""
    import torch
    from kmers_amd import _lib

    n = min(N_FULL, 60_000_000)
    rng = np.random.default_rng(k)
    one = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
    d_one = ctx.to_device(one)
    bases = d_one.repeat(n)
    torch.cuda.synchronize()
    if not two_word:
        o = orc.canonical_reduce(one, 1, L, k, hasher_k=k)
        g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        assert g.n_valid == n * o.n_valid
        assert g.sum_canon == (n * o.sum_canon) & M64
        assert g.sum_fw == (n * o.sum_fw) & M64
        assert g.xor_hash == (o.xor_hash if n % 2 else 0)
    else:
        o = orc.canonical_reduce2(one, 1, L, k, with_hash=True)
        g = ctx.canonical_reduce2(bases, n, L, k, with_hash=True)
        assert g.n_valid == n * o.n_valid
        tot = n * ((o.sum_hi << 64) | o.sum_lo)      # the two sums wrap independently (kmx_summary2: per-word sums)
        assert g.sum_lo == (n * o.sum_lo) & M64 and g.sum_hi == (n * o.sum_hi) & M64, tot
        assert (g.xor_lo, g.xor_hi) == ((o.xor_lo, o.xor_hi) if n % 2 else (0, 0))
    del bases
    torch.cuda.empty_cache()


def test_full_size_long_ragged_reads(ctx, orc):
    """6 G bases of 1 000..20 000-base reads behind an offsets array, length bound above 256: the segment path of round 4
    (kmx_segments.hip).  Exact window count, shard linearity over three cuts of the batch, an oracle-checked prefix, and the
    same summary from the per-read path on a slice."""
    from kmers_amd import _lib

    k = 31
    rng = np.random.default_rng(77)
    n = 570_000
    lens = rng.integers(1000, 20001, n)
    lens[rng.integers(0, n, 500)] = rng.integers(0, 40, 500)     # a few empty / shorter-than-k reads between them
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(offsets[-1])
    bases = ctx.gen_reads(total)
    d_off = ctx.to_device(offsets)
    whole = ctx.canonical_reduce(bases, n, 1 << 20, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=d_off)
    assert whole.n_valid == int(np.maximum(lens - k + 1, 0).sum())
    cuts = [0, n // 3 + 1, n // 2 + 77, n]
    acc = dict(n=0, s=0, x=0, f=0)
    for a, b in zip(cuts, cuts[1:]):
        part = ctx.canonical_reduce(bases, b - a, 1 << 20, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=d_off[a:b + 1])
        acc["n"] += part.n_valid
        acc["s"] = (acc["s"] + part.sum_canon) & M64
        acc["x"] ^= part.xor_hash
        acc["f"] = (acc["f"] + part.sum_fw) & M64
    assert (acc["n"], acc["s"], acc["x"], acc["f"]) == (whole.n_valid, whole.sum_canon, whole.xor_hash, whole.sum_fw)
    n_chk = 15_000
    host = bases[: int(offsets[n_chk])].cpu().numpy()
    o = orc.canonical_reduce(host, n_chk, 0, k, hasher_k=k, offsets=offsets[: n_chk + 1])
    g = ctx.canonical_reduce(bases, n_chk, 1 << 20, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=d_off[: n_chk + 1])
    assert (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw) == (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)
    p = ctx.canonical_reduce(bases, n_chk, 0, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=d_off[: n_chk + 1])   # no bound: the per-read path
    assert (p.n_valid, p.sum_canon, p.xor_hash, p.sum_fw) == (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)


@pytest.mark.parametrize("k", [63, 33])
def test_full_size_two_word_k_behind_offsets(ctx, big, k):
    """round 4, two-word k on the full 1e8 reads behind an offsets array, three ways that must agree with the uniform scan: bound
    150 (the device-side gate picks the uniform kernel), bound 160 (the two-word ragged kernel scans them as ragged reads), bound
    250 on the same bytes re-cut as 6e7 reads of 250 bases (segments of 161 - k windows) against the uniform scan of those"""
    n = N_FULL
    a = ctx.canonical_reduce2(big, n, L, k, with_hash=True)
    off = ctx.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    for bound in (150, 160):
        r = ctx.canonical_reduce2(big, n, bound, k, with_hash=True, offsets=off)
        assert tuple(getattr(r, f) for f, _ in r._fields_) == tuple(getattr(a, f) for f, _ in a._fields_), bound
    del off
    L2 = 250
    n2 = n * L // L2
    u = ctx.canonical_reduce2(big[: n2 * L2], n2, L2, k, with_hash=True)
    assert u.n_valid == n2 * (L2 - k + 1)
    lens = np.full(n2, L2, dtype=np.uint64)
    lens[5] -= 3          # (one read shorter: the batch's total no longer says "untrimmed", the reads are cut into segments)
    off2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g = ctx.canonical_reduce2(big[: int(off2[-1])], n2, L2, k, with_hash=True, offsets=ctx.to_device(off2))
    # the two batches differ in read 5 and in where every later read starts (3 bases earlier): only the counts are comparable
    assert g.n_valid == n2 * (L2 - k + 1) - 3


def test_full_size_materialised_words_add_up_to_the_scan(ctx, big):
    """round 4: the canonical words that kmx_canonical_windows writes -- uniform 150-base reads (the line-aligned kernel), the same
    bytes as reads of 1 000 bases (segments planned on the device), and as ragged reads behind offsets (the ragged ring) -- must
    add up (mod 2^64) to the sum_canon of kmx_canonical_reduce over the same reads: two independent kernels, one checksum"""
    import torch

    k = 31
    M = (1 << 64) - 1
    for Lr, n, ragged in ((150, 20_000_000, False), (1000, 3_000_000, False), (150, 20_000_000, True)):
        bases = big[: n * Lr]
        s = ctx.canonical_reduce(bases, n, Lr, k)
        if ragged:
            off = np.arange(n + 1, dtype=np.uint64) * np.uint64(Lr)
            w = ctx.canonical_windows(bases, n, 160, k, offsets=ctx.to_device(off), host_offsets=off, want=("canon",))["canon"]
        else:
            w = ctx.canonical_windows(bases, n, Lr, k, want=("canon",))["canon"]
        assert w.numel() == s.n_valid
        # (int64 wrapping sums: the same bits as the unsigned sum)
        total = int(w.sum(dtype=torch.int64).item()) & M
        assert total == s.sum_canon, (Lr, ragged)
        del w
