"""Round-6 GPU parity tests (through the C ABI, bit-exact against the CPU oracle).

  * sweep_flagged_kernel (kmx_sweep.hip): the bit-sliced scan no longer blanks the reads that hold an invalid byte -- it scans
    the tile as it is and marks them; the sweep gathers the marked reads 64 at a time, finds the windows that hold an invalid
    byte and takes them back OUT of the sums.  Every instantiation is driven here: both frames (reads up to 160 / up to 256
    bases), every k-mer width (k = 13..16, 17..31, 33..48, 49..64: one to four dwords per window), uniform reads, reads behind
    offsets (ragged, and uniform ones through the device-side gate), segments of long uniform and of long ragged reads; invalid
    bytes at both ends, in runs, several per read, whole reads of N, and in the batch's very last read (whose 16-byte loads
    must not leave the buffer).  The semantics at stake are the iterator's skip rule, canonical_kmer_iterator.rs:50-66:
    exactly the windows that hold no invalid byte are yielded.
  * kmx_minimizers: SeqVecMinimizerIter (seq_vector/minimizers.rs:39-141) over READS -- ASCII, uniform or behind offsets --
    against the oracle's monotone deque on SeqVector::from(read), and the reference's own vectors (minimizers.rs:237-290).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from kmers_amd.api import Context

    c = Context()
    yield c
    c.close()


def _dirty(host, n_reads, starts, lens, rng, share, kinds=("one", "two", "run", "ends", "all")):
    """a share of the reads gets invalid bytes: one, two, a run, both ends, the whole read"""
    reads = np.nonzero(rng.random(n_reads) < share)[0]
    for i, r in enumerate(reads):
        s, ln = int(starts[r]), int(lens[r])
        if ln == 0:
            continue
        kind = kinds[i % len(kinds)]
        if kind == "one":
            host[s + int(rng.integers(0, ln))] = ord("N")
        elif kind == "two":
            host[s + int(rng.integers(0, ln))] = ord("n")
            host[s + int(rng.integers(0, ln))] = ord("-")
        elif kind == "run":
            a = int(rng.integers(0, ln))
            host[s + a : s + min(ln, a + int(rng.integers(2, 40)))] = ord("N")
        elif kind == "ends":
            host[s] = 0
            host[s + ln - 1] = 255
        else:
            host[s : s + ln] = ord("N")
    return reads


def _check1(g, o, with_fw=True):
    assert g.n_valid == o.n_valid
    assert g.sum_canon == o.sum_canon
    assert g.xor_hash == o.xor_hash
    if with_fw:
        assert g.sum_fw == o.sum_fw


def _check2(g, o):
    assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


# (k, L): V1 = 0 / 1 on the 10-word frame, on the 16-word frame; L odd, L a multiple of 16, L at the frame's limit
UNIFORM1 = [(13, 150), (16, 100), (17, 150), (21, 151), (31, 150), (31, 160), (31, 36), (27, 112),
            (13, 200), (16, 256), (17, 161), (31, 250), (31, 256), (24, 208)]


@pytest.mark.parametrize("k,L", UNIFORM1)
@pytest.mark.parametrize("share", [0.02, 0.3])
def test_sweep_uniform(ctx, orc, k, L, share):
    from kmers_amd import _lib

    n = 64 * 300 + 17
    rng = np.random.default_rng(1000 * k + L)
    host = ctx.gen_reads(n * L, first_byte=k * L).cpu().numpy().copy()
    starts = np.arange(n, dtype=np.int64) * L
    _dirty(host, n, starts, np.full(n, L), rng, share)
    host[(n - 40) * L + 3] = ord("N")            # a read of the last full tile ...
    host[(64 * 300 - 1) * L + L - 1] = ord("N")  # ... and its last byte
    o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
    g = ctx.canonical_reduce(ctx.to_device(host), n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _check1(g, o)


@pytest.mark.parametrize("k,L", [(31, 150), (21, 100), (13, 250)])
def test_sweep_last_read_of_the_buffer(ctx, orc, k, L):
    """n a multiple of 64: the batch's last read sits in a full tile, is dirty, and ends with the allocation -- its last 16-byte
    load would run past it.  The device buffer is exactly n * L bytes, from an offset that is not 16-byte aligned too."""
    import torch
    from kmers_amd import _lib

    n = 64 * 8
    rng = np.random.default_rng(k + L)
    host = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * L)].copy()
    host[(n - 1) * L + L // 2] = ord("N")
    host[(n - 1) * L + L - 1] = ord("N")
    host[(n - 2) * L] = ord("N")
    o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
    g = ctx.canonical_reduce(ctx.to_device(host), n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _check1(g, o)
    for lead in (1, 7):
        big = torch.empty(n * L + lead, dtype=torch.uint8, device="cuda")
        big[lead:] = torch.from_numpy(host).cuda()
        g = ctx.canonical_reduce(big[lead:], n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _check1(g, o)


# two-word k: V1 = 2 (33..48) and 3 (49..64) on both frames
UNIFORM2 = [(33, 150), (40, 150), (48, 160), (49, 150), (63, 150), (64, 150), (33, 250), (47, 208), (50, 256), (63, 250), (64, 200)]


@pytest.mark.parametrize("k,L", UNIFORM2)
def test_sweep_uniform_two_word(ctx, orc, k, L):
    n = 64 * 200 + 5
    rng = np.random.default_rng(1000 * k + L)
    host = ctx.gen_reads(n * L, first_byte=k * L).cpu().numpy().copy()
    _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, 0.1)
    host[(64 * 200 - 1) * L + L - 1] = ord("N")
    o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
    g = ctx.canonical_reduce2(ctx.to_device(host), n, L, k, with_hash=True)
    _check2(g, o)


def _ragged(ctx, n, lo, hi, seed, k, share):
    rng = np.random.default_rng(seed)
    lens = np.where(rng.random(n) < 0.7, hi, rng.integers(lo, hi + 1, n)).astype(np.int64)
    lens[rng.random(n) < 0.01] = rng.integers(0, k, 1)[0]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = ctx.gen_reads(int(offs[-1]), first_byte=seed).cpu().numpy().copy()
    _dirty(host, n, offs[:-1].astype(np.int64), lens, rng, share)
    # the batch's last read, if it sits in a full tile
    if lens[-1] > 0:
        host[int(offs[-1]) - 1] = ord("N")
    return host, offs


@pytest.mark.parametrize("k,bound,lo,hi", [(31, 150, 36, 150), (13, 160, 100, 160), (21, 100, 30, 100), (31, 250, 100, 250), (16, 256, 200, 256), (31, 0, 100, 160)])
def test_sweep_ragged(ctx, orc, k, bound, lo, hi):
    from kmers_amd import _lib

    n = 64 * 256
    host, offs = _ragged(ctx, n, lo, hi, 77 * k + bound, k, 0.1)
    o = orc.canonical_reduce(host, n, bound, k, hasher_k=k, offsets=offs)
    g = ctx.canonical_reduce(ctx.to_device(host), n, bound, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=ctx.to_device(offs))
    _check1(g, o)


@pytest.mark.parametrize("k,bound,lo,hi", [(33, 150, 36, 150), (47, 160, 100, 160), (55, 150, 60, 150), (64, 160, 64, 160), (63, 250, 100, 250)])
def test_sweep_ragged_two_word(ctx, orc, k, bound, lo, hi):
    n = 64 * 256
    host, offs = _ragged(ctx, n, lo, hi, 77 * k + bound, k, 0.1)
    o = orc.canonical_reduce2(host, n, bound, k, with_hash=True, offsets=offs)
    g = ctx.canonical_reduce2(ctx.to_device(host), n, bound, k, with_hash=True, offsets=ctx.to_device(offs))
    _check2(g, o)


@pytest.mark.parametrize("k,L", [(31, 150), (21, 250), (63, 150), (40, 200)])
def test_sweep_uniform_behind_offsets(ctx, orc, k, L):
    """reads of one length behind an offsets array (untrimmed FASTQ): the device-side gate hands them to the uniform kernels, whose
    blanked reads the sweep then finds at the length the gate left"""
    from kmers_amd import _lib

    n = 64 * 128
    rng = np.random.default_rng(k * L)
    host = ctx.gen_reads(n * L, first_byte=k).cpu().numpy().copy()
    _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, 0.2)
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    bound = 160 if L <= 160 else 256
    if k <= 31:
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        g = ctx.canonical_reduce(ctx.to_device(host), n, bound, k, _lib.HASH_LEX, k, 0, offsets=ctx.to_device(offs))
        _check1(g, o, with_fw=False)
    else:
        o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
        g = ctx.canonical_reduce2(ctx.to_device(host), n, bound, k, with_hash=True, offsets=ctx.to_device(offs))
        _check2(g, o)


@pytest.mark.parametrize("k,L", [(31, 300), (31, 1000), (13, 257), (21, 5000), (33, 300), (49, 1000), (63, 300), (64, 1000), (50, 400)])
def test_sweep_segments_of_long_uniform_reads(ctx, orc, k, L):
    from kmers_amd import _lib

    n = max(64 * 40, (64 * 600 * 150) // L)
    rng = np.random.default_rng(k + L)
    host = ctx.gen_reads(n * L, first_byte=L).cpu().numpy().copy()
    _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, 0.3, kinds=("one", "two", "run", "ends"))
    host[n * L - 1] = ord("N")
    if k <= 31:
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        g = ctx.canonical_reduce(ctx.to_device(host), n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _check1(g, o)
    else:
        o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
        g = ctx.canonical_reduce2(ctx.to_device(host), n, L, k, with_hash=True)
        _check2(g, o)


@pytest.mark.parametrize("k", [31, 21, 63, 40])
def test_sweep_segments_of_long_ragged_reads(ctx, orc, k):
    from kmers_amd import _lib

    n = 4000
    rng = np.random.default_rng(k)
    lens = rng.integers(200, 3000, n).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = ctx.gen_reads(int(offs[-1]), first_byte=k).cpu().numpy().copy()
    _dirty(host, n, offs[:-1].astype(np.int64), lens, rng, 0.5, kinds=("one", "two", "run", "ends"))
    host[int(offs[-1]) - 1] = ord("N")
    if k <= 31:
        o = orc.canonical_reduce(host, n, 3000, k, hasher_k=k, offsets=offs)
        g = ctx.canonical_reduce(ctx.to_device(host), n, 3000, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=ctx.to_device(offs))
        _check1(g, o)
    else:
        o = orc.canonical_reduce2(host, n, 3000, k, with_hash=True, offsets=offs)
        g = ctx.canonical_reduce2(ctx.to_device(host), n, 3000, k, with_hash=True, offsets=ctx.to_device(offs))
        _check2(g, o)


@pytest.mark.parametrize("k,L", [(31, 150), (63, 150)])
def test_sweep_at_size_every_read_dirty(ctx, orc, k, L):
    """1e6 reads, every one of them dirty: the sweep is the whole scan (every wave sweeps many times, the masks of every tile
    are consumed and cleared) -- twice in a row on one context, the second call finding the masks as the first left them"""
    from kmers_amd import _lib

    n = 1_000_000
    rng = np.random.default_rng(k)
    host = ctx.gen_reads(n * L, first_byte=99).cpu().numpy().copy()
    host[np.arange(n, dtype=np.int64) * L + rng.integers(0, L, n)] = ord("N")
    dev = ctx.to_device(host)
    for _ in range(2):
        if k <= 31:
            o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
            g = ctx.canonical_reduce(dev, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
            _check1(g, o)
        else:
            o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
            g = ctx.canonical_reduce2(dev, n, L, k, with_hash=True)
            _check2(g, o)


# ---------------------------------------------------------------- kmx_minimizers: SeqVecMinimizerIter over READS (seq_vector/minimizers.rs:39-141)
def _acgt(rng, n):
    return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()


def _mm_oracle_uniform(orc, host, n, L, k, w, hk):
    return orc.seqvec_minimizers(orc.SeqVector(host.tobytes()), n, L, k, w, hk)


def _mm_oracle_ragged(orc, host, offs, k, w, hk):
    """the reference's iterator over SeqVector::from(read) of every read with at least k bases, one after the other"""
    words, pos = [], []
    for r in range(len(offs) - 1):
        rd = host[int(offs[r]): int(offs[r + 1])]
        if len(rd) < k:
            continue
        for a, b in orc.seqvec_iter_minimizers(orc.SeqVector(rd.tobytes()), k, w, hk):
            words.append(a)
            pos.append(b)
    return np.array(words, np.uint64), np.array(pos, np.uint32)


def test_minimizers_reads_kats(ctx, kats):
    """the reference's own minimizer iterator vectors (minimizers.rs:237-290), the read handed over as ASCII"""
    from kmers_amd import _lib

    for t in kats["minimizers"]["iter"]:
        s = np.frombuffer(t["seq"].encode(), np.uint8)
        hasher = _lib.HASH_LEX if t["hasher_k"] else _lib.HASH_IDENTITY
        mw, mp = ctx.minimizers(ctx.to_device(s), 1, len(s), t["k"], t["w"], hasher, t["hasher_k"])
        got = [[int(a), int(b)] for a, b in zip(mw.cpu().numpy().view(np.uint64), mp.cpu().numpy())]
        assert got == t["expect"], t["name"]


@pytest.mark.parametrize("L,k,w,hk", [(150, 31, 15, 15), (150, 21, 11, 0), (100, 9, 3, 32), (151, 31, 28, 28), (256, 31, 15, 15), (36, 31, 15, 6),
                                      (150, 31, 29, 29), (150, 32, 16, 16), (31, 31, 15, 15), (257, 31, 15, 15), (1000, 31, 15, 15), (700, 21, 11, 0),
                                      (5000, 40, 7, 7)])
def test_minimizers_uniform_reads(ctx, orc, L, k, w, hk):
    """ASCII reads of one length: up to 256 bases the sliding-minimum kernel on the reads, above it on pieces of 256 bases;
    hashes above 56 bits the lane-per-k-mer kernel; from an odd address too"""
    import torch
    from kmers_amd import _lib

    rng = np.random.default_rng(L * 1000 + k * 10 + w)
    n = max(16 * 5 + 3, 40000 // L)
    host = _acgt(rng, n * L)
    host[: 8 * L] = np.frombuffer(b"ACAC", np.uint8)[rng.integers(0, 2, 8 * L) * 2]   # two-letter reads: ties everywhere, the leftmost wins
    host[8 * L: 10 * L] = ord("a")                                                    # lower case is a base too
    ow, op = _mm_oracle_uniform(orc, host, n, L, k, w, hk)
    hasher = _lib.HASH_LEX if hk else _lib.HASH_IDENTITY
    for lead in (0, 3):
        big = torch.empty(n * L + lead, dtype=torch.uint8, device="cuda")
        big[lead:] = torch.from_numpy(host).cuda()
        mw, mp = ctx.minimizers(big[lead:], n, L, k, w, hasher, hk)
        assert (mw.cpu().numpy().view(np.uint64) == ow).all()
        assert (mp.cpu().numpy().view(np.uint32) == op).all()


@pytest.mark.parametrize("k,w,hk,lo,hi", [(31, 15, 15, 20, 150), (21, 11, 0, 21, 100), (31, 15, 6, 100, 256), (15, 5, 5, 0, 60), (31, 15, 15, 100, 400)])
def test_minimizers_ragged_reads(ctx, orc, k, w, hk, lo, hi):
    """reads behind offsets (what kmx_fastx_parse hands over): reads shorter than k own no slot; a batch whose longest read is above
    256 bases takes the lane-per-k-mer kernel"""
    from kmers_amd import _lib

    rng = np.random.default_rng(k * 1000 + w + hi)
    n = 16 * 20 + 7
    lens = rng.integers(lo, hi + 1, n).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    wins = np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))]).astype(np.uint64)
    host = _acgt(rng, int(offs[-1]))
    ow, op = _mm_oracle_ragged(orc, host, offs, k, w, hk)
    hasher = _lib.HASH_LEX if hk else _lib.HASH_IDENTITY
    mw, mp = ctx.minimizers(ctx.to_device(host), n, hi, k, w, hasher, hk, offsets=ctx.to_device(offs), win_offsets=ctx.to_device(wins))
    assert len(ow) == int(wins[-1])
    assert (mw.cpu().numpy().view(np.uint64) == ow).all()
    assert (mp.cpu().numpy().view(np.uint32) == op).all()


def test_minimizers_invalid_byte_is_reported(ctx):
    """SeqVector::from panics on a byte outside ACGTacgt (seq_vector.rs:230-242): the call names the first such read"""
    from kmers_amd import _lib
    from kmers_amd.api import KmxError

    rng = np.random.default_rng(5)
    n, L = 100, 150
    host = _acgt(rng, n * L)
    host[57 * L + 149] = ord("N")
    host[80 * L] = ord("N")
    with pytest.raises(KmxError) as ei:
        ctx.minimizers(ctx.to_device(host), n, L, 31, 15, _lib.HASH_LEX, 15)
    assert ei.value.status == _lib.E_INVALID_BASE
    mw, mp = ctx.minimizers(ctx.to_device(host), n, L, 31, 15, _lib.HASH_LEX, 15, check=False)     # asked not to look: no error
    assert mw.numel() == n * 120


# ---------------------------------------------------------------- kmx_histogram on dirty reads: marks + the histogram form of the sweep
@pytest.mark.parametrize("b", [10, 14, 16, 20, 22, 24, 29])
@pytest.mark.parametrize("k,L,hk,ragged", [(31, 150, 31, False), (21, 100, 0, False), (13, 250, 13, False), (31, 150, 31, True), (17, 250, 9, True), (5, 150, 5, False)])
def test_histogram_with_dirty_reads(ctx, orc, b, k, L, hk, ragged):
    """every histogram sink (block-private LDS tables <= 2^14, one- and two-level partitions, device atomics) no longer rolls a tile
    that holds an invalid byte: the tile takes the fast path, its reads are marked, and the histogram form of the sweep subtracts
    the windows that hold the byte -- every bucket against the oracle, two calls in a row into one table"""
    from kmers_amd import _lib

    n = 64 * 700 + 11
    rng = np.random.default_rng(1000 * b + k + L)
    if ragged:
        host, offs = _ragged(ctx, n, max(L // 3, 1), L, 31 * b + k, k, 0.15)
        o = orc.histogram(host, n, L, k, hk, b, offsets=offs)
        d_off = ctx.to_device(offs)
    else:
        host = ctx.gen_reads(n * L, first_byte=b * k).cpu().numpy().copy()
        _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, 0.15)
        host[(64 * 700 - 1) * L + L - 1] = ord("N")
        o = orc.histogram(host, n, L, k, hk, b)
        d_off = None
    hasher = _lib.HASH_LEX if hk else _lib.HASH_IDENTITY
    dev = ctx.to_device(host)
    g = ctx.histogram(dev, n, L, k, hasher, hk, b, offsets=d_off)
    assert np.array_equal(g.cpu().numpy().view(np.uint64), o)
    ctx.histogram(dev, n, L, k, hasher, hk, b, offsets=d_off, counts=g)
    assert np.array_equal(g.cpu().numpy().view(np.uint64), 2 * o)


# ------------------------------------------------------------------ kmx_canonical_reduce_host, and the scan that closes its own launch
# (round 6: the last block stores the summary, puts the queue block back, and can leave the answer in pinned host words)

HOST_CASES = [  # (k, L, n): the one-launch path (uniform, L <= 256, 13 <= k <= 31), partial last tiles of every size class, tiny batches
    (31, 150, 64 * 300 + 17), (31, 150, 64 * 40), (31, 150, 63), (31, 150, 1), (31, 150, 65), (21, 100, 64 * 100 + 32), (13, 36, 1000),
    (17, 250, 64 * 50 + 63), (27, 161, 4000), (31, 256, 777),
    # ... and what leaves it: k below the bit-sliced domain, reads above a frame (segments)
    (9, 150, 3000), (12, 100, 64 * 20 + 5), (31, 300, 2000), (21, 1000, 500),
]


@pytest.mark.parametrize("k,L,n", HOST_CASES)
@pytest.mark.parametrize("share", [0.0, 0.05])
def test_reduce_host(ctx, orc, k, L, n, share):
    from kmers_amd import _lib

    rng = np.random.default_rng(7 * k + L + n)
    host = ctx.gen_reads(n * L, first_byte=k + L).cpu().numpy().copy()
    if share:
        _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, share)
        host[(n - 1) * L + L - 1] = ord("N")
    d = ctx.to_device(host)
    for flags in (0, _lib.REDUCE_SUM_FW):
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        g = ctx.canonical_reduce_host(d, n, L, k, _lib.HASH_LEX, k, flags)
        _check1(g, o, with_fw=bool(flags))
        if not flags:
            assert g.sum_fw == 0
        g2 = ctx.canonical_reduce(d, n, L, k, _lib.HASH_LEX, k, flags)
        _check1(g2, o, with_fw=bool(flags))
    # another hasher: a second kernel rewrites the fold -- the call must take the long way and still be right
    o = orc.canonical_reduce(host, n, L, k, hasher_k=min(k + 1, 32))
    g = ctx.canonical_reduce_host(d, n, L, k, _lib.HASH_LEX, min(k + 1, 32), 0)
    _check1(g, o, with_fw=False)


def test_reduce_host_behind_offsets(ctx, orc):
    """reads behind an offsets array (uniform ones through the device-side gate, trimmed ones on the ragged scan): the long way"""
    from kmers_amd import _lib

    k, L, n = 31, 150, 64 * 120 + 9
    rng = np.random.default_rng(5)
    for trimmed in (False, True):
        lens = np.full(n, L, dtype=np.int64)
        if trimmed:
            sel = rng.random(n) < 0.05
            lens[sel] = rng.integers(20, L, size=int(sel.sum()))
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        host = ctx.gen_reads(int(offs[-1]), first_byte=3).cpu().numpy().copy()
        host[int(offs[5]) + 7] = ord("N")
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k, offsets=offs)
        g = ctx.canonical_reduce_host(ctx.to_device(host), n, L, k, _lib.HASH_LEX, k, 0, offsets=ctx.to_device(offs))
        _check1(g, o, with_fw=False)


def test_queue_block_survives_mixed_calls(ctx, orc):
    """The bit-sliced scan leaves the queue block as it found it and kmx_canonical_reduce stops clearing it; every other user of
    the ticket heads (histogram, materialise, the word-domain scan, the ragged scans, two-word k) clears for itself and leaves its
    tickets behind.  Any order of calls must give the answers of a fresh context."""
    from kmers_amd import _lib

    k, L, n = 31, 150, 64 * 90 + 11
    host = ctx.gen_reads(n * L, first_byte=11).cpu().numpy().copy()
    host[100 * L + 5] = ord("N")
    d = ctx.to_device(host)
    o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
    o9 = orc.canonical_reduce(host, n, L, 9, hasher_k=9)
    o2 = orc.canonical_reduce2(host, n, L, 63, with_hash=True)
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    d_offs = ctx.to_device(offs)
    steps = [
        lambda: _check1(ctx.canonical_reduce_host(d, n, L, k, _lib.HASH_LEX, k, 0), o, False),
        lambda: _check1(ctx.canonical_reduce(d, n, L, k, _lib.HASH_LEX, k, 0), o, False),
        lambda: ctx.histogram(d, n, L, k, _lib.HASH_LEX, k, 16),
        lambda: _check1(ctx.canonical_reduce(d, n, L, 9, _lib.HASH_LEX, 9, 0), o9, False),     # the word-domain scan
        lambda: ctx.canonical_windows(d, 2000, L, k, want=("canon",)),
        lambda: _check2(ctx.canonical_reduce2(d, n, L, 63, with_hash=True), o2),
        lambda: _check1(ctx.canonical_reduce_host(d, n, L, k, _lib.HASH_LEX, k, 0, offsets=d_offs), o, False),   # the gated pair
    ]
    rng = np.random.default_rng(99)
    for _ in range(6):
        for i in rng.permutation(len(steps)):
            steps[i]()
            steps[0]()
            steps[1]()


def test_reduce_host_many_in_a_row(ctx, orc):
    """the token in the pinned words: a thousand calls of alternating sizes, every answer checked against the first of its size"""
    from kmers_amd import _lib

    k, L = 31, 150
    sizes = [64 * 3 + 1, 5000, 64 * 700]
    host = ctx.gen_reads(max(sizes) * L, first_byte=1).cpu().numpy().copy()
    d = ctx.to_device(host)
    want = {}
    for n in sizes:
        want[n] = orc.canonical_reduce(host[: n * L], n, L, k, hasher_k=k)
    for i in range(1000):
        n = sizes[i % 3]
        _check1(ctx.canonical_reduce_host(d, n, L, k, _lib.HASH_LEX, k, 0), want[n], False)


# ------------------------------------------------------------------ the 11-word frame (uniform reads of 161..176 bases, round 6)
@pytest.mark.parametrize("k", [13, 16, 17, 21, 24, 27, 31])
@pytest.mark.parametrize("L", [161, 168, 175, 176])
def test_frame_of_11_words(ctx, orc, k, L):
    """both ends of the frame's range, every k family; W <= 160 takes the frame, k = 13..16 at 173..176 bases leave it (13-word frame);
    from an unaligned base (one more chunk per tile: 176 bases then leave the frame too); clean and with invalid bytes"""
    import torch
    from kmers_amd import _lib

    n = 64 * 150 + 9
    rng = np.random.default_rng(31 * k + L)
    host = ctx.gen_reads(n * L, first_byte=k + L).cpu().numpy().copy()
    for dirty in (False, True):
        if dirty:
            _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, 0.03)
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        for lead in (0, 5):
            big = torch.empty(n * L + 16, dtype=torch.uint8, device="cuda")
            big[lead : lead + n * L] = torch.from_numpy(host).cuda()
            g = ctx.canonical_reduce(big[lead : lead + n * L], n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
            _check1(g, o)


def test_frame_of_11_words_at_size(ctx, orc):
    """every wave scans many tiles (the fold of the fp32 accumulators, the ticket heads running dry): 3e6 reads of 170 bases"""
    from kmers_amd import _lib

    k, L, n = 31, 170, 3_000_000 + 21
    host = ctx.gen_reads(n * L, first_byte=9).cpu().numpy().copy()
    host[12345 * L + 7] = ord("N")
    o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
    g = ctx.canonical_reduce(ctx.to_device(host), n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _check1(g, o)


# ------------------------------------------------------------------ k = 9..12 on the bit-sliced scan (round 6; the word-domain scan before)
@pytest.mark.parametrize("k", [9, 10, 11, 12])
def test_small_k_on_the_bitsliced_scan(ctx, orc, k):
    """every frame (reads of 36 .. 256 bases), segments of longer reads, reads behind offsets (with a bound, without, trimmed, a bound
    above the frames: segments cut on the device), invalid bytes everywhere, the forward sum"""
    from kmers_amd import _lib

    rng = np.random.default_rng(k)
    for L in (36, 80, 100, 128, 150, 170, 200, 256, 300, 1000):
        n = 64 * 60 + 7 if L <= 256 else 900
        host = ctx.gen_reads(n * L, first_byte=k + L).cpu().numpy().copy()
        _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, 0.03)
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        g = ctx.canonical_reduce(ctx.to_device(host), n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _check1(g, o)
        g = ctx.canonical_reduce_host(ctx.to_device(host), n, L, k, _lib.HASH_LEX, k, 0)
        _check1(g, o, with_fw=False)
    n = 64 * 80 + 3
    for bound, lo, hi in ((150, 20, 151), (0, 5, 161), (250, 100, 251), (100, 0, 101), (2000, 200, 2001)):
        lens = rng.integers(lo, hi, n)
        lens[::3] = hi - 1
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        host = ctx.gen_reads(int(offs[-1]), first_byte=k).cpu().numpy().copy()
        _dirty(host, n, offs[:-1].astype(np.int64), lens, rng, 0.03)
        o = orc.canonical_reduce(host, n, bound, k, hasher_k=k, offsets=offs)
        for flags in (0, _lib.REDUCE_SUM_FW):
            g = ctx.canonical_reduce(ctx.to_device(host), n, bound, k, _lib.HASH_LEX, k, flags, offsets=ctx.to_device(offs))
            _check1(g, o, with_fw=bool(flags))


def test_small_k_packed(ctx, orc):
    """SeqVector input (2-bit packed): k = 9..12 take the bit-sliced scan too"""
    from kmers_amd import _lib

    for k in (9, 12):
        for L in (100, 150):
            n = 64 * 50 + 11
            host = ctx.gen_reads(n * L, first_byte=k).cpu().numpy().copy()
            o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
            words = ctx.seqvec_from_bytes(ctx.to_device(host))
            g = ctx.seqvec_canonical_reduce(words, n, L, k, _lib.HASH_LEX, k)
            _check1(g, o, with_fw=False)


# ------------------------------------------------------------------ materialise on reads with invalid bytes (round 6)
# The window sinks take the fast path on a tile with an invalid byte -- every read of it is marked, one store -- and the ZERO form of
# the sweep writes the spoiled windows' slots as the iterator leaves them: words 0, flags 0 (kmx.h; canonical_kmer_iterator.rs:50-66).
@pytest.mark.parametrize("k,L", [(31, 150), (21, 100), (13, 250), (9, 36), (2, 150), (31, 300)])
@pytest.mark.parametrize("want", [("canon",), ("fw", "rc", "canon", "flags"), ("flags",), ("fw", "flags")])
def test_windows_on_dirty_reads_at_size(ctx, orc, k, L, want):
    n = 64 * 700 + 13
    rng = np.random.default_rng(17 * k + L + len(want))
    host = ctx.gen_reads(n * L, first_byte=k).cpu().numpy().copy()
    _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, 0.03)
    host[(n - 1) * L + L - 1] = ord("N")
    host[5 * L] = ord("n")
    exp = dict(zip(("fw", "rc", "canon", "flags"), orc.canonical_windows(host, n, L, k)))
    outs = ctx.canonical_windows(ctx.to_device(host), n, L, k, want=want)
    for name in want:
        got = outs[name].cpu().numpy()
        got = got.view(np.uint64) if name != "flags" else got
        assert np.array_equal(got, exp[name]), name
    # ... and the mask array is all zero again: a clean call behind it gives clean answers
    clean = ctx.gen_reads(n * L, first_byte=k).cpu().numpy().copy()
    exp_c = orc.canonical_windows(clean, n, L, k)[2]
    got_c = ctx.canonical_windows(ctx.to_device(clean), n, L, k, want=("canon",))["canon"].cpu().numpy().view(np.uint64)
    assert np.array_equal(got_c, exp_c)


@pytest.mark.parametrize("bound", [150, 0, 250, 1000])
def test_windows_on_dirty_ragged_reads(ctx, orc, bound):
    k, n = 31, 64 * 300 + 9
    rng = np.random.default_rng(bound + 1)
    hi = bound if bound else 160
    lens = rng.integers(max(hi // 3, 1), hi + 1, n)
    lens[::5] = hi
    lens[7] = 0
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = ctx.gen_reads(int(offs[-1]), first_byte=bound).cpu().numpy().copy()
    _dirty(host, n, offs[:-1].astype(np.int64), lens, rng, 0.04)
    exp = dict(zip(("fw", "rc", "canon", "flags"), orc.canonical_windows(host, n, bound, k, offsets=offs)))
    for want in (("canon",), ("fw", "rc", "canon", "flags")):
        outs = ctx.canonical_windows(ctx.to_device(host), n, bound, k, offsets=ctx.to_device(offs), host_offsets=offs, want=want)
        for name in want:
            got = outs[name].cpu().numpy()
            got = got.view(np.uint64) if name != "flags" else got
            assert np.array_equal(got, exp[name]), (name, want)


@pytest.mark.parametrize("k,L", [(33, 150), (63, 150), (64, 250), (47, 100)])
def test_windows2_on_dirty_reads_at_size(ctx, orc, k, L):
    """[u64;2] materialise: a tile with an invalid byte stays on the tiled path, the ZERO sweep writes the spoiled slots (two words each)"""
    n = 64 * 500 + 21
    rng = np.random.default_rng(k + L)
    host = ctx.gen_reads(n * L, first_byte=k).cpu().numpy().copy()
    _dirty(host, n, np.arange(n, dtype=np.int64) * L, np.full(n, L), rng, 0.03)
    host[(n - 1) * L + L - 1] = ord("N")
    fw, rc, canon, flags = orc.canonical_windows2(host, n, L, k)
    outs = ctx.canonical_windows2(ctx.to_device(host), n, L, k)
    for name, exp in (("fw", fw), ("rc", rc), ("canon", canon)):
        assert np.array_equal(outs[name].cpu().numpy().view(np.uint64).reshape(-1, 2), exp), name
    assert np.array_equal(outs["flags"].cpu().numpy(), flags)
    # ragged: the same reads behind offsets, some trimmed
    lens = np.full(n, L, dtype=np.int64)
    sel = rng.random(n) < 0.1
    lens[sel] = rng.integers(k // 2, L, size=int(sel.sum()))
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    hostr = np.concatenate([host[i * L : i * L + lens[i]] for i in range(0, n, 1)]) if n <= 40000 else None
    if hostr is not None:
        fw, rc, canon, flags = orc.canonical_windows2(hostr, n, L, k, offsets=offs)
        outs = ctx.canonical_windows2(ctx.to_device(hostr), n, L, k, offsets=ctx.to_device(offs), host_offsets=offs)
        for name, exp in (("fw", fw), ("rc", rc), ("canon", canon)):
            assert np.array_equal(outs[name].cpu().numpy().view(np.uint64).reshape(-1, 2), exp), name
        assert np.array_equal(outs["flags"].cpu().numpy(), flags)
