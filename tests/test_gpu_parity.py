"""GPU parity tests: the HIP path (through the libkmx C ABI) against the CPU oracle on the
same inputs, bit-exact (all arithmetic is unsigned integer).  Run with `-m gpu` on an MI355X."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M64 = 2**64 - 1


@pytest.fixture(scope="module")
def ctx():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from kmers_amd.api import Context

    c = Context()
    yield c
    c.close()


def _dirty(rng, n, p_bad=0.02):
    alpha = np.frombuffer(b"ACGTacgt", np.uint8)
    a = alpha[rng.integers(0, 8, n)]
    bad = rng.random(n) < p_bad
    a = a.copy()
    a[bad] = rng.integers(0, 256, int(bad.sum()), dtype=np.uint8)
    return a


def _cmp_summary(g, o, want_hash, want_sumfw):
    assert g.n_valid == o.n_valid
    assert g.sum_canon == o.sum_canon
    assert g.xor_hash == (o.xor_hash if want_hash else 0)
    assert g.sum_fw == (o.sum_fw if want_sumfw else 0)


# ------------------------------------------------------------------ generator

@pytest.mark.parametrize("first,n", [(0, 1 << 16), (16, 4096 + 5), (7, 1000), (150 * 64, 150 * 1000), (31, 1)])
def test_gen_reads_matches_oracle(ctx, orc, first, n):
    from kmers_amd.api import SEED_DEFAULT

    g = ctx.gen_reads(n, SEED_DEFAULT, first).cpu().numpy()
    assert (g == orc.gen_reads(SEED_DEFAULT, first, n)).all()


# --------------------------------------------------------- reduce, clean input

@pytest.mark.parametrize("k", [1, 2, 5, 12] + list(range(13, 32)))
@pytest.mark.parametrize("L,n_reads", [(150, 10_000), (150, 63), (150, 64), (150, 257)])
def test_reduce_uniform_clean(ctx, orc, k, L, n_reads):
    from kmers_amd import _lib

    bases = ctx.gen_reads(n_reads * L, first_byte=L * 12345)
    host = bases.cpu().numpy()
    o = orc.canonical_reduce(host, n_reads, L, k, hasher_k=k)
    g = ctx.canonical_reduce(bases, n_reads, L, k)
    _cmp_summary(g, o, False, False)
    g = ctx.canonical_reduce(bases, n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _cmp_summary(g, o, True, True)
    g = ctx.canonical_reduce(bases, n_reads, L, k, _lib.HASH_LEX, k, 0)
    _cmp_summary(g, o, True, False)
    assert o.n_valid == n_reads * (L - k + 1)


@pytest.mark.parametrize("L", [31, 32, 33, 47, 48, 100, 101, 112, 149, 151, 159, 160, 161, 176, 250, 255, 256, 257, 300])
@pytest.mark.parametrize("k", [11, 14, 15, 17, 20, 23, 28, 29, 30, 31])
def test_reduce_uniform_lengths(ctx, orc, L, k):
    from kmers_amd import _lib

    n_reads = 64 * 5 + 17
    bases = ctx.gen_reads(n_reads * L, first_byte=99)
    host = bases.cpu().numpy()
    o = orc.canonical_reduce(host, n_reads, L, k, hasher_k=k)
    g = ctx.canonical_reduce(bases, n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _cmp_summary(g, o, True, True)


def test_reduce_config1_shape_matches_compute_naive(ctx, orc):
    """BASELINE config 1: 10k x 150 bp, k=31; sum_fw == benches' compute_naive per read."""
    from kmers_amd import _lib

    L, n, k = 150, 10_000, 31
    bases = ctx.gen_reads(n * L)
    host = bases.cpu().numpy()
    g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    tot = 0
    for r in range(0, n, 997):
        tot += 1
        assert orc.compute_naive(host[r * L:(r + 1) * L], k) == orc.canonical_reduce(host[r * L:(r + 1) * L], 1, L, k).sum_fw
    o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
    _cmp_summary(g, o, True, True)
    assert g.n_valid == n * 120


def test_reduce_hasher_k_differs_and_identity(ctx, orc):
    from kmers_amd import _lib

    L, n, k = 150, 1000, 21
    bases = ctx.gen_reads(n * L, first_byte=5)
    host = bases.cpu().numpy()
    o = orc.canonical_reduce(host, n, L, k, hasher_k=25)
    g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, 25, 0)
    _cmp_summary(g, o, True, False)
    # identity hasher: xor of the canonical words themselves
    _, _, canon, flags = orc.canonical_windows(host, n, L, k)
    g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_IDENTITY, 0, 0)
    assert g.xor_hash == int(np.bitwise_xor.reduce(canon[flags & 1 == 1]))


def test_reduce_misaligned_base_pointer(ctx, orc):
    L, n, k = 150, 500, 31
    buf = ctx.gen_reads(n * L + 16)
    view = buf[3:3 + n * L]  # 3-byte offset: not 16-byte aligned
    host = view.cpu().numpy()
    o = orc.canonical_reduce(host, n, L, k)
    r = ctx._reads(None, 0, 0, None)
    out = ctx.empty(4, __import__("torch").int64)
    from kmers_amd import _lib

    rd = _lib.Reads(C.c_void_p(view.data_ptr()), n, L, None)
    ctx._ck(ctx.lib.kmx_canonical_reduce(ctx._h, C.byref(rd), k, 0, 0, 0, C.c_void_p(out.data_ptr())))
    v = out.cpu().numpy().view(np.uint64)
    assert (int(v[0]), int(v[1])) == (o.n_valid, o.sum_canon)
    del r


# --------------------------------------------------------- reduce, dirty input

@pytest.mark.parametrize("k", [1, 7, 13, 15, 16, 17, 19, 21, 22, 25, 27, 30, 31])
@pytest.mark.parametrize("p_bad", [0.0005, 0.02, 0.5])
def test_reduce_uniform_dirty(ctx, orc, k, p_bad):
    from kmers_amd import _lib

    rng = np.random.default_rng(1000 + k)
    L, n = 150, 64 * 40 + 9
    host = _dirty(rng, n * L, p_bad)
    bases = ctx.to_device(host)
    o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
    g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _cmp_summary(g, o, True, True)


def test_reduce_every_byte_value_once(ctx, orc):
    """one bad byte at a time, every one of the 256 byte values, placed inside an otherwise clean tile"""
    L, n, k = 150, 64, 31
    clean = orc.gen_reads(1, 0, n * L)
    for b in range(256):
        host = clean.copy()
        host[(b * 37) % (n * L)] = b
        g = ctx.canonical_reduce(ctx.to_device(host), n, L, k)
        o = orc.canonical_reduce(host, n, L, k)
        assert (g.n_valid, g.sum_canon) == (o.n_valid, o.sum_canon), b


def test_reduce_edge_shapes(ctx, orc):
    from kmers_amd import _lib

    k = 31
    # empty batch
    g = ctx.canonical_reduce(ctx.empty(0, __import__("torch").uint8), 0, 150, k)
    assert (g.n_valid, g.sum_canon) == (0, 0)
    # reads shorter than k: iterator exhausted immediately (canonical_kmer_iterator.rs:50,69)
    bases = ctx.gen_reads(100 * 20)
    g = ctx.canonical_reduce(bases, 100, 20, k)
    assert g.n_valid == 0
    # L == k
    bases = ctx.gen_reads(77 * 31)
    host = bases.cpu().numpy()
    g = ctx.canonical_reduce(bases, 77, 31, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _cmp_summary(g, orc.canonical_reduce(host, 77, 31, k, hasher_k=k), True, True)
    # homopolymers / palindromes: fw == rc ties
    host = np.frombuffer((b"A" * 150 + b"T" * 150 + b"AT" * 75 + b"ACGT" * 37 + b"AC") * 32, np.uint8)
    n = host.size // 150
    g = ctx.canonical_reduce(ctx.to_device(host), n, 150, 30)
    o = orc.canonical_reduce(host, n, 150, 30)
    assert (g.n_valid, g.sum_canon) == (o.n_valid, o.sum_canon)


def test_k_range_errors(ctx):
    from kmers_amd import _lib

    bases = ctx.gen_reads(150 * 64)
    for k in (0, 32, 33):
        with pytest.raises(_lib.KmxError) as e:
            ctx.canonical_reduce(bases, 64, 150, k)
        assert e.value.status == _lib.E_K_RANGE
    with pytest.raises(_lib.KmxError) as e:
        ctx.canonical_reduce2(bases, 64, 150, 32)
    assert e.value.status == _lib.E_K_RANGE


def test_oracle_shows_the_k32_quirk(orc):
    """why k=32 is rejected: MASK_TABLE[32]==0 (kmer.rs:617) zeroes rc in the reference's own rolling"""
    host = orc.gen_reads(3, 0, 64)
    fw, rc, canon, flags = orc.canonical_windows(host, 1, 64, 32)
    assert (rc == 0).all() and (canon == 0).all() and flags.all()


@pytest.mark.parametrize("k", [31, 21, 13, 16, 24, 30])
@pytest.mark.parametrize("L", [150, 100, 250])
def test_reduce_reads_with_invalid_bytes_second_pass(ctx, orc, k, L):
    """tiles with an invalid byte: flagged by the main pass, run by the second pass with the offending reads blanked out
    and rolled 64 at a time.  Every read of a tile dirty (buffer flushes), one dirty read per tile, N as the last byte of a
    read / the first of the next (they share a 16-byte chunk), runs of N, clean tiles in between, a partial last tile."""
    from kmers_amd import _lib
    rng = np.random.default_rng(1000 * k + L)
    n = 64 * 40 + 23
    host = np.frombuffer(b"ACGTacgt", np.uint8)[rng.integers(0, 8, n * L)].copy()
    reads = host.reshape(n, L)
    reads[0:64, rng.integers(0, L, 64)] = ord("N")                 # tile 0: every read dirty (a column per read, many hits)
    reads[64:128, :][np.arange(64), rng.integers(0, L, 64)] = ord("N")   # tile 1: one N per read
    reads[64 * 3 + 5, L - 1] = ord("N")                             # tile 3: last byte of a read
    reads[64 * 3 + 6, 0] = ord("n")                                 #         first byte of the next
    reads[64 * 5 + 63, L // 2: L // 2 + 40] = ord("N")              # tile 5: a run of N in its last read
    reads[64 * 6, :] = ord("N")                                     # tile 6: a read of N only
    for t in range(8, 40, 3):                                       # scattered
        reads[64 * t + int(rng.integers(0, 64)), int(rng.integers(0, L))] = int(rng.integers(0, 256))
    reads[n - 3, 7] = ord("-")                                      # the partial last tile
    bases = ctx.to_device(host)
    for hasher, flags in ((_lib.HASH_LEX, _lib.REDUCE_SUM_FW), (_lib.HASH_NONE, 0)):
        g = ctx.canonical_reduce(bases, n, L, k, hasher, k if hasher else 0, flags)
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        _cmp_summary(g, o, hasher != 0, flags != 0)
    # twice in a row on the same context: the flags must be back to zero after a call
    g2 = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _cmp_summary(g2, orc.canonical_reduce(host, n, L, k, hasher_k=k), True, True)
    clean = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * L)].copy()
    g3 = ctx.canonical_reduce(ctx.to_device(clean), n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _cmp_summary(g3, orc.canonical_reduce(clean, n, L, k, hasher_k=k), True, True)


@pytest.mark.parametrize("k", [31, 21])
@pytest.mark.parametrize("hint", [160, 0])
def test_reduce_ragged_reads_with_invalid_bytes_second_pass(ctx, orc, k, hint):
    """the second pass on ragged reads: a set-aside read counts as an empty one (no bases, no windows), N at read ends that
    share a chunk with the neighbour, empty reads and reads shorter than k next to dirty ones, a tile outside the frame"""
    from kmers_amd import _lib
    rng = np.random.default_rng(77 * k + hint)
    lens = rng.integers(100, 161, size=64 * 30 + 11)
    lens[rng.integers(0, len(lens), 40)] = rng.choice([0, 1, k - 1, k, 45], 40)
    lens[64 * 7 + 3] = 400                                           # tile 7 leaves the frame: rolls as a whole
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = np.frombuffer(b"ACGTacgt", np.uint8)[rng.integers(0, 8, int(offsets[-1]))].copy()
    for r in rng.integers(0, len(lens), 300):                        # ~1 dirty read in 6, many tiles hit
        if lens[r]:
            host[int(offsets[r]) + int(rng.integers(0, lens[r]))] = ord("N")
    for r in range(0, 64):                                           # tile 0: every non-empty read dirty, at its last byte
        if lens[r]:
            host[int(offsets[r + 1]) - 1] = ord("N")
    host[int(offsets[64 * 7 + 5])] = ord("N") if lens[64 * 7 + 5] else host[int(offsets[64 * 7 + 5])]
    bases, d_off = ctx.to_device(host), ctx.to_device(offsets)
    o = orc.canonical_reduce(host, len(lens), 0, k, hasher_k=k, offsets=offsets)
    for _ in range(2):
        g = ctx.canonical_reduce(bases, len(lens), hint, k, _lib.HASH_LEX, k, 0, offsets=d_off)
        _cmp_summary(g, o, True, False)


@pytest.mark.parametrize("k", [63, 33, 47])
def test_reduce2_reads_with_invalid_bytes_second_pass(ctx, orc, k):
    """[u64;2] k-mers: the second pass for tiles with invalid bytes (every read of a tile dirty, scattered, read ends)"""
    rng = np.random.default_rng(k)
    L, n = 150, 64 * 20 + 9
    host = np.frombuffer(b"ACGTacgt", np.uint8)[rng.integers(0, 8, n * L)].copy()
    reads = host.reshape(n, L)
    reads[0:64, :][np.arange(64), rng.integers(0, L, 64)] = ord("N")
    reads[64 * 2 + 5, L - 1] = ord("N")
    reads[64 * 2 + 6, 0] = ord("n")
    for t in range(4, 20, 2):
        reads[64 * t + int(rng.integers(0, 64)), int(rng.integers(0, L))] = int(rng.integers(0, 256))
    reads[n - 2, 70] = ord("N")
    bases = ctx.to_device(host)
    for with_hash in (True, False):
        o = orc.canonical_reduce2(host, n, L, k, with_hash)
        for _ in range(2):
            g = ctx.canonical_reduce2(bases, n, L, k, with_hash)
            assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


# --------------------------------------------------------------- ragged reads

def test_reduce_and_windows_ragged(ctx, orc):
    from kmers_amd import _lib

    rng = np.random.default_rng(7)
    k = 21
    lens = rng.choice([0, 1, 20, 21, 22, 40, 150, 151, 300], size=500)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]), 0.01)
    bases = ctx.to_device(host)
    d_off = ctx.to_device(offsets)
    o = orc.canonical_reduce(host, len(lens), 0, k, hasher_k=k, offsets=offsets)
    g = ctx.canonical_reduce(bases, len(lens), 0, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=d_off)
    _cmp_summary(g, o, True, True)
    fw, rc, canon, flags = orc.canonical_windows(host, len(lens), 0, k, offsets=offsets)
    outs = ctx.canonical_windows(bases, len(lens), 0, k, offsets=d_off, host_offsets=offsets)
    assert (outs["fw"].cpu().numpy().view(np.uint64) == fw).all()
    assert (outs["rc"].cpu().numpy().view(np.uint64) == rc).all()
    assert (outs["canon"].cpu().numpy().view(np.uint64) == canon).all()
    assert (outs["flags"].cpu().numpy() == flags).all()


@pytest.mark.parametrize("k", [31, 21, 16, 5])
@pytest.mark.parametrize("hint", [0, 160, 256])
def test_windows_ragged_tiled_kernel(ctx, orc, k, hint):
    """materialise mode for ragged reads on the tiled kernel: per-read slot bases and window counts from win_offsets,
    reads shorter than k (no windows), empty reads, reads longer than the frame (exact path), N bytes"""
    rng = np.random.default_rng(100 * k + hint)
    lens = rng.integers(100, 161, size=64 * 9 + 21)
    lens[rng.integers(0, len(lens), 12)] = rng.choice([0, 1, k - 1, k, k + 1, 40], 12)
    lens[5] = 300 if hint != 256 else 200
    lens[64 * 4:64 * 5] = 150
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]), 0.0008)
    bases, d_off = ctx.to_device(host), ctx.to_device(offsets)
    fw, rc, canon, flags = orc.canonical_windows(host, len(lens), 0, k, offsets=offsets)
    outs = ctx.canonical_windows(bases, len(lens), hint, k, offsets=d_off, host_offsets=offsets)
    for name, exp in (("fw", fw), ("rc", rc), ("canon", canon)):
        assert (outs[name].cpu().numpy().view(np.uint64) == exp).all(), name
    assert (outs["flags"].cpu().numpy() == flags).all()
    only = ctx.canonical_windows(bases, len(lens), hint, k, offsets=d_off, host_offsets=offsets, want=("canon",))
    assert (only["canon"].cpu().numpy().view(np.uint64) == canon).all()


@pytest.mark.parametrize("k", [31, 21, 11, 5])
@pytest.mark.parametrize("hint", [0, 160, 256])
def test_reduce_ragged_tiled_kernel(ctx, orc, k, hint):
    """ragged reads on the tiled word-domain kernel: mostly 100..160 bp (fits both frames), some empty / shorter than k /
    longer than the frame (those tiles take the exact path), N bytes, a partial last tile, the last tile's tail chunk"""
    from kmers_amd import _lib

    rng = np.random.default_rng(1000 + k + hint)
    lens = rng.integers(100, 161, size=64 * 40 + 21)
    lens[rng.integers(0, lens.size, 30)] = rng.choice([0, 1, k - 1, k, k + 1, 255, 256, 257, 400], size=30)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]), 0.0005)
    bases, d_off = ctx.to_device(host), ctx.to_device(offsets)
    o = orc.canonical_reduce(host, len(lens), 0, k, hasher_k=k, offsets=offsets)
    g = ctx.canonical_reduce(bases, len(lens), hint, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=d_off)
    _cmp_summary(g, o, True, True)
    g0 = ctx.canonical_reduce(bases, len(lens), hint, k, offsets=d_off)          # k in {21, 31}: ragged bit-sliced kernel
    assert (g0.n_valid, g0.sum_canon) == (o.n_valid, o.sum_canon)
    g1 = ctx.canonical_reduce(bases, len(lens), hint, k, _lib.HASH_LEX, k, 0, offsets=d_off)
    assert (g1.n_valid, g1.sum_canon, g1.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)


def test_reduce_ragged_equals_uniform_when_lengths_are_equal(ctx, orc):
    rng = np.random.default_rng(31)
    L, n, k = 150, 64 * 30 + 5, 31
    host = _dirty(rng, n * L, 0.0002)
    offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    a = ctx.canonical_reduce(ctx.to_device(host), n, L, k)
    b = ctx.canonical_reduce(ctx.to_device(host), n, 160, k, offsets=ctx.to_device(offsets))
    assert (a.n_valid, a.sum_canon) == (b.n_valid, b.sum_canon)


@pytest.mark.parametrize("b", [12, 20])
def test_histogram_ragged(ctx, orc, b):
    rng = np.random.default_rng(b + 5)
    k = 31
    lens = rng.integers(90, 161, size=64 * 25 + 9)
    lens[::97] = 300
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]), 0.0005)
    o = orc.histogram(host, len(lens), 0, k, k, b, offsets=offsets)
    g = ctx.histogram(ctx.to_device(host), len(lens), 0, k, 1, k, b, offsets=ctx.to_device(offsets))
    assert (g.cpu().numpy().view(np.uint64) == o).all()


# ---------------------------------------------------------------- materialise

@pytest.mark.parametrize("k", [1, 16, 21, 31])
@pytest.mark.parametrize("p_bad", [0.0, 0.03])
def test_windows_uniform(ctx, orc, k, p_bad):
    rng = np.random.default_rng(k)
    L, n = 150, 300
    host = _dirty(rng, n * L, p_bad)
    bases = ctx.to_device(host)
    fw, rc, canon, flags = orc.canonical_windows(host, n, L, k)
    outs = ctx.canonical_windows(bases, n, L, k)
    assert (outs["flags"].cpu().numpy() == flags).all()
    assert (outs["fw"].cpu().numpy().view(np.uint64) == fw).all()
    assert (outs["rc"].cpu().numpy().view(np.uint64) == rc).all()
    assert (outs["canon"].cpu().numpy().view(np.uint64) == canon).all()
    only = ctx.canonical_windows(bases, n, L, k, want=("canon",))
    assert (only["canon"].cpu().numpy().view(np.uint64) == canon).all()


@pytest.mark.parametrize("L,k", [(150, 31), (150, 21), (158, 31), (100, 31), (160, 27), (47, 31), (64, 17), (151, 2), (250, 31), (33, 18), (31, 31)])
def test_windows_single_array_line_aligned_writeback(ctx, orc, L, k):
    """one u64 array requested: the write-back goes out in whole 128-byte lines of the output, shifted per read by
    read*W mod 16 slots (W = L-k+1 of every residue here), with the head and tail of each read written separately"""
    rng = np.random.default_rng(L * 100 + k)
    n = 64 * 5 + 17
    host = _dirty(rng, n * L, 0.002)
    bases = ctx.to_device(host)
    fw, rc, canon, flags = orc.canonical_windows(host, n, L, k)
    for name, exp in (("canon", canon), ("fw", fw), ("rc", rc)):
        got = ctx.canonical_windows(bases, n, L, k, want=(name,))[name].cpu().numpy().view(np.uint64)
        assert (got == exp).all(), (name, int((got != exp).sum()))
    two = ctx.canonical_windows(bases, n, L, k, want=("canon", "flags"))
    assert (two["canon"].cpu().numpy().view(np.uint64) == canon).all() and (two["flags"].cpu().numpy() == flags).all()


def test_reference_iterator_kats_on_gpu(ctx, orc, kats):
    """canonical_kmer_iterator.rs:123-206 through the HIP path: the iterator state after init/inc/inc_by
    is the (fw, rc) of the n-th valid slot."""
    k = kats["iterator"]["k"]
    for case in kats["iterator"]["cases"]:
        r = kats["read_R"].encode()
        if case["insert_N_at"] is not None:
            i = case["insert_N_at"]
            r = r[:i] + b"N" + r[i:]
        outs = ctx.canonical_windows(ctx.to_device(r), 1, len(r), k)
        flags = outs["flags"].cpu().numpy()
        valid_pos = np.nonzero(flags & 1)[0]
        pos = int(valid_pos[case["inc_by"]])
        assert pos == case["expect_pos"]
        s = case["expect_window_start"]
        fk = orc.ck_from_bytes(r[s:s + k])
        assert int(outs["fw"].cpu().numpy().view(np.uint64)[pos]) == fk.fw.data
        assert int(outs["rc"].cpu().numpy().view(np.uint64)[pos]) == fk.rc.data
    d = kats["derived_R_k31"]
    r = kats["read_R"].encode()
    g = ctx.canonical_reduce(ctx.to_device(r), 1, len(r), k, 1, k, 1)
    assert g.n_valid == d["n_windows"] and g.sum_canon == int(d["sum_canon"], 16)


# ------------------------------------------------------------ [u64;2] k-mers

@pytest.mark.parametrize("k", [33, 47, 63, 64])
def test_reduce2_windows2(ctx, orc, k):
    rng = np.random.default_rng(k)
    L, n = 150, 200
    host = _dirty(rng, n * L, 0.004)
    bases = ctx.to_device(host)
    o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
    g = ctx.canonical_reduce2(bases, n, L, k, with_hash=True)
    assert tuple(getattr(g, f) for f, _ in g._fields_) == tuple(getattr(o, f) for f, _ in o._fields_)
    fw, rc, canon, flags = orc.canonical_windows2(host, n, L, k)
    outs = ctx.canonical_windows2(bases, n, L, k)
    assert (outs["flags"].cpu().numpy() == flags).all()
    assert (outs["fw"].cpu().numpy().view(np.uint64).reshape(-1, 2) == fw).all()
    assert (outs["rc"].cpu().numpy().view(np.uint64).reshape(-1, 2) == rc).all()
    assert (outs["canon"].cpu().numpy().view(np.uint64).reshape(-1, 2) == canon).all()


@pytest.mark.parametrize("k", list(range(33, 64, 2)) + [34, 64])
@pytest.mark.parametrize("L,n,p_bad", [(150, 64 * 30 + 7, 0.0), (150, 64 * 30 + 7, 0.0005), (100, 64 * 12, 0.0), (160, 64 * 9 + 1, 0.0)])
def test_reduce2_bitsliced_every_odd_k(ctx, orc, k, L, n, p_bad):
    """[u64;2] k-mers: odd k from 33 to 63 on the bit-sliced kernel (clean tiles; dirty tiles and the partial tile roll per
    lane), even k on the lane-per-read kernel -- all against the oracle's 128-bit rolling"""
    if L < k:
        pytest.skip("read shorter than k")
    rng = np.random.default_rng(k * 7 + L)
    host = _dirty(rng, n * L, p_bad)
    o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
    g = ctx.canonical_reduce2(ctx.to_device(host), n, L, k, with_hash=True)
    assert tuple(getattr(g, f) for f, _ in g._fields_) == tuple(getattr(o, f) for f, _ in o._fields_)


# ------------------------------------------------------------------ histogram

@pytest.mark.parametrize("hasher,hk", [(1, 31), (2, 0), (1, 20)])
def test_histogram(ctx, orc, hasher, hk):
    rng = np.random.default_rng(3)
    L, n, k, b = 150, 700, 31, 12
    host = _dirty(rng, n * L, 0.002)
    bases = ctx.to_device(host)
    o = orc.histogram(host, n, L, k, hk if hasher == 1 else 0, b)
    g = ctx.histogram(bases, n, L, k, hasher, hk, b).cpu().numpy().view(np.uint64)
    assert (g == o).all()
    # accumulates: second call doubles
    c = ctx.histogram(bases, n, L, k, hasher, hk, b)
    c = ctx.histogram(bases, n, L, k, hasher, hk, b, counts=c)
    assert (c.cpu().numpy().view(np.uint64) == 2 * o).all()


@pytest.mark.parametrize("b", [6, 12, 20])
def test_histogram_fast_kernel_many_tiles(ctx, orc, b):
    """uniform reads take the word-domain scan kernel with the histogram sink; clean and dirty tiles mixed"""
    rng = np.random.default_rng(b)
    L, n, k = 150, 64 * 90 + 11, 31
    host = _dirty(rng, n * L, 0.0005)
    bases = ctx.to_device(host)
    o = orc.histogram(host, n, L, k, k, b)
    g = ctx.histogram(bases, n, L, k, 1, k, b)
    assert (g.cpu().numpy().view(np.uint64) == o).all()
    g = ctx.histogram(bases, n, L, k, 1, k, b, counts=g)      # accumulates into the caller's counters
    assert (g.cpu().numpy().view(np.uint64) == 2 * o).all()
    o21 = orc.histogram(host, n, L, 21, 0, b)                 # identity hasher, other k
    g21 = ctx.histogram(bases, n, L, 21, 2, 0, b)
    assert (g21.cpu().numpy().view(np.uint64) == o21).all()


@pytest.mark.parametrize("b", [10, 14, 15, 18, 21, 22, 23])
def test_histogram_three_regimes(ctx, orc, b):
    """2^b <= 2^14: block-private LDS tables; 2^15..2^22: 64-way partition + LDS tables (2^22: two half tables per partition; here in several chunks of reads,
    forced by a 4 MiB work buffer); above: device atomics.  All three must equal the oracle, dirty tiles included."""
    rng = np.random.default_rng(100 + b)
    L, n, k = 150, 64 * 400 + 37, 31
    host = _dirty(rng, n * L, 0.0003)
    bases = ctx.to_device(host)
    o = orc.histogram(host, n, L, k, k, b)
    ctx.set_work_buffer_limit(4 << 20)
    try:
        g = ctx.histogram(bases, n, L, k, 1, k, b)
        assert (g.cpu().numpy().view(np.uint64) == o).all()
        g = ctx.histogram(bases, n, L, k, 1, k, b, counts=g)
        assert (g.cpu().numpy().view(np.uint64) == 2 * o).all()
    finally:
        ctx.set_work_buffer_limit(0)


@pytest.mark.parametrize("b", [12, 16, 20])
@pytest.mark.parametrize("L,k", [(150, 31), (100, 21), (250, 27), (64, 11)])
def test_histogram_shapes(ctx, orc, b, L, k):
    rng = np.random.default_rng(b * 1000 + L)
    n = 64 * 130 + 5
    host = _dirty(rng, n * L, 0.0002)
    o = orc.histogram(host, n, L, k, k, b)
    g = ctx.histogram(ctx.to_device(host), n, L, k, 1, k, b)
    assert (g.cpu().numpy().view(np.uint64) == o).all()


@pytest.mark.parametrize("b", [12, 20])
def test_histogram_adversarial_single_bucket(ctx, orc, b):
    """every k-mer identical: one staging row / one partition segment takes everything and must spill to the exact fallback"""
    L, n, k = 150, 64 * 200, 31
    host = np.frombuffer(b"A" * (n * L), dtype=np.uint8).copy()
    host[1234 * L + 7] = ord("N")
    o = orc.histogram(host, n, L, k, k, b)
    g = ctx.histogram(ctx.to_device(host), n, L, k, 1, k, b)
    g = g.cpu().numpy().view(np.uint64)
    assert int(g.sum()) == int(o.sum())
    assert (g == o).all()


# ---------------------------------------------------------------- elementwise

def test_kmers_from_bytes_and_word_ops(ctx, orc, kats):
    from kmers_amd import _lib

    L = orc.lib()
    rng = random.Random(11)
    for k in (1, 3, 14, 31, 32):
        seqs = ["".join(rng.choice("ACGTacgt") for _ in range(k)) for _ in range(300)]
        blob = "".join(seqs).encode()
        words = ctx.kmers_from_bytes(ctx.to_device(blob), len(seqs), k)
        w = words.cpu().numpy().view(np.uint64)
        exp = np.array([orc.kmer_from_bytes(s.encode()).data for s in seqs], np.uint64)
        assert (w == exp).all()
        rc = ctx.revcomp_words(words, k).cpu().numpy().view(np.uint64)
        assert (rc == np.array([L.kmo_revcomp_word(int(x), k) for x in exp], np.uint64)).all()
        canon, isc = ctx.canonical_words(words, k)
        for x, c, i in zip(exp, canon.cpu().numpy().view(np.uint64), isc.cpu().numpy()):
            km = orc.Kmer(k, int(x))
            assert int(c) == L.kmo_kmer_to_canonical(km).data
            assert bool(i) == bool(L.kmo_kmer_is_canonical(km))
        h = ctx.hash_words(words, _lib.HASH_LEX, k).cpu().numpy().view(np.uint64)
        assert (h == np.array([L.kmo_lex_hash_u64(int(x), k) for x in exp], np.uint64)).all()
    # strict semantics: Kmer::from panics on N (mod.rs:35) -> error + first offending index
    blob = b"ACGTACGTACNTACGT"
    with pytest.raises(_lib.KmxError) as e:
        ctx.kmers_from_bytes(ctx.to_device(blob), 4, 4)
    assert e.value.status == _lib.E_INVALID_BASE and e.value.first_bad == 10
    with pytest.raises(_lib.KmxError) as e:
        ctx.kmers_from_bytes(ctx.to_device(b"A" * 66), 2, 33)
    assert e.value.status == _lib.E_TOO_LONG
    # reference KATs (kmer.rs:386-424, hash.rs:83-104) through the GPU
    for s, r in kats["reverse_complement"]["cases"]:
        w = ctx.kmers_from_bytes(ctx.to_device(s.encode()), 1, len(s))
        assert int(ctx.revcomp_words(w, len(s)).cpu().numpy().view(np.uint64)[0]) == orc.kmer_from_bytes(r.encode()).data
    for s, v in kats["lex_hasher"]["cases"]:
        w = ctx.kmers_from_bytes(ctx.to_device(s.encode()), 1, len(s))
        assert int(ctx.hash_words(w, _lib.HASH_LEX, kats["lex_hasher"]["k"]).cpu().numpy().view(np.uint64)[0]) == v


def test_ck_shift_and_match(ctx, orc):
    L = orc.lib()
    rng = random.Random(5)
    k = 31
    n = 500
    words = np.array([rng.getrandbits(62) for _ in range(n)], np.uint64)
    cks = [L.kmo_ck_from_u64(int(w), k) for w in words]
    fw = ctx.to_device(np.array([c.fw.data for c in cks], np.uint64))
    rc = ctx.to_device(np.array([c.rc.data for c in cks], np.uint64))
    bases = np.array([rng.randrange(4) for _ in range(n)], np.uint8)
    d_b = ctx.to_device(bases)
    for append in (True, False):
        f2, r2 = fw.clone(), rc.clone()
        dropped = ctx.ck_shift(f2, r2, d_b, k, append=append).cpu().numpy()
        for i in range(n):
            c = L.kmo_ck_from_u64(int(words[i]), k)
            d = (L.kmo_ck_append_base if append else L.kmo_ck_prepend_base)(C.byref(c), int(bases[i]))
            assert d == dropped[i]
            assert int(f2[i].item()) & M64 == c.fw.data and int(r2[i].item()) & M64 == c.rc.data
    other = np.array([cks[i].fw.data if i % 3 == 0 else (cks[i].rc.data if i % 3 == 1 else rng.getrandbits(62)) for i in range(n)], np.uint64)
    m = ctx.match_words(fw, rc, ctx.to_device(other)).cpu().numpy()
    for i in range(n):
        assert m[i] == L.kmo_ck_get_word_equivalency(C.byref(cks[i]), int(other[i]))


def test_generic_encodings(ctx, orc, kats):
    encs = kats["naive_encodings"]["enc_bytes"]
    rng = random.Random(2)
    # reference KATs (naive.rs:297-445) for P=u64 through the GPU, then all 24 encodings vs the oracle
    c = [x for x in kats["naive_encode_kats"]["cases"] if x["name"] == "k45pu64"][0]
    w = ctx.encode_kmers(ctx.to_device(c["seq"].encode()), 1, c["K"], encs["ACGT"], 2)
    assert [int(x) for x in w.cpu().numpy().view(np.uint64)] == [int(x) for x in c["words"]]
    rcw = ctx.encoding_rev_comp(w, c["K"], encs["ACGT"], 2)
    assert ctx.encoding_decode(rcw, encs["ACGT"], 2).cpu().numpy().tobytes() == c["decode_rev_comp"].encode()
    assert ctx.encoding_decode(w, encs["ACGT"], 2).cpu().numpy().tobytes() == c["decode"].encode()
    for name, enc in encs.items():
        for K, B in ((2, 1), (15, 1), (31, 1), (32, 1), (33, 2), (45, 2), (63, 2), (64, 2), (100, 4)):
            n = 20
            seqs = bytes(rng.choice(b"ACGTacgtNnUu") for _ in range(n * K))
            w = ctx.encode_kmers(ctx.to_device(seqs), n, K, enc, B)
            wn = w.cpu().numpy().view(np.uint64).reshape(n, B)
            rcw = ctx.encoding_rev_comp(w, K, enc, B).cpu().numpy().view(np.uint64).reshape(n, B)
            dec = ctx.encoding_decode(w, enc, B).cpu().numpy().reshape(n, 32 * B)
            for i in range(n):
                arr = orc.naive_encode(enc, seqs[i * K:(i + 1) * K], 8 * B)
                assert orc.words(arr, 64) == [int(x) for x in wn[i]], (name, K)
                assert orc.words(orc.naive_rev_comp(enc, K, arr), 64) == [int(x) for x in rcw[i]], (name, K)
                assert orc.naive_decode(enc, arr) == dec[i].tobytes()
    # Xor10 == Naive::ACTG (xor10.rs:17-22 vs naive.rs:50)
    seq = b"TAAGGATTCTAATCATAAGGATTCTAATCATAAGGATTCTAATCA"
    w = ctx.encode_kmers(ctx.to_device(seq), 1, 45, encs["ACTG"], 2)
    assert [int(x) for x in w.cpu().numpy().view(np.uint64)] == orc.words(orc.xor10_encode(seq, 16), 64)
    # encode_windows == the benches' b.windows(K).map(Kmer::new) shape
    L, n, k = 80, 30, 31
    host = orc.gen_reads(9, 0, L * n)
    ww = ctx.encode_windows(ctx.to_device(host), n, L, k, encs["ACGT"], 1).cpu().numpy().view(np.uint64)
    fw, _, _, _ = orc.canonical_windows(host, n, L, k)
    assert (ww == fw).all()  # Naive::ACGT encode == naive_impl::Kmer::from on valid input


def test_encoding_errors(ctx):
    from kmers_amd import _lib

    seqs = ctx.to_device(b"A" * 200)
    with pytest.raises(_lib.KmxError) as e:
        ctx.encode_kmers(seqs, 1, 33, 0x1E, 1)
    assert e.value.status == _lib.E_TOO_LONG
    with pytest.raises(_lib.KmxError) as e:
        ctx.encode_kmers(seqs, 1, 10, 0x00, 1)  # not one of the 24 discriminants
    assert e.value.status == _lib.E_ARG
    w = ctx.encode_kmers(seqs, 1, 10, 0x1E, 1)
    with pytest.raises(_lib.KmxError) as e:
        ctx.encoding_rev_comp(w, 1, 0x1E, 1)  # K=1 underflows in the reference
    assert e.value.status == _lib.E_K_RANGE


# ---------------------------------------------------------------- SeqVector (packed 2-bit input)

def _acgt(rng, n):
    return np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, n)]


def test_seqvec_from_bytes_to_bytes_push(ctx, orc):
    rng = np.random.default_rng(21)
    for n in (1, 31, 32, 33, 64, 1000, 4097):
        host = _acgt(rng, n)
        sv = orc.SeqVector(host.tobytes())
        words = ctx.seqvec_from_bytes(ctx.to_device(host))
        assert (words.cpu().numpy().view(np.uint64) == sv.words[: (n + 31) // 32]).all(), n
        assert ctx.seqvec_to_bytes(words, n).cpu().numpy().tobytes() == host.tobytes().upper()
    # push_chars in pieces == one from(); seq_vector.rs:141-161
    import torch

    n = 700
    host = _acgt(rng, n)
    words = torch.zeros((n + 31) // 32, dtype=torch.int64, device=ctx.device)
    cuts = [0, 5, 37, 64, 70, 333, n]
    for lo, hi in zip(cuts, cuts[1:]):
        ctx.seqvec_push_chars(words, lo, ctx.to_device(host[lo:hi].copy()))
    assert (words.cpu().numpy().view(np.uint64) == orc.SeqVector(host.tobytes()).words[: (n + 31) // 32]).all()
    # strict like Kmer::from: first offending byte reported
    from kmers_amd import _lib
    bad = host.copy()
    bad[123] = ord("N")
    bad[500] = ord("x")
    with pytest.raises(Exception) as ei:
        ctx.seqvec_from_bytes(ctx.to_device(bad))
    assert ei.value.status == _lib.E_INVALID_BASE and ei.value.first_bad == 123


def test_seqvec_kats_on_device(ctx, orc, kats):
    sv_k = kats["seq_vector"]
    sw = sv_k["slice_words"]
    words = ctx.to_device(np.array(sw["words"], dtype=np.uint64))
    for g in sw["get_kmer_u64"]:
        got = ctx.seqvec_get_kmers(words, sw["len"], ctx.to_device(np.array([g["pos"]], dtype=np.uint64)), g["k"])
        assert int(got.cpu().numpy().view(np.uint64)[0]) == g["expect"]
    for e in sw["slice_equalities"]:
        a, b = e["slice"]
        it = ctx.seqvec_iter_kmers(words, sw["len"], e["k"], a, b).cpu().numpy().view(np.uint64)
        one = ctx.seqvec_get_kmers(words, sw["len"], ctx.to_device(np.array([e["same_as_pos"]], dtype=np.uint64)), e["k"])
        assert int(it[e["pos"]]) == int(one.cpu().numpy().view(np.uint64)[0])
    ik = sv_k["iter_kmers"]
    w = ctx.seqvec_from_bytes(ctx.to_device(ik["seq"].encode()))
    got = [orc.kmer_to_string(orc.lib().kmo_kmer_from_u64(int(x), ik["k"])) for x in ctx.seqvec_iter_kmers(w, len(ik["seq"]), ik["k"]).cpu().numpy().view(np.uint64)]
    assert got == ik["expect"]
    a, b = ik["slice"]
    got = [orc.kmer_to_string(orc.lib().kmo_kmer_from_u64(int(x), ik["k"])) for x in ctx.seqvec_iter_kmers(w, len(ik["seq"]), ik["k"], a, b).cpu().numpy().view(np.uint64)]
    assert got == ik["slice_expect"]


def test_seqvec_get_and_iter_random(ctx, orc):
    from kmers_amd import _lib
    rng = np.random.default_rng(22)
    n = 5000
    host = _acgt(rng, n)
    sv = orc.SeqVector(host.tobytes())
    words = ctx.seqvec_from_bytes(ctx.to_device(host))
    for k in (1, 7, 31, 32):
        pos = rng.integers(0, n - k + 1, 300).astype(np.uint64)
        pos[:3] = (0, n - k, 31)
        got = ctx.seqvec_get_kmers(words, n, ctx.to_device(pos), k).cpu().numpy().view(np.uint64)
        assert [int(x) for x in got] == [sv.get_kmer_u64(int(p), k) for p in pos]
        assert (ctx.seqvec_iter_kmers(words, n, k).cpu().numpy().view(np.uint64) == sv.iter_kmers(k)).all()
        assert (ctx.seqvec_iter_kmers(words, n, k, 77, 1234).cpu().numpy().view(np.uint64) == sv.iter_kmers(k, 77, 1234)).all()
    with pytest.raises(Exception) as ei:   # the reference asserts pos < len
        ctx.seqvec_get_kmers(words, n, ctx.to_device(np.array([5, n], dtype=np.uint64)), 3)
    assert ei.value.status == _lib.E_ARG
    assert ctx.seqvec_iter_kmers(words, n, 31, 10, 20).numel() == 0


@pytest.mark.parametrize("k", [31, 30, 21, 27, 16, 13, 11, 1])
@pytest.mark.parametrize("L,n", [(150, 64 * 50 + 9), (150, 63), (100, 64 * 20), (250, 64 * 11 + 3), (151, 64 * 7), (40, 500)])
def test_seqvec_canonical_reduce(ctx, orc, L, n, k):
    """reads stored back to back in a SeqVector: bit-sliced packed kernel (k in {21,31}) and the generic packed kernel
    against the oracle's get_kmer_u64 + to_canonical walk, and against the ASCII path on the same letters"""
    from kmers_amd import _lib
    rng = np.random.default_rng(L * 1000 + n + k)
    host = _acgt(rng, n * L)
    words = ctx.seqvec_from_bytes(ctx.to_device(host))
    if L < k:
        s = ctx.seqvec_canonical_reduce(words, n, L, k)
        assert s.n_valid == 0
        return
    o = orc.SeqVector(host.tobytes()).canonical_reduce(n, L, k, k)
    g = ctx.seqvec_canonical_reduce(words, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    assert (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw) == (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)
    a = ctx.canonical_reduce(ctx.to_device(host), n, L, k)
    g0 = ctx.seqvec_canonical_reduce(words, n, L, k)
    assert (g0.n_valid, g0.sum_canon, g0.xor_hash, g0.sum_fw) == (a.n_valid, a.sum_canon, 0, 0)


# ---------------------------------------------------------------- minimizers

def test_minimizer_kats_on_device(ctx, orc, kats):
    from kmers_amd import _lib
    for t in kats["minimizers"]["iter"]:
        s = t["seq"].encode()
        words = ctx.seqvec_from_bytes(ctx.to_device(s))
        hasher = _lib.HASH_LEX if t["hasher_k"] else _lib.HASH_IDENTITY
        mw, mp = ctx.seqvec_minimizers(words, 1, len(s), t["k"], t["w"], hasher, t["hasher_k"])
        got = [[int(a), int(b)] for a, b in zip(mw.cpu().numpy().view(np.uint64), mp.cpu().numpy())]
        assert got == t["expect"], t["name"]


@pytest.mark.parametrize("k,w,hk", [(31, 15, 15), (21, 11, 0), (9, 3, 32), (5, 5, 5), (32, 1, 1), (40, 7, 7)])
def test_seqvec_minimizers_vs_oracle(ctx, orc, k, w, hk):
    from kmers_amd import _lib
    rng = np.random.default_rng(k * 100 + w)
    L, n = 150, 257
    host = _acgt(rng, n * L)
    sv = orc.SeqVector(host.tobytes())
    words = ctx.seqvec_from_bytes(ctx.to_device(host))
    hasher = _lib.HASH_LEX if hk else _lib.HASH_IDENTITY
    mw, mp = ctx.seqvec_minimizers(words, n, L, k, w, hasher, hk)
    ow, op = orc.seqvec_minimizers(sv, n, L, k, w, hk)
    assert (mw.cpu().numpy().view(np.uint64) == ow).all()
    assert (mp.cpu().numpy().view(np.uint32) == op).all()


@pytest.mark.parametrize("L,k,w,hk", [(150, 31, 28, 28), (150, 31, 29, 29), (150, 31, 15, 6), (150, 31, 15, 0), (256, 31, 15, 15),
                                      (257, 31, 15, 15), (100, 21, 20, 20), (64, 33, 2, 1), (150, 32, 16, 16), (40, 40, 32, 3), (150, 31, 5, 5)])
def test_seqvec_minimizers_sliding_minimum(ctx, orc, L, k, w, hk):
    """the sliding-window-minimum kernel ((hash << 8) | position keys, doubling passes) and its limits: 56-bit hashes,
    L = 256, hashers shorter than the l-mer (many equal hashes: the leftmost must win), low-complexity reads"""
    from kmers_amd import _lib
    rng = np.random.default_rng(L * 1000 + k * 10 + w)
    n = 16 * 7 + 5
    host = _acgt(rng, n * L)
    host[: 20 * L] = np.frombuffer(b"ACAC", np.uint8)[rng.integers(0, 2, 20 * L) * 2]   # two-letter reads: ties everywhere
    host[20 * L: 24 * L] = ord("A")
    sv = orc.SeqVector(host.tobytes())
    words = ctx.seqvec_from_bytes(ctx.to_device(host))
    hasher = _lib.HASH_LEX if hk else _lib.HASH_IDENTITY
    mw, mp = ctx.seqvec_minimizers(words, n, L, k, w, hasher, hk)
    ow, op = orc.seqvec_minimizers(sv, n, L, k, w, hk)
    assert (mw.cpu().numpy().view(np.uint64) == ow).all()
    assert (mp.cpu().numpy().view(np.uint32) == op).all()


def test_minimizer_words_vs_oracle(ctx, orc):
    from kmers_amd import _lib
    rng = np.random.default_rng(77)
    for k, w, hk in ((31, 15, 15), (7, 1, 0), (7, 7, 7), (32, 9, 20), (12, 5, 0)):
        vals = rng.integers(0, 1 << 62, 500, dtype=np.uint64) & np.uint64((1 << (2 * k)) - 1 if k < 32 else (1 << 64) - 1)
        hasher = _lib.HASH_LEX if hk else _lib.HASH_IDENTITY
        mm, off = ctx.minimizer_words(ctx.to_device(vals), k, w, hasher, hk)
        exp = [orc.minimizer_word(int(v), k, w, hk) for v in vals]
        assert [int(x) for x in mm.cpu().numpy().view(np.uint64)] == [e[0] for e in exp]
        assert [int(x) for x in off.cpu().numpy()] == [e[1] for e in exp]
    with pytest.raises(Exception) as ei:
        ctx.minimizer_words(ctx.to_device(np.zeros(1, np.uint64)), 5, 6, _lib.HASH_IDENTITY, 0)
    assert ei.value.status == _lib.E_K_RANGE


def test_seqvec_minimizers_long_and_short_reads(ctx, orc):
    """reads too long for the LDS-staged kernel take the direct one; reads exactly k long yield one minimizer"""
    from kmers_amd import _lib
    rng = np.random.default_rng(404)
    for L, n, k, w in ((7000, 3, 31, 15), (31, 200, 31, 15), (33, 130, 31, 31)):
        host = _acgt(rng, n * L)
        sv = orc.SeqVector(host.tobytes())
        words = ctx.seqvec_from_bytes(ctx.to_device(host))
        mw, mp = ctx.seqvec_minimizers(words, n, L, k, w, _lib.HASH_LEX, w)
        ow, op = orc.seqvec_minimizers(sv, n, L, k, w, w)
        assert (mw.cpu().numpy().view(np.uint64) == ow).all() and (mp.cpu().numpy().view(np.uint32) == op).all()
