"""Round-4 GPU parity tests (all through the C ABI, bit-exact against the CPU oracle):
  * kmx_canonical_reduce behind an offsets array from a base that is NOT 16-byte aligned (the device-gated
    uniform / ragged pair used to be entered with a base only the uniform launcher takes, and the generic kernel then counted
    the batch a second time);
  * kmx_canonical_windows2 with a caller's win_offsets on uniform reads (the tiled kernel ignored them);
  * the work-buffer ABI (kmx_ctx_set_work_buffer_limit / kmx_ctx_work_buffer_info) that replaced two environment knobs;
  * pass 2 of the bit-sliced scan on the matrix pipe: inputs that put extreme counts into its fp32 accumulators
    (every window canonical-forward on every read, every read identical) and every k / frame it is instantiated for."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from kmers_amd.api import Context

    c = Context()
    yield c
    c.close()


def _dirty(rng, n, p_bad):
    alpha = np.frombuffer(b"ACGTacgt", np.uint8)
    a = alpha[rng.integers(0, 8, n)].copy()
    bad = rng.random(n) < p_bad
    a[bad] = rng.integers(0, 256, int(bad.sum()), dtype=np.uint8)
    return a


def _same(g, o, want_hash=True):
    assert g.n_valid == o.n_valid
    assert g.sum_canon == o.sum_canon
    assert g.xor_hash == (o.xor_hash if want_hash else 0)


# ------------------------------------------------------------------ offsets + a misaligned base

@pytest.mark.parametrize("k", [13, 31])
@pytest.mark.parametrize("lead", [1, 7, 8, 15])
@pytest.mark.parametrize("case", ["uniform", "one_trimmed"])
def test_reduce_offsets_from_a_misaligned_base(ctx, orc, k, lead, case):
    from kmers_amd import _lib

    L, n_reads = 150, 64 * 50 + 11
    rng = np.random.default_rng(1000 * k + lead)
    lens = np.full(n_reads, L, np.int64)
    if case == "one_trimmed":
        lens[n_reads // 3] = L - 9
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, lead + int(offsets[-1]), 0.0005)
    dev = ctx.to_device(host)
    o = orc.canonical_reduce(host[lead:], n_reads, 0, k, hasher_k=k, offsets=offsets)
    d_off = ctx.to_device(offsets)
    for hint in (L, 0, 160):
        g = ctx.canonical_reduce(dev[lead:], n_reads, hint, k, _lib.HASH_LEX, k, 0, offsets=d_off)
        _same(g, o)
    # and the summary is not doubled on the second call either
    g = ctx.canonical_reduce(dev[lead:], n_reads, L, k, _lib.HASH_NONE, 0, 0, offsets=d_off)
    _same(g, o, False)


# ------------------------------------------------------------------ windows2: a caller's win_offsets on uniform reads

@pytest.mark.parametrize("k", [33, 63])
def test_windows2_honours_win_offsets_on_uniform_reads(ctx, orc, k):
    import torch
    from kmers_amd.api import _ptr

    L, n_reads, gap = 150, 64 * 3 + 5, 5
    W = L - k + 1
    rng = np.random.default_rng(k)
    host = _dirty(rng, n_reads * L, 0.001)
    bases = ctx.to_device(host)
    dense = ctx.canonical_windows2(bases, n_reads, L, k)
    wo = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(W + gap))
    d_wo = ctx.to_device(wo)
    total = int(wo[-1])
    outs = {n: torch.full((2 * total,), -1, dtype=torch.int64, device=bases.device) for n in ("fw", "rc", "canon")}
    flags = torch.full((total,), 0x7F, dtype=torch.uint8, device=bases.device)
    r = ctx._reads(bases, n_reads, L, None)
    ctx._ck(ctx.lib.kmx_canonical_windows2(ctx._h, C.byref(r), _ptr(d_wo), k, _ptr(outs["fw"]), _ptr(outs["rc"]), _ptr(outs["canon"]), _ptr(flags)))
    ctx.synchronize()
    for name in ("fw", "rc", "canon"):
        got = outs[name].cpu().numpy().reshape(total, 2)
        want = dense[name].cpu().numpy().reshape(n_reads * W, 2)
        for rd in (0, 1, n_reads // 2, n_reads - 1):
            assert (got[rd * (W + gap): rd * (W + gap) + W] == want[rd * W: (rd + 1) * W]).all(), (name, rd)
            assert (got[rd * (W + gap) + W: (rd + 1) * (W + gap)] == -1).all()      # the gap is the caller's: untouched
    gf = flags.cpu().numpy()
    df = dense["flags"].cpu().numpy()
    for rd in (0, n_reads - 1):
        assert (gf[rd * (W + gap): rd * (W + gap) + W] == df[rd * W: (rd + 1) * W]).all()


# ------------------------------------------------------------------ the work buffer as ABI

def test_work_buffer_limit_and_info(ctx, orc):
    """a small limit forces the chunked histogram (several chunks, same table); 0 restores the automatic budget; the context
    reports what it holds and how often it allocated"""
    L, k, b, n_reads = 150, 31, 18, 64 * 600 + 7
    rng = np.random.default_rng(5)
    host = _dirty(rng, n_reads * L, 0.0003)
    bases = ctx.to_device(host)
    o = orc.histogram(host, n_reads, L, k, k, b)
    held0, n0 = ctx.work_buffer_info()
    ctx.set_work_buffer_limit(8 << 20)
    try:
        g = ctx.histogram(bases, n_reads, L, k, 1, k, b)
        assert (g.cpu().numpy().view(np.uint64) == o).all()
        held1, n1 = ctx.work_buffer_info()
        assert held1 >= held0 and n1 >= n0
        g = ctx.histogram(bases, n_reads, L, k, 1, k, b)
        assert ctx.work_buffer_info()[1] == n1          # the second call re-uses the buffer
        assert (g.cpu().numpy().view(np.uint64) == o).all()
    finally:
        ctx.set_work_buffer_limit(0)
    g = ctx.histogram(bases, n_reads, L, k, 1, k, b)
    assert (g.cpu().numpy().view(np.uint64) == o).all()
    lib = ctx.lib
    assert lib.kmx_ctx_set_work_buffer_limit(None, 0) != 0 and lib.kmx_ctx_work_buffer_info(None, None, None) != 0


# ------------------------------------------------------------------ pass 2 on the matrix pipe: extreme accumulator contents

@pytest.mark.parametrize("k", [13, 17, 18, 21, 31])
@pytest.mark.parametrize("pattern", ["all_A", "all_T", "identical_reads", "alternating_AT", "poly_ACGT"])
def test_scan_extreme_counts(ctx, orc, k, pattern):
    """every accumulator entry of a window block gets the same sign of contribution from all 64 reads of every tile: all-A reads
    (fw < rc in every window, every plane zero), all-T (never), 64 identical random reads per tile, and periodic reads whose
    windows tie (fw == rc never happens for odd k; even k is not instantiated below 33 ... 64 either way the iterator's rule
    `fw < rc` decides)"""
    from kmers_amd import _lib

    L, n_reads = 150, 64 * 40 + 9
    rng = np.random.default_rng(k)
    if pattern == "all_A":
        host = np.full(n_reads * L, ord("A"), np.uint8)
    elif pattern == "all_T":
        host = np.full(n_reads * L, ord("T"), np.uint8)
    elif pattern == "identical_reads":
        one = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
        host = np.tile(one, n_reads)
    elif pattern == "alternating_AT":
        host = np.tile(np.frombuffer(b"AT", np.uint8), n_reads * L // 2)
    else:
        host = np.tile(np.frombuffer(b"ACGT", np.uint8), (n_reads * L + 3) // 4)[: n_reads * L].copy()
    o = orc.canonical_reduce(host, n_reads, L, k, hasher_k=k)
    g = ctx.canonical_reduce(ctx.to_device(host), n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _same(g, o)
    assert g.sum_fw == o.sum_fw


@pytest.mark.parametrize("k", [33, 34, 48, 49, 50, 63, 64])
@pytest.mark.parametrize("pattern", ["all_A", "identical_reads", "random"])
def test_scan2_extreme_counts(ctx, orc, k, pattern):
    L, n_reads = 150, 64 * 30 + 3
    rng = np.random.default_rng(k)
    if pattern == "all_A":
        host = np.full(n_reads * L, ord("A"), np.uint8)
    elif pattern == "identical_reads":
        host = np.tile(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)], n_reads)
    else:
        host = _dirty(rng, n_reads * L, 0.0005)
    o = orc.canonical_reduce2(host, n_reads, L, k, with_hash=True)
    g = ctx.canonical_reduce2(ctx.to_device(host), n_reads, L, k, with_hash=True)
    assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


@pytest.mark.parametrize("L", [36, 50, 64, 75, 80, 81, 100, 112, 113, 128, 129, 150, 160, 161, 200, 208, 209, 224, 250, 256])
def test_scan_every_frame(ctx, orc, L):
    """the six frames (5, 7, 8, 10, 13, 16 words) at the ends of their length ranges, k = 31 and 21, dirty"""
    from kmers_amd import _lib

    n_reads = 64 * 20 + 33
    for k in (31, 21):
        if L < k:
            continue
        rng = np.random.default_rng(L * 100 + k)
        host = _dirty(rng, n_reads * L, 0.0004)
        o = orc.canonical_reduce(host, n_reads, L, k, hasher_k=k)
        g = ctx.canonical_reduce(ctx.to_device(host), n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _same(g, o)
        assert g.sum_fw == o.sum_fw


# ------------------------------------------------------------------ two-word k on the 13- and 16-word frames

@pytest.mark.parametrize("k", [33, 47, 48, 49, 63, 64])
@pytest.mark.parametrize("L", [161, 176, 200, 208, 209, 224, 250, 256])
def test_reduce2_on_the_long_frames(ctx, orc, k, L):
    """kmx_canonical_reduce2 on reads of 161..256 bases took the lane-per-read kernel (0.4 TB/s) until round 4; now the bit-sliced
    kernel's 13- / 16-word frames with 6 / 7 windows per lane (window blocks past the read's last window are skipped)"""
    n_reads = 64 * 9 + 21
    rng = np.random.default_rng(k * 1000 + L)
    host = _dirty(rng, n_reads * L, 0.0004)
    o = orc.canonical_reduce2(host, n_reads, L, k, with_hash=True)
    g = ctx.canonical_reduce2(ctx.to_device(host), n_reads, L, k, with_hash=True)
    assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


@pytest.mark.parametrize("lead", [3, 8])
def test_reduce2_long_frames_unaligned_base(ctx, orc, lead):
    k, L, n_reads = 63, 250, 64 * 5 + 2
    rng = np.random.default_rng(lead)
    host = _dirty(rng, lead + n_reads * L, 0.0005)
    dev = ctx.to_device(host)
    o = orc.canonical_reduce2(host[lead:], n_reads, L, k, with_hash=True)
    g = ctx.canonical_reduce2(dev[lead:], n_reads, L, k, with_hash=True)
    assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


# ------------------------------------------------------------------ long ragged reads: segments built on the device

@pytest.mark.parametrize("k", [13, 21, 31])
@pytest.mark.parametrize("case", ["long", "mixed", "few_huge", "with_empty"])
def test_reduce_long_ragged_reads(ctx, orc, k, case):
    """a length bound above 256 = "long reads": every read is cut into overlapping segments on the device (kmx_segments.hip) and
    the ragged bit-sliced kernel scans those; the summary is the per-read iterator's"""
    from kmers_amd import _lib

    rng = np.random.default_rng(k * 31 + len(case))
    if case == "long":
        lens = rng.integers(300, 5000, 700)
    elif case == "mixed":
        lens = np.where(rng.random(3000) < 0.3, rng.integers(257, 3000, 3000), rng.integers(0, 257, 3000))
    elif case == "few_huge":
        lens = np.array([250_000, 17, 131_313, k, k - 1, 90_001])
    else:
        lens = rng.integers(200, 2000, 900)
        lens[rng.integers(0, 900, 60)] = 0
        lens[rng.integers(0, 900, 60)] = k - 1
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n_reads = len(lens)
    host = _dirty(rng, int(offsets[-1]), 0.0002)
    o = orc.canonical_reduce(host, n_reads, 0, k, hasher_k=k, offsets=offsets)
    dev, d_off = ctx.to_device(host), ctx.to_device(offsets)
    for _ in range(2):   # (twice: the work buffer and the masks of the dirty reads are re-used)
        g = ctx.canonical_reduce(dev, n_reads, 1 << 20, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW, offsets=d_off)
        _same(g, o)
        assert g.sum_fw == o.sum_fw
    g = ctx.canonical_reduce(dev, n_reads, 300, k, _lib.HASH_NONE, 0, 0, offsets=d_off)
    _same(g, o, False)
    # and from an offsets array that does not start at 0 (a slice of a larger batch)
    cut = n_reads // 3
    o2 = orc.canonical_reduce(host, n_reads - cut, 0, k, hasher_k=k, offsets=offsets[cut:])
    g2 = ctx.canonical_reduce(dev, n_reads - cut, 100_000_000, k, _lib.HASH_LEX, k, 0, offsets=d_off[cut:])
    _same(g2, o2)


@pytest.mark.parametrize("L,n", [(1000, 150_000), (300, 400_000)])
def test_long_uniform_reads_at_size(ctx, orc, L, n):
    """uniform reads longer than a frame, enough of them that every wave scans several tiles of segments: a three-wave build of the
    ragged kernel (68 bytes of spills) returned a wrong sum_canon here -- n_valid and xor_hash right -- while every smaller
    test passed (round 4; the ragged variants are compiled without spills since)"""
    from kmers_amd import _lib

    k = 31
    bases = ctx.gen_reads(n * L, first_byte=L)
    host = bases.cpu().numpy()
    o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
    for hasher in (_lib.HASH_NONE, _lib.HASH_LEX):
        g = ctx.canonical_reduce(bases, n, L, k, hasher, k if hasher else 0, 0)
        _same(g, o, hasher != _lib.HASH_NONE)
    # the same bytes as ragged reads of 2 L / L bases behind an offsets array with a bound above 256 (segments built on the device)
    lens = np.where(np.arange(n // 2) % 2 == 0, 2 * L, 0) + np.where(np.arange(n // 2) % 2 == 1, 2 * L, 0)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    o2 = orc.canonical_reduce(host, len(lens), 0, k, hasher_k=k, offsets=offsets)
    g2 = ctx.canonical_reduce(bases, len(lens), 1 << 16, k, _lib.HASH_LEX, k, 0, offsets=ctx.to_device(offsets))
    _same(g2, o2)


@pytest.mark.parametrize("k,L,n", [(63, 150, 1_500_000), (50, 150, 1_500_000), (64, 120, 1_500_000), (31, 256, 900_000), (21, 220, 1_000_000), (18, 250, 900_000)])
def test_variants_with_spills_at_size(ctx, orc, k, L, n):
    """the instantiations hipcc compiles with a few spilled registers (the two-word k from 50 up at three waves per SIMD, the
    16-word frame), at a size where every wave scans many tiles, against the oracle (see test_long_uniform_reads_at_size)"""
    from kmers_amd import _lib

    bases = ctx.gen_reads(n * L, first_byte=7 * k)
    host = bases.cpu().numpy()
    if k <= 31:
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _same(g, o)
        assert g.sum_fw == o.sum_fw
    else:
        o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
        g = ctx.canonical_reduce2(bases, n, L, k, with_hash=True)
        assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


# ------------------------------------------------------------------ uniform reads above 256 bases: segments on the uniform kernel

@pytest.mark.parametrize("k", [13, 21, 30, 31])
@pytest.mark.parametrize("L,n", [(257, 64 * 9 + 5), (270, 64 * 3 + 7), (287, 64 * 4), (300, 64 * 6 + 1), (383, 200), (400, 64 * 5 + 2), (450, 64 * 2 + 9), (1000, 64 * 3 + 1), (1021, 130), (5003, 150), (20000, 70), (100_003, 9)])
def test_long_uniform_reads_as_segments(ctx, orc, k, L, n):
    """round 4: a read of L > 256 bases is cut into J segments of T or T - 1 windows (bs_seg_plan) that the UNIFORM kernel scans
    (a short segment's last window masked, the closed form corrected by the short segments' plane totals); dirty bytes, the
    hash fold, sum_fw, tiles that end inside a read, a final partial tile of segments"""
    from kmers_amd import _lib

    rng = np.random.default_rng(k * 1000 + L)
    host = _dirty(rng, n * L, 0.0003)
    o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
    dev = ctx.to_device(host)
    for _ in range(2):
        g = ctx.canonical_reduce(dev, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _same(g, o)
        assert g.sum_fw == o.sum_fw
    g = ctx.canonical_reduce(dev, n, L, k, _lib.HASH_IDENTITY, 0, 0)
    assert (g.n_valid, g.sum_canon) == (o.n_valid, o.sum_canon)


@pytest.mark.parametrize("k", [33, 47, 48, 63, 64])
@pytest.mark.parametrize("L,n", [(257, 64 * 5 + 3), (300, 64 * 6 + 1), (1000, 64 * 3 + 1), (5003, 90), (20000, 40)])
def test_long_uniform_reads_as_segments_two_word(ctx, orc, k, L, n):
    rng = np.random.default_rng(k * 1000 + L)
    host = _dirty(rng, n * L, 0.0003)
    o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
    dev = ctx.to_device(host)
    for _ in range(2):
        g = ctx.canonical_reduce2(dev, n, L, k, with_hash=True)
        assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


@pytest.mark.parametrize("k,L,n", [(31, 1000, 150_000), (31, 300, 400_000), (31, 400, 300_000), (63, 1000, 100_000), (21, 777, 150_000), (13, 10_000, 15_000)])
def test_long_uniform_segments_at_size(ctx, orc, k, L, n):
    """the same at sizes where every wave scans many tiles of segments (the accumulators and the short-segment totals carry over)"""
    from kmers_amd import _lib

    bases = ctx.gen_reads(n * L, first_byte=3 * L)
    host = bases.cpu().numpy()
    if k <= 31:
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _same(g, o)
        assert g.sum_fw == o.sum_fw
    else:
        o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
        g = ctx.canonical_reduce2(bases, n, L, k, with_hash=True)
        assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


# ---------------------------------------------------------------- materialise: the output line two reads share

@pytest.mark.parametrize("p_bad", [0.0, 0.001])
@pytest.mark.parametrize("L,k", [(52, 21), (53, 21), (61, 31), (67, 21), (100, 21), (129, 21), (140, 31), (150, 21), (151, 21), (155, 21),
                                 (158, 21), (160, 21), (160, 31), (150, 5), (45, 13), (51, 21)])
def test_windows_single_array_shared_lines(ctx, orc, L, k, p_bad):
    """one u64 array of uniform reads whose window count is not a multiple of 16: the output line that two neighbouring reads
    of a tile share is written once, whole -- the head of the second read is rebuilt after the tile's last block
    (kmx_scan.hip: heads_done).  Every residue of W mod 16, W around 32 (below it the pieces go out where they fall), tiles
    with and without an invalid byte, a partial last tile; canonical_kmer_iterator.rs:42-70 per read."""
    rng = np.random.default_rng(L * 1000 + k)
    n = 64 * 7 + 29
    host = _dirty(rng, n * L, p_bad)
    bases = ctx.to_device(host)
    fw, rc, canon, flags = orc.canonical_windows(host, n, L, k)
    for name, exp in (("canon", canon), ("fw", fw), ("rc", rc)):
        got = ctx.canonical_windows(bases, n, L, k, want=(name,))[name].cpu().numpy().view(np.uint64)
        assert (got == exp).all(), (name, int((got != exp).sum()), np.flatnonzero(got != exp)[:8])
    # several arrays: one pass of the same kernel per array of words, the flags from their own kernel (two bits per window in
    # registers, the tile's bytes written 16 per lane)
    for want in (("fw", "rc", "canon", "flags"), ("canon", "flags"), ("flags",), ("fw", "canon")):
        got = ctx.canonical_windows(bases, n, L, k, want=want)
        for name, exp in (("canon", canon), ("fw", fw), ("rc", rc)):
            if name in want:
                assert (got[name].cpu().numpy().view(np.uint64) == exp).all(), (want, name)
        if "flags" in want:
            gf = got["flags"].cpu().numpy()
            assert (gf == flags).all(), (want, int((gf != flags).sum()), np.flatnonzero(gf != flags)[:8])


# ---------------------------------------------------------------- two-word k behind an offsets array: uniform or not, decided on the device

@pytest.mark.parametrize("k", [33, 47, 63, 64])
@pytest.mark.parametrize("L", [150, 250])
@pytest.mark.parametrize("case", ["uniform", "one_trimmed", "last_trimmed", "no_bound"])
def test_reduce2_offsets_picks_the_uniform_kernel_on_the_device(ctx, orc, k, case, L):
    """kmx_canonical_reduce2 on reads behind an offsets array with a length bound (what kmx_fastx_parse hands over): a small
    kernel checks offsets[i] == i * L, the tiled uniform kernel and the lane-per-read kernel are launched behind its verdict and
    exactly one of them counts (kmer.rs:21-28,67-69 per read either way).  Twice in a row: the gate is never left armed."""
    rng = np.random.default_rng(k * 7 + len(case) + L)
    n = 64 * 11 + 23
    lens = np.full(n, L)
    if case == "one_trimmed":
        lens[n // 3] = 97
    if case == "last_trimmed":
        lens[-1] = L - 1
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]), 0.0004)
    o = orc.canonical_reduce2(host, n, 0, k, with_hash=True, offsets=offsets)
    dev, d_off = ctx.to_device(host), ctx.to_device(offsets)
    hint = 0 if case == "no_bound" else L
    for _ in range(2):
        g = ctx.canonical_reduce2(dev, n, hint, k, with_hash=True, offsets=d_off)
        assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)
    # and a plain uniform call right after (the gate word is shared with it)
    if case == "uniform":
        u = ctx.canonical_reduce2(dev, n, L, k, with_hash=True)
        assert (u.n_valid, u.sum_lo, u.sum_hi, u.xor_lo, u.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


# ---------------------------------------------------------------- two-word k on truly ragged reads: the ragged bit-sliced kernel

@pytest.mark.parametrize("k", [33, 40, 47, 48, 49, 50, 56, 63, 64])
@pytest.mark.parametrize("case", ["trimmed", "mix_100_160", "no_bound_some_long", "short_and_empty", "bound_above_frame"])
def test_reduce2_ragged_reads_tiled(ctx, orc, k, case):
    """kmx_canonical_reduce2 on reads of unequal length behind an offsets array: the ragged bit-sliced kernel (validity planes,
    read-end planes, the closed form over two words) in the 10-word frame -- a bound of at most 160 bases or none; tiles holding
    a longer read roll per lane, a bound above the frame leaves the call to the lane-per-read kernel.  Dirty bytes, reads
    shorter than k, empty reads, a partial last tile (kmer.rs:21-28,67-69 per read)."""
    rng = np.random.default_rng(k * 13 + len(case))
    n = 64 * 9 + 31
    if case == "trimmed":
        lens, hint = np.where(rng.random(n) < 0.05, rng.integers(36, 150, n), 150), 150
    elif case == "mix_100_160":
        lens, hint = rng.integers(100, 161, n), 160
    elif case == "no_bound_some_long":
        lens, hint = np.where(rng.random(n) < 0.02, rng.integers(161, 400, n), rng.integers(60, 161, n)), 0
    elif case == "short_and_empty":
        lens, hint = rng.integers(0, 2 * k, n), 2 * k
    else:
        lens, hint = rng.integers(100, 251, n), 250
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]), 0.0005)
    o = orc.canonical_reduce2(host, n, 0, k, with_hash=True, offsets=offsets)
    dev, d_off = ctx.to_device(host), ctx.to_device(offsets)
    for with_hash in (True, False, True):
        g = ctx.canonical_reduce2(dev, n, hint, k, with_hash=with_hash, offsets=d_off)
        assert (g.n_valid, g.sum_lo, g.sum_hi) == (o.n_valid, o.sum_lo, o.sum_hi), (case, k)
        assert (g.xor_lo, g.xor_hi) == ((o.xor_lo, o.xor_hi) if with_hash else (0, 0))


@pytest.mark.parametrize("k,n", [(63, 400_000), (33, 400_000), (50, 300_000)])
def test_reduce2_ragged_reads_at_size(ctx, orc, k, n):
    """the same where every wave scans many tiles (the accumulators fold, the per-wave sums carry over)"""
    rng = np.random.default_rng(k)
    lens = np.where(rng.random(n) < 0.02, rng.integers(36, 150, n), 150)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = ctx.gen_reads(int(offsets[-1]), first_byte=7)
    host = bases.cpu().numpy()
    o = orc.canonical_reduce2(host, n, 0, k, with_hash=True, offsets=offsets)
    g = ctx.canonical_reduce2(bases, n, 150, k, with_hash=True, offsets=ctx.to_device(offsets))
    assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


@pytest.mark.parametrize("k", [33, 48, 63, 64])
@pytest.mark.parametrize("case", ["long", "mixed", "few_huge", "with_empty"])
def test_reduce2_long_ragged_reads(ctx, orc, k, case):
    """two-word k, a length bound above 256: segments of at most 161 - k windows cut on the device, scanned by the two-word ragged
    bit-sliced kernel (kmer.rs:21-28,67-69 per read)"""
    rng = np.random.default_rng(k * 37 + len(case))
    if case == "long":
        lens = rng.integers(300, 5000, 600)
    elif case == "mixed":
        lens = np.where(rng.random(2500) < 0.3, rng.integers(257, 3000, 2500), rng.integers(0, 257, 2500))
    elif case == "few_huge":
        lens = np.array([250_000, 17, 131_313, k, k - 1, 90_001])
    else:
        lens = rng.integers(200, 2000, 800)
        lens[rng.integers(0, 800, 60)] = 0
        lens[rng.integers(0, 800, 60)] = k - 1
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n_reads = len(lens)
    host = _dirty(rng, int(offsets[-1]), 0.0002)
    o = orc.canonical_reduce2(host, n_reads, 0, k, with_hash=True, offsets=offsets)
    dev, d_off = ctx.to_device(host), ctx.to_device(offsets)
    for bound in (1 << 20, 300):
        g = ctx.canonical_reduce2(dev, n_reads, bound, k, with_hash=True, offsets=d_off)
        assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi), (case, k, bound)
    cut = n_reads // 3
    o2 = orc.canonical_reduce2(host, n_reads - cut, 0, k, with_hash=True, offsets=offsets[cut:])
    g2 = ctx.canonical_reduce2(dev, n_reads - cut, 100_000_000, k, with_hash=True, offsets=d_off[cut:])
    assert (g2.n_valid, g2.sum_lo, g2.sum_hi, g2.xor_lo, g2.xor_hi) == (o2.n_valid, o2.sum_lo, o2.sum_hi, o2.xor_lo, o2.xor_hi)


# ---------------------------------------------------------------- two-word materialise of ragged reads: tiled

@pytest.mark.parametrize("k", [33, 47, 63, 64])
@pytest.mark.parametrize("case", ["trimmed_150", "mix_100_160", "mix_to_250", "no_bound_some_long", "short_and_empty", "gaps"])
def test_windows2_ragged_reads_tiled(ctx, orc, k, case):
    """kmx_canonical_windows2 on reads behind an offsets array: the tiled kernel with a per-lane start, window count and first
    output slot (kmx_generic.hip: windows2_tiled_kernel<.., RAGGED>); a tile with a read above the bound, or an invalid byte,
    takes the per-read path inside it.  Every array alone (the staged whole-line write-back), all of them, flags alone;
    `gaps`: a caller's win_offsets that leave room between the reads -- what lies there stays untouched."""
    import torch
    from kmers_amd.api import _ptr

    rng = np.random.default_rng(k * 17 + len(case))
    n = 64 * 7 + 19
    gap = 0
    if case == "trimmed_150":
        lens, hint = np.where(rng.random(n) < 0.05, rng.integers(36, 150, n), 150), 150
    elif case == "mix_100_160":
        lens, hint = rng.integers(100, 161, n), 160
    elif case == "mix_to_250":
        lens, hint = rng.integers(80, 251, n), 250
    elif case == "no_bound_some_long":
        lens, hint = np.where(rng.random(n) < 0.02, rng.integers(257, 600, n), rng.integers(60, 257, n)), 0
    elif case == "short_and_empty":
        lens, hint = rng.integers(0, 2 * k, n), 2 * k
    else:
        lens, hint, gap = rng.integers(100, 151, n), 150, 3
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nw = np.maximum(lens - k + 1, 0)
    wo = np.concatenate([[0], np.cumsum(nw + gap)]).astype(np.uint64)
    total = int(wo[-1])
    host = _dirty(rng, int(offsets[-1]), 0.0003)
    fw, rc, canon, flags = orc.canonical_windows2(host, n, 0, k, offsets=offsets)       # dense, read after read
    dense_at = np.concatenate([[0], np.cumsum(nw)])
    keep = np.concatenate([np.arange(wo[i], wo[i] + nw[i], dtype=np.int64) for i in range(n)]) if total else np.zeros(0, np.int64)
    dev, d_off, d_wo = ctx.to_device(host), ctx.to_device(offsets), ctx.to_device(wo)
    r = ctx._reads(dev, n, hint, d_off)
    exp = {"fw": fw, "rc": rc, "canon": canon}
    for want in (("canon",), ("fw",), ("rc",), ("fw", "rc", "canon", "flags"), ("flags",)):
        outs = {nm: torch.full((2 * total,), -1, dtype=torch.int64, device=dev.device) for nm in ("fw", "rc", "canon") if nm in want}
        fl = torch.full((total,), 0x7F, dtype=torch.uint8, device=dev.device) if "flags" in want else None
        ctx._ck(ctx.lib.kmx_canonical_windows2(ctx._h, C.byref(r), _ptr(d_wo), k, _ptr(outs.get("fw")), _ptr(outs.get("rc")), _ptr(outs.get("canon")), _ptr(fl)))
        ctx.synchronize()
        for nm, t in outs.items():
            got = t.cpu().numpy().view(np.uint64).reshape(total, 2)
            assert (got[keep] == exp[nm].reshape(-1, 2)).all(), (case, k, want, nm, int((got[keep] != exp[nm].reshape(-1, 2)).any(axis=1).sum()))
            if gap:
                mask = np.ones(total, bool); mask[keep] = False
                assert (t.cpu().numpy().reshape(total, 2)[mask] == -1).all(), "the gaps are the caller's"
        if fl is not None:
            gf = fl.cpu().numpy()
            assert (gf[keep] == flags).all(), (case, k, want)
    assert dense_at[-1] == len(flags)


# ---------------------------------------------------------------- single-word materialise of ragged reads, one array: whole lines

@pytest.mark.parametrize("k", [13, 21, 31])
@pytest.mark.parametrize("case", ["trimmed_150", "mix_100_160", "mix_to_250", "no_bound_some_long", "short_and_empty", "one_huge"])
def test_windows_ragged_single_array_ring(ctx, orc, k, case):
    """kmx_canonical_windows on reads behind an offsets array, ONE u64 array and no flags: the line-aligned ring with each read's
    line shift taken from its own first output slot and its window count from the offsets (SinkWindowsT<true, true>); a read's
    last pass writes what is left of it.  Tiles with an invalid byte or a read above the bound roll per lane through the same
    ring (canonical_kmer_iterator.rs:42-70 per read)."""
    rng = np.random.default_rng(k * 19 + len(case))
    n = 64 * 8 + 37
    if case == "trimmed_150":
        lens, hint = np.where(rng.random(n) < 0.05, rng.integers(36, 150, n), 150), 150
    elif case == "mix_100_160":
        lens, hint = rng.integers(100, 161, n), 160
    elif case == "mix_to_250":
        lens, hint = rng.integers(80, 251, n), 250
    elif case == "no_bound_some_long":
        lens, hint = np.where(rng.random(n) < 0.02, rng.integers(257, 600, n), rng.integers(60, 257, n)), 0
    elif case == "one_huge":      # (a rolled tile whose slots span far more than a frame's worth)
        lens, hint = rng.integers(100, 151, n), 150
        lens[70] = 70_000
    else:
        lens, hint = rng.integers(0, 3 * k, n), 3 * k
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]), 0.0003)
    fw, rc, canon, flags = orc.canonical_windows(host, n, 0, k, offsets=offsets)
    dev, d_off = ctx.to_device(host), ctx.to_device(offsets)
    for name, exp in (("canon", canon), ("fw", fw), ("rc", rc)):
        got = ctx.canonical_windows(dev, n, hint, k, offsets=d_off, host_offsets=offsets, want=(name,))[name].cpu().numpy().view(np.uint64)
        assert (got == exp).all(), (case, k, name, int((got != exp).sum()), np.flatnonzero(got != exp)[:8])


@pytest.mark.parametrize("lead", [1, 5, 8, 15])
@pytest.mark.parametrize("L,k", [(150, 21), (150, 31), (100, 13), (200, 31), (250, 27)])
def test_windows_uniform_from_a_misaligned_base(ctx, orc, lead, L, k):
    """kmx_canonical_windows on uniform reads whose first byte is not 16-byte aligned: the line-aligned kernel streams from the
    aligned address below (a tile spans one more chunk; the next tile's prefetch starts there too), one array and all four"""
    rng = np.random.default_rng(lead * 1000 + L + k)
    n = 64 * 6 + 11
    host_all = _dirty(rng, n * L + 32, 0.0004)
    dev_all = ctx.to_device(host_all)
    host, dev = host_all[lead:lead + n * L], dev_all[lead:lead + n * L]
    fw, rc, canon, flags = orc.canonical_windows(host, n, L, k)
    got = ctx.canonical_windows(dev, n, L, k, want=("canon",))["canon"].cpu().numpy().view(np.uint64)
    assert (got == canon).all()
    outs = ctx.canonical_windows(dev, n, L, k)
    assert (outs["fw"].cpu().numpy().view(np.uint64) == fw).all() and (outs["rc"].cpu().numpy().view(np.uint64) == rc).all()
    assert (outs["canon"].cpu().numpy().view(np.uint64) == canon).all() and (outs["flags"].cpu().numpy() == flags).all()


# ---------------------------------------------------------------- materialise of uniform reads above 256 bases: segments

@pytest.mark.parametrize("k", [13, 21, 31])
@pytest.mark.parametrize("L,n", [(257, 64 * 3 + 5), (300, 200), (483, 70), (1000, 64 + 9), (5003, 21), (40_000, 3)])
def test_windows_long_uniform_reads_as_segments(ctx, orc, k, L, n):
    """kmx_canonical_windows on uniform reads longer than a frame (round 4): every read as segments of 257 - k windows -- a start,
    an end and a first output slot each, planned on the device -- that the ragged materialise kernels take as reads of their
    own (they overlap by k - 1 bases in memory; their windows are consecutive in the output).  One array (the ring) and all
    four (the staged write-back); dirty bytes; canonical_kmer_iterator.rs:42-70 per read."""
    rng = np.random.default_rng(k * 23 + L)
    host = _dirty(rng, n * L, 0.0002)
    dev = ctx.to_device(host)
    fw, rc, canon, flags = orc.canonical_windows(host, n, L, k)
    for name, exp in (("canon", canon), ("fw", fw)):
        got = ctx.canonical_windows(dev, n, L, k, want=(name,))[name].cpu().numpy().view(np.uint64)
        assert (got == exp).all(), (name, int((got != exp).sum()), np.flatnonzero(got != exp)[:8])
    outs = ctx.canonical_windows(dev, n, L, k)
    assert (outs["fw"].cpu().numpy().view(np.uint64) == fw).all() and (outs["rc"].cpu().numpy().view(np.uint64) == rc).all()
    assert (outs["canon"].cpu().numpy().view(np.uint64) == canon).all() and (outs["flags"].cpu().numpy() == flags).all()


@pytest.mark.parametrize("k", [33, 47, 64])
@pytest.mark.parametrize("L,n", [(257, 64 * 2 + 5), (300, 150), (1000, 64 + 9), (5003, 21), (40_000, 3)])
def test_windows2_long_uniform_reads_as_segments(ctx, orc, k, L, n):
    """kmx_canonical_windows2 on uniform reads longer than a frame: the same device-side plan (segments with a start, an end and a
    first output slot) through windows2_tiled_kernel<.., RAGGED> with its separate ends array; kmer.rs:21-28,67-69 per read"""
    rng = np.random.default_rng(k * 29 + L)
    host = _dirty(rng, n * L, 0.0002)
    dev = ctx.to_device(host)
    fw, rc, canon, flags = orc.canonical_windows2(host, n, L, k)
    outs = ctx.canonical_windows2(dev, n, L, k)
    for name, exp in (("fw", fw), ("rc", rc), ("canon", canon)):
        assert (outs[name].cpu().numpy().view(np.uint64).reshape(-1, 2) == exp.reshape(-1, 2)).all(), (name, k, L)
    assert (outs["flags"].cpu().numpy() == flags).all()


@pytest.mark.parametrize("k", [13, 21, 31])
@pytest.mark.parametrize("case", ["long", "mixed", "few_huge", "with_empty"])
def test_windows_long_ragged_reads_as_segments(ctx, orc, k, case):
    """kmx_canonical_windows on ragged reads with a length bound above 256: every read cut into segments on the device
    (kmx_segments.hip: a start, an end and a first output slot per segment), materialised by the ragged kernels as reads of their
    own; one array (the ring) and all four; an offsets array that does not start at 0"""
    rng = np.random.default_rng(k * 41 + len(case))
    if case == "long":
        lens = rng.integers(300, 4000, 300)
    elif case == "mixed":
        lens = np.where(rng.random(1500) < 0.3, rng.integers(257, 2000, 1500), rng.integers(0, 257, 1500))
    elif case == "few_huge":
        lens = np.array([120_000, 17, 61_313, k, k - 1, 40_001])
    else:
        lens = rng.integers(200, 1500, 500)
        lens[rng.integers(0, 500, 40)] = 0
        lens[rng.integers(0, 500, 40)] = k - 1
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = len(lens)
    host = _dirty(rng, int(offsets[-1]), 0.0002)
    fw, rc, canon, flags = orc.canonical_windows(host, n, 0, k, offsets=offsets)
    dev, d_off = ctx.to_device(host), ctx.to_device(offsets)
    got = ctx.canonical_windows(dev, n, 1 << 20, k, offsets=d_off, host_offsets=offsets, want=("canon",))["canon"].cpu().numpy().view(np.uint64)
    assert (got == canon).all(), (case, k, int((got != canon).sum()), np.flatnonzero(got != canon)[:8])
    outs = ctx.canonical_windows(dev, n, 300, k, offsets=d_off, host_offsets=offsets)
    assert (outs["fw"].cpu().numpy().view(np.uint64) == fw).all() and (outs["rc"].cpu().numpy().view(np.uint64) == rc).all()
    assert (outs["canon"].cpu().numpy().view(np.uint64) == canon).all() and (outs["flags"].cpu().numpy() == flags).all()


@pytest.mark.parametrize("k", [33, 64])
@pytest.mark.parametrize("case", ["long", "mixed", "few_huge"])
def test_windows2_long_ragged_reads_as_segments(ctx, orc, k, case):
    """kmx_canonical_windows2 on ragged reads with a length bound above 256: the same device-side segments (start, end, first
    output slot) through windows2_tiled_kernel<.., RAGGED>"""
    rng = np.random.default_rng(k * 43 + len(case))
    if case == "long":
        lens = rng.integers(300, 3000, 200)
    elif case == "mixed":
        lens = np.where(rng.random(900) < 0.3, rng.integers(257, 1500, 900), rng.integers(0, 257, 900))
    else:
        lens = np.array([60_000, 17, 31_313, k, k - 1, 20_001])
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = len(lens)
    host = _dirty(rng, int(offsets[-1]), 0.0002)
    fw, rc, canon, flags = orc.canonical_windows2(host, n, 0, k, offsets=offsets)
    outs = ctx.canonical_windows2(ctx.to_device(host), n, 1 << 20, k, offsets=ctx.to_device(offsets), host_offsets=offsets)
    for name, exp in (("fw", fw), ("rc", rc), ("canon", canon)):
        assert (outs[name].cpu().numpy().view(np.uint64).reshape(-1, 2) == exp.reshape(-1, 2)).all(), (name, k, case)
    assert (outs["flags"].cpu().numpy() == flags).all()


@pytest.mark.parametrize("k", [33, 63])
def test_reduce2_offsets_total_matches_the_bound_but_reads_differ(ctx, orc, k):
    """kmx_canonical_reduce2, a bound of 161...256: a batch whose bases add up to n_reads * bound is taken for untrimmed and the
    segments are not built -- here the total matches while two reads differ (one longer than the bound, one shorter): the
    device-side gate says "not uniform" and the lane-per-read kernel counts.  Still the per-read result."""
    rng = np.random.default_rng(k)
    n, L = 64 * 6 + 3, 250
    lens = np.full(n, L)
    lens[10] += 7
    lens[200] -= 7
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    assert int(offsets[-1]) == n * L
    host = _dirty(rng, int(offsets[-1]), 0.0003)
    o = orc.canonical_reduce2(host, n, 0, k, with_hash=True, offsets=offsets)
    g = ctx.canonical_reduce2(ctx.to_device(host), n, L, k, with_hash=True, offsets=ctx.to_device(offsets))
    assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)
