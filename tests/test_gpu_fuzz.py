"""Randomised parity sweep over the arguments that select kernels (k, read length / length bound, layout, dirt, hasher,
base alignment): kmx_canonical_reduce / kmx_canonical_reduce2 / kmx_histogram / kmx_canonical_windows / kmx_canonical_windows2
against the oracle.  Fixed seeds; every case
is small enough for the oracle, together they walk the frames (7 / 10 / 16 words, segments), the windows-per-lane
variants, the blanking of dirty reads and the fallbacks of each."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from kmers_amd.api import Context

    c = Context()
    yield c
    c.close()


def _bytes(rng, n, p_bad, lower):
    a = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).copy()
    if lower:
        m = rng.random(n) < 0.3
        a[m] |= 0x20
    if p_bad > 0 and n:
        m = rng.random(n) < p_bad
        a[m] = rng.choice(np.frombuffer(b"NnRY.-*\x00\xff@", dtype=np.uint8), size=int(m.sum()))
    return a


N_FUZZ = int(os.environ.get("KMX_FUZZ_N", 120))   # a one-off wider sweep: KMX_FUZZ_N=3000


@pytest.mark.parametrize("seed", range(N_FUZZ))
def test_reduce_random_configuration(ctx, orc, seed):
    from kmers_amd import _lib

    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([2, 5, 11, 12, 13, 14, 16, 17, 19, 21, 24, 27, 29, 30, 31]))
    layout = rng.choice(["uniform", "uniform_misaligned", "ragged", "ragged_hint", "ragged_tight", "uniform_behind_offsets"])
    p_bad = float(rng.choice([0.0, 0.0, 0.0005, 0.004, 0.05]))
    n = int(rng.choice([1, 63, 64, 65, 64 * 7 + 3, 64 * 23 + 41]))
    hasher, hk = [(_lib.HASH_NONE, 0), (_lib.HASH_LEX, k), (_lib.HASH_LEX, max(1, k - 3)), (_lib.HASH_IDENTITY, 0)][int(rng.integers(0, 4))]
    want_fw = bool(rng.integers(0, 2)) and layout.startswith("uniform")
    if layout.startswith("uniform"):
        L = int(rng.choice([k, k + 1, 36, 50, 64, 75, 100, 101, 111, 112, 113, 125, 150, 151, 160, 161, 200, 250, 256, 257, 300, 400, 1000]))
        L = max(L, k)
        lead = int(rng.integers(1, 16)) if layout == "uniform_misaligned" else 0
        raw = _bytes(rng, n * L + lead + 64, p_bad, bool(rng.integers(0, 2)))
        host = raw[lead: lead + n * L]
        d_all = ctx.to_device(raw)
        d = d_all[lead: lead + n * L]
        o = orc.canonical_reduce(host, n, L, k, hasher_k=hk if hasher == _lib.HASH_LEX else 0)
        g = ctx.canonical_reduce(d, n, L, k, hasher, hk, _lib.REDUCE_SUM_FW if want_fw else 0)
    elif layout == "uniform_behind_offsets":
        # (round 5) reads of ONE length behind an offsets array, with the bound equal to it, above it, or absent: the device-side gate
        # hands them to the uniform kernels (laid out for the bound, scanning with the length found); now and then one read differs
        L = int(rng.choice([k, 36, 50, 100, 111, 125, 150, 160, 200, 256]))
        L = max(L, k)
        hint = int(rng.choice([0, L, L + int(rng.integers(1, 40)), 160, 256]))
        lens = np.full(n, L, dtype=np.int64)
        if rng.integers(0, 4) == 0:
            lens[rng.integers(0, n)] = max(0, L - int(rng.integers(1, 20)))
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        host = _bytes(rng, int(offsets[-1]) + 16, p_bad, bool(rng.integers(0, 2)))[: int(offsets[-1])]
        o = orc.canonical_reduce(host, n, 0, k, hasher_k=hk if hasher == _lib.HASH_LEX else 0, offsets=offsets)
        g = ctx.canonical_reduce(ctx.to_device(host) if len(host) else ctx.to_device(np.zeros(16, np.uint8)), n, hint, k, hasher, hk, 0,
                                 offsets=ctx.to_device(offsets))
    else:
        top = int(rng.choice([40, 64, 100, 111, 112, 150, 160, 250]))
        lens = rng.integers(0, top + 1, size=n)
        if rng.integers(0, 2):
            lens[rng.integers(0, n)] = top + int(rng.integers(1, 200))      # one read past the bound
        hint = {"ragged": 0, "ragged_hint": max(top, 160), "ragged_tight": top}[layout]
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        host = _bytes(rng, int(offsets[-1]) + 16, p_bad, False)[: int(offsets[-1])]
        o = orc.canonical_reduce(host, n, 0, k, hasher_k=hk if hasher == _lib.HASH_LEX else 0, offsets=offsets)
        g = ctx.canonical_reduce(ctx.to_device(host) if len(host) else ctx.to_device(np.zeros(16, np.uint8)), n, hint, k, hasher, hk, 0,
                                 offsets=ctx.to_device(offsets))
    assert (g.n_valid, g.sum_canon) == (o.n_valid, o.sum_canon), (k, layout, n, p_bad)
    if want_fw:
        assert g.sum_fw == o.sum_fw
    if hasher == _lib.HASH_LEX:
        assert g.xor_hash == o.xor_hash, (k, layout, n, hk)


@pytest.mark.parametrize("seed", range(max(40, N_FUZZ // 3)))
def test_reduce2_and_histogram_random_configuration(ctx, orc, seed):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([64, 64 * 5 + 9, 64 * 70 + 1]))
    p_bad = float(rng.choice([0.0, 0.001, 0.02]))
    if seed % 2 == 0:
        k = int(rng.integers(33, 65))
        wh = bool(rng.integers(0, 2))
        if seed % 4 == 0:
            L = max(k, int(rng.choice([64, 100, 112, 150, 160, 161, 250, 257, 300, 1000])))
            if L > 256:
                n = min(n, 64 * 5 + 9)
            host = _bytes(rng, n * L, p_bad, False)
            o = orc.canonical_reduce2(host, n, L, k, with_hash=wh)
            g = ctx.canonical_reduce2(ctx.to_device(host), n, L, k, with_hash=wh)
        else:      # behind an offsets array: uniform in fact (the device-side gate), ragged (the two-word ragged kernel), any bound
            top = int(rng.choice([70, 100, 150, 160, 250]))
            lens = np.full(n, top) if rng.integers(0, 3) == 0 else rng.integers(0, top + 1, size=n)
            if rng.integers(0, 3) == 0:
                lens[rng.integers(0, n)] = top + int(rng.integers(1, 400))      # one read past the bound
            L = int(rng.choice([0, top, max(top, 160), 100_000]))
            offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            host = _bytes(rng, int(offsets[-1]) + 16, p_bad, False)[: int(offsets[-1])]
            o = orc.canonical_reduce2(host, n, 0, k, with_hash=wh, offsets=offsets)
            g = ctx.canonical_reduce2(ctx.to_device(host), n, L, k, with_hash=wh, offsets=ctx.to_device(offsets))
        assert tuple(getattr(g, f) for f, _ in g._fields_) == tuple(getattr(o, f) for f, _ in o._fields_), (k, L, n)
    else:
        k = int(rng.choice([9, 15, 21, 31]))
        b = int(rng.choice([10, 14, 15, 18, 20, 21, 22, 23]))
        hasher, hk = [(1, k), (2, 0), (1, max(1, k - 2))][int(rng.integers(0, 3))]
        if rng.integers(0, 2):
            L = max(k, int(rng.choice([50, 100, 150, 200])))
            host = _bytes(rng, n * L, p_bad, False)
            o = orc.histogram(host, n, L, k, hk if hasher == 1 else 0, b)
            g = ctx.histogram(ctx.to_device(host), n, L, k, hasher, hk, b)
        else:
            lens = rng.integers(0, 161, size=n)
            offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            host = _bytes(rng, int(offsets[-1]) + 16, p_bad, False)[: int(offsets[-1])]
            o = orc.histogram(host, n, 0, k, hk if hasher == 1 else 0, b, offsets=offsets)
            d_host = ctx.to_device(host) if len(host) else ctx.to_device(np.zeros(16, np.uint8))
            g = ctx.histogram(d_host, n, int(rng.choice([0, 160])), k, hasher, hk, b, offsets=ctx.to_device(offsets))
        assert (g.cpu().numpy().view(np.uint64) == o).all(), (k, b, hasher, hk, n)


@pytest.mark.parametrize("seed", range(max(40, N_FUZZ // 3)))
def test_windows_random_configuration(ctx, orc, seed):
    """kmx_canonical_windows (materialise: staged / line-aligned write-back, uniform or ragged) against the oracle"""
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.choice([1, 2, 7, 13, 16, 17, 21, 27, 31]))
    n = int(rng.choice([1, 64, 65, 64 * 9 + 5]))
    p_bad = float(rng.choice([0.0, 0.0, 0.003, 0.05]))
    want = [("canon",), ("fw",), ("rc", "flags"), ("fw", "rc", "canon", "flags")][int(rng.integers(0, 4))]
    if rng.integers(0, 2):
        L = max(k, int(rng.choice([k, 40, 100, 128, 150, 158, 160, 200, 256, 300, 500, 1000])))
        if L > 300:
            n = min(n, 65)
        host = _bytes(rng, n * L, p_bad, bool(rng.integers(0, 2)))
        o = orc.canonical_windows(host, n, L, k)
        g = ctx.canonical_windows(ctx.to_device(host), n, L, k, want=want)
    else:
        lens = rng.integers(0, int(rng.choice([60, 160, 260])) + 1, size=n)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        host = _bytes(rng, int(offsets[-1]) + 16, p_bad, False)[: int(offsets[-1])]
        o = orc.canonical_windows(host, n, 0, k, offsets=offsets)
        d_host = ctx.to_device(host) if len(host) else ctx.to_device(np.zeros(16, np.uint8))   # (a NULL base pointer is an argument error even for empty reads)
        g = ctx.canonical_windows(d_host, n, int(rng.choice([0, 160])), k, offsets=ctx.to_device(offsets), host_offsets=offsets, want=want)
    ref = dict(zip(("fw", "rc", "canon", "flags"), o))
    for name in want:
        got = g[name].cpu().numpy()
        got = got.view(np.uint64) if name != "flags" else got
        assert np.array_equal(got, ref[name]), (name, k, n, p_bad)


@pytest.mark.parametrize("seed", range(max(40, N_FUZZ // 3)))
def test_windows2_random_configuration(ctx, orc, seed):
    """kmx_canonical_windows2 (two-word materialise: tiled uniform, tiled ragged, long reads as segments, the per-read fallbacks)"""
    rng = np.random.default_rng(15000 + seed)
    k = int(rng.integers(33, 65))
    n = int(rng.choice([1, 64, 65, 64 * 5 + 7]))
    p_bad = float(rng.choice([0.0, 0.0, 0.002, 0.03]))
    if rng.integers(0, 2):
        L = max(k, int(rng.choice([k, 100, 150, 160, 161, 250, 256, 257, 300, 700])))
        if L > 300:
            n = min(n, 65)
        host = _bytes(rng, n * L, p_bad, False)
        o = orc.canonical_windows2(host, n, L, k)
        g = ctx.canonical_windows2(ctx.to_device(host), n, L, k)
    else:
        lens = rng.integers(0, int(rng.choice([100, 160, 300])) + 1, size=n)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        host = _bytes(rng, int(offsets[-1]) + 16, p_bad, False)[: int(offsets[-1])]
        o = orc.canonical_windows2(host, n, 0, k, offsets=offsets)
        d_host = ctx.to_device(host) if len(host) else ctx.to_device(np.zeros(16, np.uint8))
        g = ctx.canonical_windows2(d_host, n, int(rng.choice([0, 160, 256])), k, offsets=ctx.to_device(offsets), host_offsets=offsets)
    ref = dict(zip(("fw", "rc", "canon", "flags"), o))
    for name in ("fw", "rc", "canon"):
        assert np.array_equal(g[name].cpu().numpy().view(np.uint64).reshape(-1, 2), ref[name].reshape(-1, 2)), (name, k, n, p_bad)
    assert np.array_equal(g["flags"].cpu().numpy(), ref["flags"]), (k, n, p_bad)


@pytest.mark.parametrize("seed", range(max(40, N_FUZZ // 3)))
def test_seqvec_random_configuration(ctx, orc, seed):
    """2-bit packed reads: kmx_seqvec_canonical_reduce (the packed bit-sliced kernel and its fallbacks) against the oracle's
    scan of the same bases as ASCII; kmx_seqvec_minimizers against the oracle's deque"""
    from kmers_amd import _lib

    rng = np.random.default_rng(12000 + seed)
    k = int(rng.choice([1, 3, 12, 13, 16, 17, 21, 24, 29, 31]))
    L = max(k, int(rng.choice([k, 31, 32, 50, 64, 100, 128, 150, 151, 160, 161, 200, 256, 300])))
    n = int(rng.choice([1, 16, 64, 64 * 6 + 3]))
    host = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * L)]
    words = ctx.seqvec_from_bytes(ctx.to_device(host))
    hasher, hk = [(_lib.HASH_NONE, 0), (_lib.HASH_LEX, k), (_lib.HASH_LEX, int(rng.integers(1, 33)))][int(rng.integers(0, 3))]
    o = orc.canonical_reduce(host, n, L, k, hasher_k=hk if hasher == _lib.HASH_LEX else 0)
    g = ctx.seqvec_canonical_reduce(words, n, L, k, hasher, hk, 0)
    assert (g.n_valid, g.sum_canon) == (o.n_valid, o.sum_canon), (k, L, n)
    if hasher == _lib.HASH_LEX:
        assert g.xor_hash == o.xor_hash, (k, L, n, hk)
    if seed % 3 == 0 and L <= 200 and n <= 64:
        w = int(rng.integers(1, k + 1))
        mhk = int(rng.choice([0, w, min(32, w + 3)]))
        sv = orc.SeqVector(host.tobytes())
        ow, op = orc.seqvec_minimizers(sv, n, L, k, w, mhk)
        gw, gp = ctx.seqvec_minimizers(words, n, L, k, w, _lib.HASH_LEX if mhk else _lib.HASH_IDENTITY, mhk)
        assert np.array_equal(gw.cpu().numpy().view(np.uint64), ow) and np.array_equal(gp.cpu().numpy().view(np.uint32), op), (k, w, mhk, L, n)
