"""GPU parity tests of the entry points added in round 2 (all through the libkmx C ABI, against the CPU oracle):
decode / display direction, Encoding<P, B> for every utils::Data word type (row a18),
the RCCL communicator behind the ABI (row e; one rank: a GPU box has one GPU), stream discipline of the
torch-facing layer, the calibration kernel."""
import ctypes as C

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


def _words(rng, n, k):
    w = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    if k < 32:
        w &= np.uint64((1 << (2 * k)) - 1)
    return w


# ------------------------------------------------------------------ f3: sub_kmer_word / String::from / bitmer_to_bytes

@pytest.mark.parametrize("k,pos,width", [(31, 0, 31), (31, 5, 12), (31, 30, 1), (32, 0, 32), (32, 1, 31), (21, 7, 14), (1, 0, 1),
                                         (13, 12, 0)])
def test_sub_kmer_words(ctx, orc, k, pos, width):
    rng = np.random.default_rng(k * 1000 + pos * 37 + width)
    w = _words(rng, 1000 + k, k)
    g = ctx.sub_kmer_words(ctx.to_device(w), k, pos, width).cpu().numpy().view(np.uint64)
    exp = np.array([orc.sub_kmer_word(int(x), k, pos, width) for x in w], dtype=np.uint64)
    assert (g == exp).all()


def test_sub_kmer_words_asserts_become_codes(ctx):
    from kmers_amd import _lib

    import torch
    w = torch.zeros(4, dtype=torch.int64, device=ctx.device)
    for k, pos, width in [(31, 31, 0), (31, 20, 12), (31, 40, 1)]:      # kmer.rs:157-158
        with pytest.raises(_lib.KmxError) as e:
            ctx.sub_kmer_words(w, k, pos, width)
        assert e.value.status == _lib.E_ARG
    with pytest.raises(_lib.KmxError) as e:
        ctx.sub_kmer_words(w, 33, 0, 1)
    assert e.value.status == _lib.E_K_RANGE


def test_sub_kmer_reference_kat(ctx, kats):
    """src/naive_impl/kmer.rs:529-542 (sub_kmer of "ACTTGAT" == Kmer::from of the substring), every (pos, width)"""
    from oracle import oracle

    s = kats["sub_kmer"]["seq"].encode()
    km = oracle.kmer_from_bytes(s)
    w = ctx.to_device(np.array([km.data], dtype=np.uint64))
    for pos in range(len(s)):
        for width in range(1, len(s) - pos + 1):
            got = int(ctx.sub_kmer_words(w, len(s), pos, width).cpu().numpy().view(np.uint64)[0])
            assert got == oracle.kmer_from_bytes(s[pos:pos + width]).data


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 8, 13, 21, 31, 32])
def test_kmers_to_strings_and_bitmer_to_bytes(ctx, orc, k):
    rng = np.random.default_rng(k)
    n = 777
    w = _words(rng, n, k)
    d = ctx.to_device(w)
    low = ctx.kmers_to_strings(d, k).cpu().numpy().tobytes()
    up = ctx.bitmers_to_bytes(d, k).cpu().numpy().tobytes()
    for i in range(n):
        exp = orc.bitmer_to_bytes(int(w[i]), k)                          # src/kmer.rs:71-91
        assert up[i * k:(i + 1) * k] == exp
        assert low[i * k:(i + 1) * k] == exp.lower()                       # kmer.rs:196-207 (BASE_TABLE is lower case)
    assert orc.kmer_to_string(orc.Kmer(k, int(w[0]))).encode() == low[:k]


def test_kmer_string_round_trip(ctx, orc):
    """Kmer::from(&str) -> String::from(Kmer) is the identity on lower-case ACGT (kmer.rs:260-271 round trips)"""
    rng = np.random.default_rng(5)
    k, n = 27, 500
    seqs = np.frombuffer(b"acgt", np.uint8)[rng.integers(0, 4, n * k)]
    words = ctx.kmers_from_bytes(ctx.to_device(seqs), n, k)
    back = ctx.kmers_to_strings(words, k).cpu().numpy()
    assert (back == seqs).all()


# ------------------------------------------------------------------ a18: Encoding<P, B> for u8 / u16 / u32 / u64 / u128

P_SHAPES = [(8, 1), (8, 3), (8, 8), (16, 1), (16, 2), (16, 5), (32, 1), (32, 3), (64, 1), (64, 2), (128, 1), (128, 2), (128, 4)]
ENCS = [0x1E, 0x1B, 0xE4, 0x27, 0x4B, 0xD8]   # ACGT, ACTG (= Xor10), TGCA, ... (naive.rs:48-74 discriminants)


@pytest.mark.parametrize("word_bits,B", P_SHAPES)
def test_encoding_p_encode_decode_revcomp(ctx, orc, word_bits, B):
    rng = np.random.default_rng(word_bits * 100 + B)
    nb = word_bits // 8 * B
    cap = 4 * nb
    n = 300
    for enc in ENCS:
        for seq_len in sorted({1, 2, cap // 2, cap - 1, cap} - {0}):
            seqs = rng.integers(0, 256, n * seq_len, dtype=np.uint8)      # no validity check: every byte maps (naive.rs:14-16)
            arr = ctx.encode_kmers_p(ctx.to_device(seqs), n, seq_len, enc, word_bits, B)
            got = arr.cpu().numpy().reshape(n, nb)
            for i in (0, 1, n // 2, n - 1):
                exp = orc.naive_encode(enc, seqs[i * seq_len:(i + 1) * seq_len].tobytes(), nb)
                assert (got[i] == exp).all(), (enc, seq_len, i)
            dec = ctx.encoding_decode_p(arr, enc, word_bits, B).cpu().numpy().reshape(n, cap)
            for i in (0, n - 1):
                assert dec[i].tobytes() == orc.naive_decode(enc, got[i])
            if seq_len >= 2:
                rc = ctx.encoding_rev_comp_p(arr, seq_len, enc, word_bits, B).cpu().numpy().reshape(n, nb)
                for i in (0, 2, n - 1):
                    assert (rc[i] == orc.naive_rev_comp(enc, seq_len, got[i])).all(), (enc, seq_len, i)


def test_encoding_p_reference_kats(ctx, kats, orc):
    """the exact-word KATs of src/encoding/naive.rs:297-445 (P = u8, u16, u32, u64, u128) through the GPU:
    encode words, decode, and decode of rev_comp"""
    encs = kats["naive_encodings"]["enc_bytes"]
    cases = kats["naive_encode_kats"]["cases"]
    assert {c["p_bits"] for c in cases} >= {8, 16, 32, 64}
    for c in cases:
        enc, p_bits, B, K = encs[c["enc"]], c["p_bits"], c["B"], c["K"]
        seq = np.frombuffer(c["seq"].encode(), np.uint8)
        arr = ctx.encode_kmers_p(ctx.to_device(seq), 1, len(seq), enc, p_bits, B)
        assert orc.words(arr.cpu().numpy(), p_bits) == [int(w) for w in c["words"]], c["name"]
        assert ctx.encoding_decode_p(arr, enc, p_bits, B).cpu().numpy().tobytes() == c["decode"].encode(), c["name"]
        rc = ctx.encoding_rev_comp_p(arr, K, enc, p_bits, B)
        assert ctx.encoding_decode_p(rc, enc, p_bits, B).cpu().numpy().tobytes() == c["decode_rev_comp"].encode(), c["name"]


def test_encoding_p_matches_u64_entry_points(ctx):
    rng = np.random.default_rng(9)
    n, K = 500, 45
    seqs = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * K)]
    d = ctx.to_device(seqs)
    a = ctx.encode_kmers(d, n, K, 0x1E, 2).cpu().numpy().view(np.uint8)
    b = ctx.encode_kmers_p(d, n, K, 0x1E, 64, 2).cpu().numpy()
    assert (a == b).all()
    c = ctx.encode_kmers_p(d, n, K, 0x1E, 16, 8).cpu().numpy()
    assert (a == c).all()          # same flat bit string whatever P


def test_encoding_p_capacity_is_a_code(ctx):
    from kmers_amd import _lib

    import torch
    s = torch.zeros(100, dtype=torch.uint8, device=ctx.device)
    with pytest.raises(_lib.KmxError) as e:
        ctx.encode_kmers_p(s, 1, 5, 0x1E, 8, 1)       # 5 bases into a [u8; 1]: bit_field panics
    assert e.value.status == _lib.E_TOO_LONG
    with pytest.raises(_lib.KmxError) as e:
        ctx.encode_kmers_p(s, 1, 4, 0x1E, 24, 1)
    assert e.value.status == _lib.E_ARG


# ------------------------------------------------------------------ RCCL communicator behind the ABI (one rank)

def test_kmx_comm_single_rank(ctx, orc):
    import torch

    from kmers_amd.api import Comm

    comm = Comm(ctx, 1, 0)
    assert comm.size() == 1
    counts = torch.arange(1 << 12, dtype=torch.int64, device=ctx.device) * 3
    before = counts.clone()
    comm.histogram_allreduce(counts)
    ctx.synchronize()
    assert torch.equal(counts, before)
    n, L, k = 1000, 150, 31
    bases = ctx.gen_reads(n * L)
    from kmers_amd import _lib
    s = ctx.canonical_reduce_async(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    ref = s.clone()
    comm.summary_allreduce(s)
    ctx.synchronize()
    assert torch.equal(s, ref)
    comm.close()


def test_torch_nccl_single_rank_on_device(ctx, orc):
    """kmers_amd.dist on a real nccl (= RCCL) process group of one rank, device tensors"""
    import os

    import torch
    import torch.distributed as dist

    from kmers_amd import dist as kd

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=ctx.device)
    try:
        assert kd.rccl_rank_count(ctx.device) == 1
        local = {"n_valid": 123, "sum_canon": M64 - 5, "xor_hash": 0x8000000000000001, "sum_fw": 77}
        assert kd.combine_summaries(local, device=ctx.device) == local
        counts = torch.ones(1 << 10, dtype=torch.int64, device=ctx.device)
        kd.allreduce_histogram(counts)
        assert int(counts.sum().item()) == 1 << 10
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ stream discipline (api.py:46)

def test_context_on_side_stream(orc):
    import torch

    from kmers_amd import _lib
    from kmers_amd.api import Context

    side = torch.cuda.Stream()
    c = Context(stream=side)
    n, L, k = 20_000, 150, 31
    for it in range(5):
        bases = c.gen_reads(n * L, first_byte=it * 1000)
        g = c.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, 0)
        host = bases.cpu().numpy()   # default stream: order it behind the side stream first
        side.synchronize()
        host = bases.cpu().numpy()
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)
        outs = c.canonical_windows(bases, 256, L, k, want=("canon",))
        side.synchronize()
        _, _, canon, _ = orc.canonical_windows(host, 256, L, k)
        assert (outs["canon"].cpu().numpy().view(np.uint64) == canon).all()
    c.close()


def test_default_context_used_inside_other_stream(orc):
    import torch

    from kmers_amd.api import Context

    c = Context()
    other = torch.cuda.Stream()
    n, L, k = 10_000, 150, 21
    bases = c.gen_reads(n * L)
    torch.cuda.synchronize()
    host = bases.cpu().numpy()
    o = orc.canonical_reduce(host, n, L, k)
    with torch.cuda.stream(other):
        for _ in range(10):
            g = c.canonical_reduce(bases, n, L, k)
            assert (g.n_valid, g.sum_canon) == (o.n_valid, o.sum_canon)
    c.close()


# ------------------------------------------------------------------ calibration kernel

@pytest.mark.parametrize("nbytes", [0, 16, 9600, 9600 * 7 + 160, 150 * 64 * 1000 + 48])
def test_calib_stream_read_folds_every_chunk(ctx, nbytes):
    import torch

    rng = np.random.default_rng(nbytes + 1)
    host = rng.integers(0, 256, max(nbytes, 16), dtype=np.uint8)[:nbytes]
    buf = ctx.to_device(host) if nbytes else torch.zeros(16, dtype=torch.uint8, device=ctx.device)[:0]
    out = ctx.calib_stream_read(buf)
    got = int(out.cpu().numpy().view(np.uint64)[0])
    words = host[: nbytes // 16 * 16].view(np.uint32)
    exp = int(np.bitwise_xor.reduce(words)) if words.size else 0
    assert got == exp


# ------------------------------------------------------------------ no argument sends a call to the slow kernel

def _dirty(rng, n, p_bad=0.004):
    alpha = np.frombuffer(b"ACGTacgt", np.uint8)
    a = alpha[rng.integers(0, 8, n)].copy()
    bad = rng.random(n) < p_bad
    a[bad] = rng.integers(0, 256, int(bad.sum()), dtype=np.uint8)
    return a


def _identity_fold(orc, host, n, L, k, offsets=None):
    """xor of the canonical words of every yielded window = hash fold under the identity hasher (hash.rs:4-8)"""
    _, _, canon, flags = orc.canonical_windows(host, n, L, k, offsets=offsets)
    v = canon[(flags & 1) != 0]
    return int(np.bitwise_xor.reduce(v)) if v.size else 0


@pytest.mark.parametrize("k", [5, 12, 13, 21, 27, 31])
@pytest.mark.parametrize("L,n", [(150, 64 * 40 + 9), (100, 64 * 7), (250, 64 * 5 + 3)])
def test_reduce_any_hasher(ctx, orc, k, L, n):
    """LexHasher(hasher_k != k) and the identity hasher on every tiled reduce path (they used to drop to the per-lane kernel)"""
    from kmers_amd import _lib

    rng = np.random.default_rng(k * 31 + L)
    host = _dirty(rng, n * L)
    bases = ctx.to_device(host)
    for hk in sorted({1, 7, k - 1, k, min(k + 1, 32), 32} - {0}):
        o = orc.canonical_reduce(host, n, L, k, hasher_k=hk)
        g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, hk, _lib.REDUCE_SUM_FW)
        assert (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw) == (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw), hk
    g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_IDENTITY, 0, 0)
    assert g.xor_hash == _identity_fold(orc, host, n, L, k)
    # ragged reads
    lens = rng.integers(0, L + 1, n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    hostr = _dirty(rng, int(off[-1]) + 16)[: int(off[-1])]
    d_r, d_off = ctx.to_device(np.concatenate([hostr, np.zeros(16, np.uint8)]))[: hostr.size], ctx.to_device(off)
    for hk in (3, k, 29):
        o = orc.canonical_reduce(hostr, n, 0, k, hasher_k=hk, offsets=off)
        g = ctx.canonical_reduce(d_r, n, L, k, _lib.HASH_LEX, hk, 0, offsets=d_off)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash), hk
    g = ctx.canonical_reduce(d_r, n, L, k, _lib.HASH_IDENTITY, 0, 0, offsets=d_off)
    assert g.xor_hash == _identity_fold(orc, hostr, n, 0, k, offsets=off)


@pytest.mark.parametrize("k", [9, 21, 31])
def test_seqvec_reduce_any_hasher(ctx, orc, k):
    from kmers_amd import _lib

    rng = np.random.default_rng(k)
    L, n = 150, 64 * 20 + 5
    host = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * L)]
    words = ctx.seqvec_from_bytes(ctx.to_device(host))
    for hk in (4, k, 32):
        o = orc.canonical_reduce(host, n, L, k, hasher_k=hk)
        g = ctx.seqvec_canonical_reduce(words, n, L, k, _lib.HASH_LEX, hk, 0)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)
    g = ctx.seqvec_canonical_reduce(words, n, L, k, _lib.HASH_IDENTITY, 0, 0)
    assert g.xor_hash == _identity_fold(orc, host, n, L, k)


@pytest.mark.parametrize("k", list(range(34, 65, 2)))
@pytest.mark.parametrize("L,n,p_bad", [(150, 64 * 30 + 7, 0.0), (150, 64 * 30 + 7, 0.0005), (100, 64 * 12, 0.0), (160, 64 * 9 + 1, 0.0)])
def test_reduce2_bitsliced_every_even_k(ctx, orc, k, L, n, p_bad):
    """[u64;2] k-mers: the even k from 34 to 64 on the bit-sliced kernel too (round 1: lane-per-read kernel)"""
    if L < k:
        pytest.skip("read shorter than k")
    rng = np.random.default_rng(k * 7 + L)
    host = _dirty(rng, n * L, p_bad)
    for with_hash in (True, False):
        o = orc.canonical_reduce2(host, n, L, k, with_hash=with_hash)
        g = ctx.canonical_reduce2(ctx.to_device(host), n, L, k, with_hash=with_hash)
        assert tuple(getattr(g, f) for f, _ in g._fields_) == tuple(getattr(o, f) for f, _ in o._fields_)


@pytest.mark.parametrize("lead", [1, 2, 3, 6, 8, 13, 15])
@pytest.mark.parametrize("k,L", [(31, 150), (21, 150), (11, 150), (5, 100), (31, 159), (27, 250), (47, 150), (63, 100), (31, 160), (17, 120), (31, 100), (21, 75), (13, 112), (29, 111)])
def test_base_pointer_not_16_byte_aligned(ctx, orc, lead, k, L):
    """reads that start at any byte address (a slice of a larger buffer): reduce, windows and histogram against the oracle.
    (L = 160 / 256 with an unaligned base does not fit the frame and takes the per-lane kernel: still exact.)"""
    from kmers_amd import _lib

    rng = np.random.default_rng(lead * 100 + k)
    n = 64 * 11 + 5
    host = _dirty(rng, n * L, 0.0008)
    buf = ctx.to_device(np.concatenate([rng.integers(0, 256, lead, dtype=np.uint8), host, np.zeros(32, np.uint8)]))
    bases = buf[lead: lead + n * L]
    assert bases.data_ptr() % 16 == lead
    if k <= 31:
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        assert (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw) == (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)
        _, _, canon, flags = orc.canonical_windows(host, n, L, k)
        outs = ctx.canonical_windows(bases, n, L, k, want=("canon", "flags"))
        assert (outs["flags"].cpu().numpy() == flags).all()
        assert (outs["canon"].cpu().numpy().view(np.uint64) == canon).all()
        for b in (10, 16):
            h = ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b).cpu().numpy().view(np.uint64)
            assert (h == orc.histogram(host, n, L, k, k, b)).all()
    else:
        o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
        g = ctx.canonical_reduce2(bases, n, L, k, with_hash=True)
        assert tuple(getattr(g, f) for f, _ in g._fields_) == tuple(getattr(o, f) for f, _ in o._fields_)


@pytest.mark.parametrize("L", [100, 150, 200])
def test_k17_on_the_tiled_kernels(ctx, orc, L):
    """k = 17 (two dwords per k-mer at the V = 1 register index) used to be the one single-word k without a tiled
    materialise / histogram kernel"""
    from kmers_amd import _lib

    rng = np.random.default_rng(L)
    n, k = 64 * 9 + 3, 17
    host = _dirty(rng, n * L, 0.001)
    bases = ctx.to_device(host)
    fw, rc, canon, flags = orc.canonical_windows(host, n, L, k)
    outs = ctx.canonical_windows(bases, n, L, k)
    assert (outs["flags"].cpu().numpy() == flags).all()
    for name, exp in (("fw", fw), ("rc", rc), ("canon", canon)):
        assert (outs[name].cpu().numpy().view(np.uint64) == exp).all(), name
    for b in (8, 14, 18):
        h = ctx.histogram(bases, n, L, k, _lib.HASH_LEX, k, b).cpu().numpy().view(np.uint64)
        assert (h == orc.histogram(host, n, L, k, k, b)).all()


@pytest.mark.parametrize("k", [13, 21, 30, 31])
@pytest.mark.parametrize("L,n", [(257, 64 * 9 + 5), (300, 64 * 6), (1000, 64 * 3 + 1), (5003, 150), (20000, 70)])
def test_long_uniform_reads_take_the_segmented_bitsliced_path(ctx, orc, k, L, n):
    """uniform reads longer than the 256-base frame (round 1: lane-per-read kernel at 0.36 TB/s): cut into overlapping
    160-base segments on the ragged bit-sliced kernel; clean and with invalid bytes, with and without the hash fold"""
    from kmers_amd import _lib

    rng = np.random.default_rng(k * 1000 + L)
    for p_bad in (0.0, 0.0004):
        host = _dirty(rng, n * L, p_bad)
        bases = ctx.to_device(host)
        o = orc.canonical_reduce(host, n, L, k, hasher_k=k)
        g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, 0)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash), p_bad
        g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_NONE, 0, 0)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, 0)
        g = ctx.canonical_reduce(bases, n, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)   # sum_fw: the per-lane kernel, still exact
        assert (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw) == (o.n_valid, o.sum_canon, o.xor_hash, o.sum_fw)


# ------------------------------------------------------------------ bench.py, N > 1 control flow on one GPU

@pytest.mark.gpu
@pytest.mark.parametrize("k", [13, 21, 31])
@pytest.mark.parametrize("hint", [100, 111, 64, 40])
def test_short_ragged_reads_on_the_7_word_frame(ctx, orc, k, hint):
    """ragged reads with a length bound of at most 111: the 7-word frame of the bit-sliced kernel; reads longer than the bound
    (the hint is only a hint), shorter than k, empty, dirty"""
    from kmers_amd import _lib
    rng = np.random.default_rng(k * 100 + hint)
    n = 64 * 40 + 13
    lens = rng.integers(0, min(hint, 100) + 1, size=n)
    lens[::131] = hint + 9          # past the bound: the tile must still come out right
    lens[7::257] = 0
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]) + 16, 0.002)[: int(offsets[-1])]
    d_host, d_off = ctx.to_device(host), ctx.to_device(offsets)
    o = orc.canonical_reduce(host, n, 0, k, hasher_k=k, offsets=offsets)
    for _ in range(2):
        g = ctx.canonical_reduce(d_host, n, hint, k, _lib.HASH_LEX, k, 0, offsets=d_off)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)


@pytest.mark.gpu
def test_reads_length_range(ctx):
    """kmx_reads_length_range: the bound kmx_reads.read_len wants for ragged input"""
    rng = np.random.default_rng(3)
    lens = rng.integers(37, 152, size=100_003)
    lens[77] = 7
    lens[99_999] = 301
    off = ctx.to_device(np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64))
    assert ctx.reads_length_range(off) == (7, 301)
    assert ctx.reads_length_range(ctx.to_device(np.arange(1001, dtype=np.uint64) * np.uint64(150))) == (150, 150)
    assert ctx.reads_length_range(ctx.to_device(np.zeros(1, dtype=np.uint64))) == (0, 0)
    # (round 3) a read of 2^31 bases or more is refused: KMX_E_ARG (the scans skip such reads; kmx.h "Limits")
    from kmers_amd._lib import E_ARG, KmxError

    big = np.array([0, 5, 5 + (1 << 33)], dtype=np.uint64)
    with pytest.raises(KmxError) as ei:
        ctx.reads_length_range(ctx.to_device(big))
    assert ei.value.status == E_ARG
    ok = np.array([0, 5, 5 + (1 << 31) - 1], dtype=np.uint64)     # the longest read the scans take
    assert ctx.reads_length_range(ctx.to_device(ok)) == (5, (1 << 31) - 1)


@pytest.mark.gpu
@pytest.mark.parametrize("k,L", [(31, 150), (21, 150), (13, 100), (47, 150), (64, 150), (31, 159)])
@pytest.mark.parametrize("layout", ["uniform", "offsets"])
def test_blanked_reads_in_every_position(ctx, orc, k, L, layout):
    """the reads with an invalid byte are blanked out of their tile in the main pass and handled by sweep_flagged_kernel: a tile
    with ALL its reads dirty, with exactly one, N at the first / last byte of a read (the chunk it shares with its neighbour
    blanks both: they are rolled exactly), N in every read's k-th base, two calls in a row (the masks must be back to zero)"""
    from kmers_amd import _lib
    rng = np.random.default_rng(k * 1000 + L)
    n = 64 * 14 + 5
    host = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n * L).copy()
    def put(read, pos):
        host[read * L + pos] = ord("N")
    for r in range(64):                 # tile 0: every read
        put(r, int(rng.integers(0, L)))
    put(64 + 17, 0)                     # tile 1: one read, first byte
    put(128 + 63, L - 1)                # tile 2: last read of the tile, last byte (shares its chunk with tile 3's first read)
    put(192, L - 1); put(193, 0)        # tile 3: neighbours across one chunk
    for r in range(256, 320):           # tile 4: base k-1 of every read (no window of the first k survives)
        put(r, k - 1)
    put(5 * 64 + 3, 7); put(5 * 64 + 3, 90)   # tile 5: two in one read
    put(n - 2, 11)                      # the final partial tile (rolled per lane anyway)
    offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L)) if layout == "offsets" else None
    if layout == "offsets" and k > 32:
        pytest.skip("two-word k-mers: uniform layout only")
    d_host, d_off = ctx.to_device(host), (ctx.to_device(offsets) if offsets is not None else None)
    for _ in range(2):
        if k <= 32:
            o = orc.canonical_reduce(host, n, L, k, hasher_k=k, offsets=offsets)
            g = ctx.canonical_reduce(d_host, n, L if offsets is None else 160, k, _lib.HASH_LEX, k, 0, offsets=d_off)
            assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)
        else:
            o = orc.canonical_reduce2(host, n, L, k, with_hash=True)
            g = ctx.canonical_reduce2(d_host, n, L, k, with_hash=True)
            assert tuple(getattr(g, f) for f, _ in g._fields_) == tuple(getattr(o, f) for f, _ in o._fields_)


@pytest.mark.gpu
@pytest.mark.parametrize("hasher,hk", [(1, 31), (2, 0), (1, 27)])
@pytest.mark.parametrize("hint", [0, 160, 120])
def test_histogram_ragged_takes_the_partitioned_path(ctx, orc, hasher, hk, hint):
    """>= 4096 ragged reads, 2^15..2^21 buckets: partition pass + per-partition tables, with the caller's length bound loose, absent
    or too small (longer reads then overfill their segments and go to the global table), against the oracle"""
    rng = np.random.default_rng(hint + hk)
    k, b = 31, 20
    lens = rng.integers(20, 161, size=64 * 90 + 17)
    lens[::211] = 300          # past the frame: rolled per lane
    lens[5::389] = 0
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]) + 16, 0.0005)[: int(offsets[-1])]
    o = orc.histogram(host, len(lens), 0, k, hk if hasher == 1 else 0, b, offsets=offsets)
    g = ctx.histogram(ctx.to_device(host), len(lens), hint, k, hasher, hk, b, offsets=ctx.to_device(offsets))
    g = g.cpu().numpy().view(np.uint64)
    assert int(g.sum()) == int(o.sum())
    assert (g == o).all()


@pytest.mark.parametrize("cfg", ["1", "3", "4"])
def test_bench_two_ranks_share_the_gpu(cfg):
    """`python bench.py --gpus 2` (it starts its own ranks) with both ranks on cuda:0 and a gloo process group
    (KMX_BENCH_TEST_SHARED_GPU): barriers, shard streams, summary combine, histogram exchange (the torch.distributed
    route: RCCL cannot put two ranks on one device), verdict broadcast -- everything of the N > 1 path but RCCL itself."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["KMX_BENCH_TEST_SHARED_GPU"] = "1"
    n = 2_000_000
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", cfg, "--reads-per-gpu", str(n),
                        "--steps", "3", "--warmup", "1", "--sustain-steps", "5", "--no-traffic", "--cpu-baseline-seconds", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]     # stdout is the ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["parity_vs_oracle"] == "ok"
    assert "TEST_MODE" in d["config"]
    # an N > 1 line is complete: the CPU baseline of the same run (rank 0, host cores stated), the spread over the ranks
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "sample" in cb
    pr = d["per_rank"]
    assert 0 < pr["ms_per_step_min"] <= pr["ms_per_step_max"] <= d["ms_per_step"] * 1.0001
    assert 0 < pr["kernel_ms_min"] <= pr["kernel_ms_max"]
    per_rank = n * (150 - 31 + 1)
    if cfg == "4":
        h = d["histogram"]
        assert h["total_count"] == h["expect"] == 2 * per_rank
        assert "torch.distributed" in h["collective"]
        assert h["allreduce_ms"] > 0 and h["scan_ms"] > 0
    else:
        # value = k-mers of BOTH ranks per step / time
        assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * per_rank) < 1e-6 * 2 * per_rank
        assert pr["summary_combine_ms"] is not None and pr["summary_combine_ms"] >= 0
