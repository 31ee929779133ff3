"""GPU parity of kmx_fastx_parse (build-defined semantics, see tests/test_oracle_fastx.py) against
the CPU oracle, and of the whole chain file image -> reads -> canonical k-mer summary."""
import numpy as np
import pytest

from fastx_cases import EDGE_TEXTS, fasta_text, fastq_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from kmers_amd.api import Context

    c = Context()
    yield c
    c.close()


def _check(ctx, orc, text, fmt=0):
    from kmers_amd.api import u64_numpy

    eb, eo = orc.fastx_parse(text, fmt)
    bases, offsets = ctx.fastx_parse(ctx.to_device(text) if len(text) else ctx.empty(0, __import__("torch").uint8), fmt)
    assert np.array_equal(u64_numpy(offsets), eo)
    assert np.array_equal(bases.cpu().numpy(), eb)
    # the single call into buffers sized for a bound on the records: exact and generous bound
    for bound in (len(eo) - 1, len(text) // 2 + 1):
        b1, o1 = ctx.fastx_parse(ctx.to_device(text) if len(text) else ctx.empty(0, __import__("torch").uint8), fmt, max_reads=bound)
        assert np.array_equal(u64_numpy(o1), eo)
        assert np.array_equal(b1.cpu().numpy(), eb)
    return bases, offsets, eb, eo


@pytest.mark.parametrize("i", range(len(EDGE_TEXTS)))
def test_edge_texts(ctx, orc, i):
    _check(ctx, orc, EDGE_TEXTS[i])


@pytest.mark.parametrize("crlf", [False, True])
@pytest.mark.parametrize("trail", [False, True])
def test_random_fastq(ctx, orc, crlf, trail):
    rng = np.random.default_rng(21 + 2 * crlf + trail)
    for n, lo, hi in ((1, 0, 50), (9, 0, 10), (3000, 0, 200), (2000, 140, 160), (40, 3000, 9000)):
        _check(ctx, orc, fastq_text(rng, n, lo, hi, crlf=crlf, trail=trail))
        _check(ctx, orc, fastq_text(rng, n, lo, hi, crlf=crlf, trail=trail), 1)


@pytest.mark.parametrize("crlf", [False, True])
@pytest.mark.parametrize("trail", [False, True])
def test_random_fasta(ctx, orc, crlf, trail):
    rng = np.random.default_rng(31 + 2 * crlf + trail)
    for n, lo, hi, width in ((1, 0, 50, 60), (9, 0, 10, 3), (3000, 0, 400, 60), (500, 0, 400, 1), (30, 20000, 90000, 80), (3, 300000, 400000, 1 << 30), (120, 30000, 60000, 70)):
        _check(ctx, orc, fasta_text(rng, n, lo, hi, width=width, crlf=crlf, trail=trail))
        _check(ctx, orc, fasta_text(rng, n, lo, hi, width=width, crlf=crlf, trail=trail), 2)


def test_chunk_boundaries(ctx, orc):
    """line starts, '>' and the record structure right at the 16-byte lane, 4 KiB row and 128 KiB chunk boundaries"""
    rng = np.random.default_rng(5)
    for boundary in (16, 4096, 131072, 262144):
        for delta in range(-3, 4):
            head = b">" + b"h" * (boundary + delta - 2) + b"\n"          # the newline lands at boundary + delta - 1
            _check(ctx, orc, head + b"ACGT\n>b\nGG\n")
            seq = b">a\n" + b"A" * (boundary + delta - 4) + b"\n"         # a sequence line ends there, a header follows
            _check(ctx, orc, seq + b">b\nCC\n" + seq)
            fq = b"@" + b"h" * (boundary + delta - 2) + b"\n" + b"ACGT\n+\nIIII\n"
            _check(ctx, orc, fq + fastq_text(rng, 5))


def test_wrong_format(ctx):
    from kmers_amd._lib import KmxError

    with pytest.raises(KmxError):
        ctx.fastx_parse(ctx.to_device(b"ACGT\nACGT\n"))
    with pytest.raises(KmxError):
        ctx.fastx_parse(ctx.to_device(b">x\nACGT\n"), 1)
    with pytest.raises(KmxError):
        ctx.fastx_parse(ctx.to_device(b">x\nACGT\n"), 1, max_reads=4)


def test_more_records_than_the_bound(ctx, orc):
    """the single call with a bound the image exceeds: KMX_E_NOMEM, the counts come back, nothing is written past the buffers"""
    import ctypes as C

    import torch

    from kmers_amd import _lib
    from kmers_amd.api import _ptr, u64_numpy

    rng = np.random.default_rng(3)
    text = fastq_text(rng, 5000, 20, 200)
    eb, eo = orc.fastx_parse(text, 0)
    d = ctx.to_device(text)
    bases = ctx.empty(len(text), torch.uint8)
    offsets = torch.full((1000 + 1 + 64,), -1, dtype=torch.int64, device=bases.device)
    nr, nb = C.c_uint64(0), C.c_uint64(0)
    rc = ctx.lib.kmx_fastx_parse(ctx._h, _ptr(d), d.numel(), 1, _ptr(bases), _ptr(offsets), 1000, C.byref(nr), C.byref(nb))
    assert rc == _lib.E_NOMEM and nr.value == len(eo) - 1 and nb.value == len(eb)
    assert bool((offsets[1001:] == -1).all())


@pytest.mark.parametrize("k", [21, 31])
def test_file_image_to_summary(ctx, orc, k):
    """the chain a caller runs on real data: FASTQ image -> kmx_fastx_parse -> kmx_canonical_reduce (ragged reads)"""
    from kmers_amd import _lib

    rng = np.random.default_rng(k)
    text = fastq_text(rng, 20000, 30, 160)
    bases, offsets, eb, eo = _check(ctx, orc, text)
    n = len(eo) - 1
    g = ctx.canonical_reduce(bases, n, 160, k, _lib.HASH_LEX, k, 0, offsets=offsets)
    o = orc.canonical_reduce(eb, n, 0, k, hasher_k=k, offsets=eo)
    assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)


def test_same_text_flag_only_reuses_what_it_may(ctx, orc):
    """KMX_FASTX_SAME_TEXT on the emit call reuses the counting call's chunk summaries; another image, another size, a
    histogram call in between (it overwrites the work buffer) or no counting call at all must fall back to a full parse"""
    import ctypes as C

    import torch

    from kmers_amd import _lib
    from kmers_amd.api import _ptr, u64_numpy

    rng = np.random.default_rng(77)
    ta, tb = fastq_text(rng, 3000, 20, 200), fastq_text(rng, 2500, 30, 180)
    da, db = ctx.to_device(ta), ctx.to_device(tb)

    def count(d):
        nr, nb = C.c_uint64(0), C.c_uint64(0)
        ctx._ck(ctx.lib.kmx_fastx_parse(ctx._h, _ptr(d), d.numel(), 0, None, None, 0, C.byref(nr), C.byref(nb)))
        return nr.value, nb.value

    def emit(d, n_bytes, nr, nb, flag):
        bases, offsets = ctx.empty(max(nb, 1), torch.uint8), ctx.empty(nr + 1, torch.int64)
        r, b = C.c_uint64(0), C.c_uint64(0)
        ctx._ck(ctx.lib.kmx_fastx_parse(ctx._h, _ptr(d), n_bytes, flag, _ptr(bases), _ptr(offsets), nr, C.byref(r), C.byref(b)))
        return bases[:b.value].cpu().numpy(), u64_numpy(offsets)[: r.value + 1]

    ea, eb = orc.fastx_parse(ta, 0), orc.fastx_parse(tb, 0)
    nra, nba = count(da)
    got = emit(db, db.numel(), len(eb[1]) - 1, len(eb[0]), _lib.FASTX_SAME_TEXT)           # another image
    assert np.array_equal(got[0], eb[0]) and np.array_equal(got[1], eb[1])
    nra, nba = count(da)
    ctx.histogram(ctx.gen_reads(150 * 8192), 8192, 150, 31, 1, 31, 20)                     # the work buffer is overwritten
    got = emit(da, da.numel(), nra, nba, _lib.FASTX_SAME_TEXT)
    assert np.array_equal(got[0], ea[0]) and np.array_equal(got[1], ea[1])
    nra, nba = count(da)
    cut = int(ea[1][1500]) * 0 + (len(ta) // 2)                                            # another size: a prefix of the image
    ec = orc.fastx_parse(ta[:cut], 0)
    got = emit(da, cut, len(ec[1]) - 1, len(ec[0]), _lib.FASTX_SAME_TEXT)
    assert np.array_equal(got[0], ec[0]) and np.array_equal(got[1], ec[1])
    nra, nba = count(da)
    got = emit(da, da.numel(), nra, nba, _lib.FASTX_SAME_TEXT)                             # the reuse itself
    assert np.array_equal(got[0], ea[0]) and np.array_equal(got[1], ea[1])


def test_fastx_reads_picks_the_layout(ctx, orc):
    """Context.fastx_reads: uniform layout when every read has the same length, else ragged with the tight bound; the
    summaries equal the oracle's over the parsed reads either way"""
    from kmers_amd import _lib

    rng = np.random.default_rng(5)
    for fixed in (150, None):
        text = fastq_text(rng, 700, fixed=fixed) if fixed else fastq_text(rng, 700, 40, 150)
        eb, eo = orc.fastx_parse(text, 0)
        bases, n, L, offsets = ctx.fastx_reads(ctx.to_device(text))
        assert n == len(eo) - 1 and L == int(np.diff(eo.astype(np.int64)).max())
        assert (offsets is None) == (fixed is not None)
        o = orc.canonical_reduce(eb, n, 0, 21, hasher_k=21, offsets=eo)
        g = ctx.canonical_reduce(bases, n, L, 21, _lib.HASH_LEX, 21, 0, offsets=offsets)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)


@pytest.mark.parametrize("shift", [1, 3, 8, 13])
def test_output_buffer_at_any_alignment(ctx, orc, shift):
    """pass 3 stages a row's bytes in LDS at their offset modulo 16 in MEMORY and writes whole aligned 16-byte pieces: the caller's
    d_bases need not be aligned, and nothing may be written before or behind the n_bases bytes"""
    import ctypes as C

    import torch

    from kmers_amd.api import _ptr, u64_numpy

    rng = np.random.default_rng(90 + shift)
    for text, fmt in ((fastq_text(rng, 3000, 0, 180), 1), (fasta_text(rng, 300, 0, 3000, width=61), 2)):
        eb, eo = orc.fastx_parse(text, fmt)
        d = ctx.to_device(text)
        buf = torch.full((len(eb) + 64,), 0xEE, dtype=torch.uint8, device=d.device)
        bases = buf[shift : shift + len(eb)]
        offsets = torch.empty(len(eo), dtype=torch.int64, device=d.device)
        nr, nb = C.c_uint64(0), C.c_uint64(0)
        ctx._ck(ctx.lib.kmx_fastx_parse(ctx._h, _ptr(d), d.numel(), fmt, _ptr(bases), _ptr(offsets), len(eo) - 1, C.byref(nr), C.byref(nb)))
        assert nr.value == len(eo) - 1 and nb.value == len(eb)
        assert np.array_equal(u64_numpy(offsets), eo)
        assert np.array_equal(bases.cpu().numpy(), eb)
        assert bool((buf[:shift] == 0xEE).all()) and bool((buf[shift + len(eb) :] == 0xEE).all())
