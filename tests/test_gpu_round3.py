"""Round-3 GPU parity tests (all through the C ABI, bit-exact against the CPU oracle):
  * the k = 31 scan on the inputs written for round 3's producer/consumer variant (removed in round 4: measured 1-3 % slower;
    the cases stay): clean, dirty, unaligned base, partial tiles, every read length of the 10-word frame with 4 windows per lane;
  * the error path of kmx_comm_create: two ranks on ONE device -- RCCL refuses the duplicate GPU, both processes must come
    back with KMX_E_HIP and a text in kmx_last_error, without hanging;
  * libkmx.so loads without librccl on the link line (RCCL is resolved at the first kmx_comm_* call)."""
import os
import subprocess
import sys

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


def _same(g, o, want_hash, want_sumfw):
    assert g.n_valid == o.n_valid
    assert g.sum_canon == o.sum_canon
    assert g.xor_hash == (o.xor_hash if want_hash else 0)
    assert g.sum_fw == (o.sum_fw if want_sumfw else 0)


@pytest.mark.parametrize("n_reads", [1, 63, 64, 65, 64 * 7 + 5, 20_000, 200_003])
def test_scan_r3_clean_reads(ctx, orc, n_reads):
    from kmers_amd import _lib

    L, k = 150, 31
    bases = ctx.gen_reads(n_reads * L, first_byte=L * 777)
    o = orc.canonical_reduce(bases.cpu().numpy(), n_reads, L, k, hasher_k=k)
    _same(ctx.canonical_reduce(bases, n_reads, L, k), o, False, False)
    _same(ctx.canonical_reduce(bases, n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW), o, True, True)
    assert o.n_valid == n_reads * (L - k + 1)


@pytest.mark.parametrize("L", [127, 128, 131, 140, 149, 151, 157, 158])
def test_scan_r3_every_length_of_the_frame(ctx, orc, L):
    """k = 31 with 97..128 windows per read: 4 windows per lane, four window blocks of pass 2"""
    from kmers_amd import _lib

    k, n_reads = 31, 64 * 11 + 9
    bases = ctx.gen_reads(n_reads * L, first_byte=4242)
    o = orc.canonical_reduce(bases.cpu().numpy(), n_reads, L, k, hasher_k=k)
    _same(ctx.canonical_reduce(bases, n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW), o, True, True)


@pytest.mark.parametrize("p_bad", [0.0002, 0.002, 0.02])
def test_scan_r3_dirty_reads(ctx, orc, p_bad):
    """reads with an invalid byte: blanked in phase B, masked out of m, handled by sweep_flagged_kernel"""
    from kmers_amd import _lib

    L, k, n_reads = 150, 31, 64 * 300 + 17
    rng = np.random.default_rng(31 + int(p_bad * 1e5))
    host = _dirty(rng, n_reads * L, p_bad)
    o = orc.canonical_reduce(host, n_reads, L, k, hasher_k=k)
    g = ctx.canonical_reduce(ctx.to_device(host), n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _same(g, o, True, True)
    assert o.n_valid < n_reads * (L - k + 1)
    # and again: the masks the main pass leaves behind are cleared by the roll kernel, nothing carries over between calls
    g = ctx.canonical_reduce(ctx.to_device(host), n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _same(g, o, True, True)


@pytest.mark.parametrize("lead", [1, 5, 8, 15])
def test_scan_r3_unaligned_base(ctx, orc, lead):
    from kmers_amd import _lib

    L, k, n_reads = 150, 31, 64 * 40 + 3
    rng = np.random.default_rng(lead)
    host = _dirty(rng, lead + n_reads * L, 0.0005)
    dev = ctx.to_device(host)
    o = orc.canonical_reduce(host[lead:], n_reads, L, k, hasher_k=k)
    g = ctx.canonical_reduce(dev[lead:], n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    _same(g, o, True, True)


def test_scan_r3_repeatable_at_size(ctx):
    """2e6 reads (31 250 tiles: every wave runs many tiles; the accumulators of pass 2 are folded and carried): two calls, same summary"""
    from kmers_amd import _lib

    L, k, n_reads = 150, 31, 2_000_000
    bases = ctx.gen_reads(n_reads * L)
    g = ctx.canonical_reduce(bases, n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    r = ctx.canonical_reduce(bases, n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
    assert (g.n_valid, g.sum_canon, g.xor_hash, g.sum_fw) == (r.n_valid, r.sum_canon, r.xor_hash, r.sum_fw)
    assert g.n_valid == n_reads * (L - k + 1)


# ------------------------------------------------------------------ kmx_comm_create: the error path

_CHILD = r"""
import sys, ctypes as C
sys.path.insert(0, sys.argv[1])
from kmers_amd import _lib
from kmers_amd.api import Context
rank, id_path = int(sys.argv[2]), sys.argv[3]
ctx = Context(0)
raw = open(id_path, "rb").read()
buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(raw)
h = C.c_void_p()
st = ctx.lib.kmx_comm_create(ctx._h, buf, 2, rank, C.byref(h))
txt = ctx.lib.kmx_last_error(ctx._h)
print("STATUS", st, "HANDLE", h.value, "TEXT", (txt or b"").decode(), flush=True)
# the context is still usable after the failed create (its stream was not leaked or left in error)
import torch
b = ctx.gen_reads(150 * 64)
s = ctx.canonical_reduce(b, 64, 150, 31)
print("AFTER", s.n_valid, flush=True)
ctx.close()
"""


def test_comm_create_refuses_two_ranks_on_one_gpu(tmp_path):
    import ctypes as C

    from kmers_amd import _lib

    lib = _lib.load()
    buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    assert lib.kmx_comm_get_unique_id(buf) == 0
    idp = tmp_path / "id.bin"
    idp.write_bytes(bytes(buf))
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(idp)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()     # exactly the two children started above
            pytest.fail("kmx_comm_create hung with two ranks on one device")
        outs.append((p.returncode, out, err))
    for rc, out, err in outs:
        assert rc == 0, (out[-1500:], err[-1500:])
        # (RCCL writes its own warnings to stdout as well, not always newline-terminated: search, do not split)
        import re

        m = re.search(r"STATUS (\d+) HANDLE (\S+) TEXT ([^\n]*)", out)
        assert m, out[-1500:]
        assert int(m.group(1)) == _lib.E_HIP                                                # KMX_E_HIP
        assert m.group(2) in ("None", "0")                                                  # no handle came back
        assert "RCCL" in m.group(3) and len(m.group(3)) > 10                                # kmx_last_error names the failure
        assert "AFTER 7680" in out                                                          # 64 reads x 120 windows: the context still works


def test_libkmx_does_not_link_rccl():
    """single-GPU hosts load libkmx.so without librccl: RCCL is dlopen'ed by the first kmx_comm_* call"""
    so = os.path.join(ROOT, "kmers_amd", "libkmx.so")
    r = subprocess.run(["readelf", "-d", so], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("readelf not available")
    needed = [ln for ln in r.stdout.splitlines() if "NEEDED" in ln]
    assert needed and not any("rccl" in ln for ln in needed), needed


# ------------------------------------------------------------------ offsets array whose reads are in fact uniform

@pytest.mark.parametrize("k", [13, 21, 31])
@pytest.mark.parametrize("case", ["uniform", "one_trimmed", "shifted_start", "loose_bound", "dirty_uniform"])
def test_reduce_picks_the_uniform_kernel_on_the_device(ctx, orc, k, case):
    """kmx_canonical_reduce with an offsets array and a length bound: a device-side check of offsets[i] == i*L gates the
    uniform and the ragged scan (both are launched, one runs).  Whatever it picks, the summary is the oracle's."""
    from kmers_amd import _lib

    L, n_reads = 150, 64 * 50 + 11
    rng = np.random.default_rng(k * 7 + len(case))
    lens = np.full(n_reads, L, np.int64)
    start = 0
    hint = L
    if case == "one_trimmed":
        lens[n_reads // 2] = L - 7
    elif case == "shifted_start":
        start = 16
    elif case == "loose_bound":
        hint = 160
    offsets = (start + np.concatenate([[0], np.cumsum(lens)])).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]), 0.001 if case == "dirty_uniform" else 0.0)
    o = orc.canonical_reduce(host, n_reads, 0, k, hasher_k=k, offsets=offsets)
    g = ctx.canonical_reduce(ctx.to_device(host), n_reads, hint, k, _lib.HASH_LEX, k, 0, offsets=ctx.to_device(offsets))
    _same(g, o, True, False)
    # and a second call right after (the gate word does not stay armed, the queue heads start from zero again)
    g = ctx.canonical_reduce(ctx.to_device(host), n_reads, hint, k, _lib.HASH_NONE, 0, 0, offsets=ctx.to_device(offsets))
    _same(g, o, False, False)
    if case in ("uniform", "dirty_uniform"):
        u = ctx.canonical_reduce(ctx.to_device(host), n_reads, L, k, _lib.HASH_LEX, k, 0)
        assert (u.n_valid, u.sum_canon, u.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)


# ------------------------------------------------------------------ histogram, 2^23..2^28 buckets: two partition levels

@pytest.mark.parametrize("b", [23, 24, 26])
@pytest.mark.parametrize("hasher,hk", [(1, 31), (2, 0), (1, 17)])
def test_histogram_two_levels(ctx, orc, b, hasher, hk):
    """2^23..2^28 buckets took device atomics before (0.5 s per 1e8 reads); now pass 1 writes whole buckets (u32), a second
    pass splits each partition by the next six bits, the third counts in LDS.  Bit-exact against the oracle's table."""
    if b == 26 and hasher != 1:
        pytest.skip("one hasher at the 512 MB table")
    L, k, n_reads = 150, 31, 64 * 700 + 29
    rng = np.random.default_rng(b * 11 + hasher)
    host = _dirty(rng, n_reads * L, 0.0004)
    o = orc.histogram(host, n_reads, L, k, hk if hasher == 1 else 0, b)
    g = ctx.histogram(ctx.to_device(host), n_reads, L, k, hasher, hk, b).cpu().numpy().view(np.uint64)
    assert int(g.sum()) == int(o.sum())
    assert (g == o).all()


def test_histogram_two_levels_at_2_28_folds_to_2_23(ctx, orc):
    """the 2 GB table of 2^28 buckets against the oracle's 2^23 table: both are top bits of the same 32-bit mix"""
    L, k, n_reads = 150, 31, 64 * 900 + 3
    bases = ctx.gen_reads(n_reads * L, first_byte=12345)
    o = orc.histogram(bases.cpu().numpy(), n_reads, L, k, k, 23)
    g = ctx.histogram(bases, n_reads, L, k, 1, k, 28)
    folded = g.view(-1, 32).sum(dim=1).cpu().numpy().view(np.uint64)
    assert int(folded.sum()) == n_reads * (L - k + 1)
    assert (folded == o).all()


def test_histogram_two_levels_ragged(ctx, orc):
    L_hint, k, b = 160, 21, 24
    rng = np.random.default_rng(5)
    lens = rng.integers(20, 161, size=64 * 400 + 7)
    lens[::97] = 0
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]) + 16, 0.0005)[: int(offsets[-1])]
    o = orc.histogram(host, len(lens), 0, k, k, b, offsets=offsets)
    g = ctx.histogram(ctx.to_device(host), len(lens), L_hint, k, 1, k, b, offsets=ctx.to_device(offsets))
    g = g.cpu().numpy().view(np.uint64)
    assert int(g.sum()) == int(o.sum())
    assert (g == o).all()


# ------------------------------------------------------------------ [u64;2] materialise: the tiled kernel

def _cmp_windows2(ctx, orc, host, dev, n, L, k):
    fw, rc, canon, flags = orc.canonical_windows2(host, n, L, k)
    outs = ctx.canonical_windows2(dev, n, L, k)
    assert (outs["flags"].cpu().numpy() == flags).all()
    assert (outs["fw"].cpu().numpy().view(np.uint64).reshape(-1, 2) == fw).all()
    assert (outs["rc"].cpu().numpy().view(np.uint64).reshape(-1, 2) == rc).all()
    assert (outs["canon"].cpu().numpy().view(np.uint64).reshape(-1, 2) == canon).all()


@pytest.mark.parametrize("k", [33, 34, 40, 47, 48, 49, 56, 63, 64])
@pytest.mark.parametrize("L", [64, 65, 100, 150, 159, 160, 161, 200, 256])
def test_windows2_tiled_every_frame(ctx, orc, k, L):
    """kmx_canonical_windows2 on uniform reads: the tiled kernel (kmx_generic.hip: rolling 128-bit windows out of the packed
    tile) against the oracle's per-read rolling -- clean reads, 64 * n + tail reads, both frames"""
    n = 64 * 6 + 13
    bases = ctx.gen_reads(n * L, first_byte=k * 1000 + L)
    _cmp_windows2(ctx, orc, bases.cpu().numpy(), bases, n, L, k)


@pytest.mark.parametrize("k", [33, 47, 63, 64])
@pytest.mark.parametrize("p_bad", [0.0003, 0.01])
def test_windows2_tiled_dirty(ctx, orc, k, p_bad):
    """tiles with an invalid byte take the exact per-read body; slots of skipped windows are zero with flags 0"""
    n, L = 64 * 20 + 5, 150
    rng = np.random.default_rng(k + int(p_bad * 1e4))
    host = _dirty(rng, n * L, p_bad)
    _cmp_windows2(ctx, orc, host, ctx.to_device(host), n, L, k)


@pytest.mark.parametrize("lead", [1, 7, 8, 15])
def test_windows2_tiled_unaligned_base(ctx, orc, lead):
    n, L, k = 64 * 5 + 1, 150, 63
    rng = np.random.default_rng(lead)
    host = _dirty(rng, lead + n * L, 0.0002)
    dev = ctx.to_device(host)
    _cmp_windows2(ctx, orc, host[lead:], dev[lead:], n, L, k)


def test_windows2_single_output_arrays(ctx, orc):
    """any of the four outputs may be NULL"""
    import ctypes as C

    import torch

    from kmers_amd.api import _ptr

    n, L, k = 64 * 9, 150, 47
    bases = ctx.gen_reads(n * L, first_byte=5)
    fw, rc, canon, flags = orc.canonical_windows2(bases.cpu().numpy(), n, L, k)
    tot = n * (L - k + 1)
    r = ctx._reads(bases, n, L, None)
    for which, ref in (("canon", canon), ("fw", fw), ("rc", rc)):
        buf = ctx.empty(2 * tot, torch.int64)
        args = {"fw": None, "rc": None, "canon": None}
        args[which] = _ptr(buf)
        ctx._ck(ctx.lib.kmx_canonical_windows2(ctx._h, C.byref(r), None, k, args["fw"], args["rc"], args["canon"], None))
        torch.cuda.synchronize()
        assert (buf.cpu().numpy().view(np.uint64).reshape(-1, 2) == ref).all(), which
    fl = ctx.empty(tot, torch.uint8)
    ctx._ck(ctx.lib.kmx_canonical_windows2(ctx._h, C.byref(r), None, k, None, None, None, _ptr(fl)))
    torch.cuda.synchronize()
    assert (fl.cpu().numpy() == flags).all()


# ------------------------------------------------------------------ the 13-word frame (reads of 161..208 bases)

@pytest.mark.parametrize("L", [161, 163, 170, 176, 185, 192, 200, 207, 208, 209])
@pytest.mark.parametrize("k", [13, 16, 21, 27, 31])
def test_reduce_thirteen_word_frame(ctx, orc, L, k):
    """uniform reads of 161..208 bases take the 13-word frame of the bit-sliced kernel (5, 6 or 7 windows per lane); 209 is the
    first length of the 16-word frame again.  Clean and dirty reads, an unaligned base, against the oracle."""
    from kmers_amd import _lib

    n_reads = 64 * 9 + 21
    rng = np.random.default_rng(L * 31 + k)
    for p_bad, lead in ((0.0, 0), (0.0004, 0), (0.0, 9)):
        host = _dirty(rng, lead + n_reads * L, p_bad)
        dev = ctx.to_device(host)
        o = orc.canonical_reduce(host[lead:], n_reads, L, k, hasher_k=k)
        g = ctx.canonical_reduce(dev[lead:], n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _same(g, o, True, True)


# ------------------------------------------------------------------ a ragged read of 2^31 bases is diagnosed

def test_read_of_2_31_bases_is_skipped_and_reported(ctx, orc):
    """kmx.h "Limits": positions are i32 like the reference's (canonical_kmer_iterator.rs:15).  A ragged read of >= 2^31
    bases used to be scanned with wrapped positions, silently; now the scans skip it, the summary is that of the other
    reads, kmx_ctx_synchronize returns KMX_E_ARG once (sticky flag), and kmx_reads_length_range refuses the batch."""
    import torch

    from kmers_amd import _lib
    from kmers_amd._lib import KmxError

    k, small = 31, 64 * 3 + 5
    huge = (1 << 31) + 160
    rng = np.random.default_rng(3)
    lens = np.full(small + 1, 150, np.int64)
    lens[70] = huge                                   # one read past the limit, in the middle of a tile
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    dev = torch.full((int(offsets[-1]) + 64,), ord("A"), dtype=torch.uint8, device=ctx.device)
    ok_reads = _dirty(rng, small * 150, 0.0)
    # place the ordinary reads where the offsets say they are
    pos = 0
    chunks = []
    for r in range(small + 1):
        if r == 70:
            continue
        o0 = int(offsets[r])
        dev[o0:o0 + 150] = torch.from_numpy(ok_reads[pos:pos + 150].copy()).to(ctx.device)
        pos += 150
    d_off = ctx.to_device(offsets)
    ctx.synchronize()                                  # clean slate
    for hint in (150, 0):
        g = ctx.canonical_reduce(dev, small + 1, hint, k, _lib.HASH_LEX, k, 0, offsets=d_off)
        o = orc.canonical_reduce(ok_reads, small, 150, k, hasher_k=k)
        assert (g.n_valid, g.sum_canon, g.xor_hash) == (o.n_valid, o.sum_canon, o.xor_hash)
        with pytest.raises(KmxError) as ei:
            ctx.synchronize()
        assert ei.value.status == _lib.E_ARG
        ctx.synchronize()                              # the flag was cleared by the report
    with pytest.raises(KmxError) as ei:
        ctx.reads_length_range(d_off)
    assert ei.value.status == _lib.E_ARG
    # the histogram and the per-read kernels take the same exit
    h = ctx.histogram(dev, small + 1, 150, k, 1, k, 12, offsets=d_off)
    assert int(h.sum().item()) == small * 120
    with pytest.raises(KmxError):
        ctx.synchronize()
    del dev


# ------------------------------------------------------------------ the 5-word frame (reads of up to 80 bases)

@pytest.mark.parametrize("L", [31, 36, 50, 64, 65, 75, 76, 79, 80, 81])
@pytest.mark.parametrize("k", [13, 17, 21, 31])
def test_reduce_five_word_frame(ctx, orc, L, k):
    """uniform reads of up to 80 bases take the 5-word frame (2 or 3 windows per lane, 5 waves/SIMD); 81 is the first length
    of the 7-word frame.  Clean and dirty reads, an unaligned base, against the oracle."""
    from kmers_amd import _lib

    if L < k:
        pytest.skip("read shorter than k")
    n_reads = 64 * 11 + 3
    rng = np.random.default_rng(L * 131 + k)
    for p_bad, lead in ((0.0, 0), (0.001, 0), (0.0, 5)):
        host = _dirty(rng, lead + n_reads * L, p_bad)
        dev = ctx.to_device(host)
        o = orc.canonical_reduce(host[lead:], n_reads, L, k, hasher_k=k)
        g = ctx.canonical_reduce(dev[lead:], n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _same(g, o, True, True)


# ------------------------------------------------------------------ the 8-word frame (reads of 113..128 bases)

@pytest.mark.parametrize("L", [112, 113, 120, 125, 126, 127, 128, 129])
@pytest.mark.parametrize("k", [13, 19, 25, 31])
def test_reduce_eight_word_frame(ctx, orc, L, k):
    """uniform reads of 113..128 bases take the 8-word frame (3 or 4 windows per lane); 112 is the last length of the 7-word
    frame and 129 the first of the 10-word one.  Clean and dirty reads, an unaligned base, against the oracle."""
    from kmers_amd import _lib

    n_reads = 64 * 10 + 33
    rng = np.random.default_rng(L * 17 + k)
    for p_bad, lead in ((0.0, 0), (0.0006, 0), (0.0, 13)):
        host = _dirty(rng, lead + n_reads * L, p_bad)
        dev = ctx.to_device(host)
        o = orc.canonical_reduce(host[lead:], n_reads, L, k, hasher_k=k)
        g = ctx.canonical_reduce(dev[lead:], n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _same(g, o, True, True)


# ------------------------------------------------------------------ the rolling kernel of the blanked reads
@pytest.mark.parametrize("n_dirty", [1, 3, 8, 9, 16, 17, 32, 33, 64, 65, 200])
@pytest.mark.parametrize("k,ragged", [(31, False), (21, False), (31, True), (63, False)])
def test_rolled_reads_shared_by_lanes(ctx, orc, n_dirty, k, ragged):
    """the kernel behind the blanked reads (round 3: roll_flagged_kernel, 8 / 4 / 2 lanes sharing a read; round 6:
    sweep_flagged_kernel, 64 reads as a tile of the word domain) with 1 .. 200 reads to gather, one set of atomics per block:
    single- and two-word k, uniform and ragged reads, invalid bytes at the ends, in the middle and twice in a read -- against
    the oracle"""
    from kmers_amd import _lib

    L, n_reads = 150, 64 * 40          # 40 tiles: their masks are one group, gathered by one wave
    rng = np.random.default_rng(1000 * k + n_dirty + (7 if ragged else 0))
    host = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_reads * L)].copy()
    for i, r in enumerate(rng.choice(n_reads, n_dirty, replace=False)):
        for p in ((0,), (L - 1,), (L // 2,), (k - 1, L - k), (int(rng.integers(0, L)),))[i % 5]:
            host[r * L + p] = ord("N")
    dev = ctx.to_device(host)
    if ragged:
        lens = np.full(n_reads, L, np.int64)
        lens[:: 97] -= 1 + (np.arange(len(lens[:: 97])) % 60)     # some reads shorter: the reads stay where they are, gaps between them are not allowed
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        hostr = np.concatenate([host[r * L : r * L + lens[r]] for r in range(n_reads)])
        o = orc.canonical_reduce(hostr, n_reads, L, k, hasher_k=k, offsets=offs)
        g = ctx.canonical_reduce(ctx.to_device(hostr), n_reads, L, k, _lib.HASH_LEX, k, 0, offsets=ctx.to_device(offs))
        _same(g, o, True, False)
    elif k <= 32:
        o = orc.canonical_reduce(host, n_reads, L, k, hasher_k=k)
        g = ctx.canonical_reduce(dev, n_reads, L, k, _lib.HASH_LEX, k, _lib.REDUCE_SUM_FW)
        _same(g, o, True, True)
    else:
        o = orc.canonical_reduce2(host, n_reads, L, k, with_hash=True)
        g = ctx.canonical_reduce2(dev, n_reads, L, k, with_hash=True)
        assert (g.n_valid, g.sum_lo, g.sum_hi, g.xor_lo, g.xor_hi) == (o.n_valid, o.sum_lo, o.sum_hi, o.xor_lo, o.xor_hi)


# ------------------------------------------------------------------ sum_fw of ragged reads on the bit-sliced kernel
@pytest.mark.parametrize("k", [13, 16, 21, 27, 31])
@pytest.mark.parametrize("case", ["150_trimmed", "mix_100_160", "short_20_100", "long_161_250", "tiny_and_empty", "dirty"])
def test_ragged_sum_fw(ctx, orc, k, case):
    """KMX_REDUCE_SUM_FW for reads behind an offsets array: the closed form of the bit-sliced kernel (plane totals weighted by
    position from the start, minus the totals of every read's last k-1 bases weighted by position from the end) against the
    oracle's per-read rolling -- every frame, reads shorter than k and than 2k-1, empty reads, reads with an N"""
    from kmers_amd import _lib

    rng = np.random.default_rng(k * 11 + len(case))
    n_reads = 64 * 30 + 17
    if case == "150_trimmed":
        lens, hint = np.where(rng.random(n_reads) < 0.1, rng.integers(36, 150, n_reads), 150), 150
    elif case == "mix_100_160":
        lens, hint = rng.integers(100, 161, n_reads), 160
    elif case == "short_20_100":
        lens, hint = rng.integers(20, 101, n_reads), 100
    elif case == "long_161_250":
        lens, hint = rng.integers(161, 251, n_reads), 250
    elif case == "tiny_and_empty":
        lens, hint = rng.integers(0, 2 * k + 3, n_reads), 0
    else:
        lens, hint = rng.integers(60, 151, n_reads), 150
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    host = _dirty(rng, int(offsets[-1]) + 16, 0.002 if case == "dirty" else 0.0)[: int(offsets[-1])]
    o = orc.canonical_reduce(host, n_reads, 0, k, hasher_k=k, offsets=offsets)
    g = ctx.canonical_reduce(ctx.to_device(host) if len(host) else ctx.empty(0, __import__("torch").uint8), n_reads, hint, k, _lib.HASH_LEX, k,
                             _lib.REDUCE_SUM_FW, offsets=ctx.to_device(offsets))
    _same(g, o, True, True)
