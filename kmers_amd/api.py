"""Batch API over the libkmx C ABI with torch tensors as device buffers.

torch is plumbing here (device memory, streams, torch.distributed); every computation is a
hand-written HIP kernel behind include/kmx.h.  All functions run on `Context.stream` (by
default torch's current stream of the device at construction) and are asynchronous like any torch CUDA op.

Stream discipline: kmx kernels are enqueued on `Context.stream`.  Every method makes that stream torch's current
stream for its whole body, so the buffers it allocates, the launch and the device-to-host read-back are ordered on ONE
stream -- also when the context was built on a side stream, or is used inside `with torch.cuda.stream(other)`.  Tensors
handed IN by the caller must be ready on `Context.stream` (produced there, or after a `wait_stream`): they are marked
with `record_stream` so the caching allocator does not recycle them under a running kmx kernel.
"""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np
import torch

from . import _lib
from ._lib import (HASH_IDENTITY, HASH_LEX, HASH_NONE, REDUCE_SUM_FW, KmxError, Reads, Summary, Summary2)

SEED_DEFAULT = 0x6B6D6572735F7631  # "kmers_v1"


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("kmx expects contiguous CUDA tensors")
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(int(t))


def u64_numpy(t: torch.Tensor) -> np.ndarray:
    """int64 CUDA tensor holding u64 words -> numpy uint64 (host)."""
    return t.detach().cpu().numpy().view(np.uint64)


def _on_ctx_stream(fn):
    """Run a Context method with Context.stream as torch's current stream (see "Stream discipline" above)."""
    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == self.stream.cuda_stream:
            return fn(self, *args, **kwargs)
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                a.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            return fn(self, *args, **kwargs)
    return wrapped


class Context:
    """One kmx_ctx bound to a device and a HIP stream (borrowed from torch)."""

    def __init__(self, device: int | torch.device | None = None, stream: torch.cuda.Stream | None = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise KmxError(_lib.E_HIP, "no HIP device visible: kmers_amd is GPU-only (no CPU fallback)")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        with torch.cuda.device(self.device):
            self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        st = self.lib.kmx_ctx_create_on_stream(self.device.index, C.c_void_p(self.stream.cuda_stream), C.byref(h))
        _lib.check(self.lib, None, st)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.kmx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        _lib.check(self.lib, self._h, st)

    def synchronize(self):
        self._ck(self.lib.kmx_ctx_synchronize(self._h))

    def set_work_buffer_limit(self, nbytes: int):
        """cap (bytes; 0 = automatic) on the context's device work buffer: kmx_ctx_set_work_buffer_limit"""
        self._ck(self.lib.kmx_ctx_set_work_buffer_limit(self._h, int(nbytes)))

    def work_buffer_info(self):
        """(bytes held, number of (re)allocations so far): kmx_ctx_work_buffer_info"""
        held, n = C.c_size_t(0), C.c_uint64(0)
        self._ck(self.lib.kmx_ctx_work_buffer_info(self._h, C.byref(held), C.byref(n)))
        return int(held.value), int(n.value)

    # ------------------------------------------------------------- helpers
    @_on_ctx_stream
    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    @_on_ctx_stream
    def to_device(self, a) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(self.device).contiguous()
        if isinstance(a, (bytes, bytearray)):
            a = np.frombuffer(bytes(a), dtype=np.uint8)
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        return torch.from_numpy(a.copy()).to(self.device)

    def _reads(self, bases: torch.Tensor, n_reads: int, read_len: int, offsets: torch.Tensor | None) -> Reads:
        return Reads(_ptr(bases) if bases is not None and bases.numel() else None, int(n_reads), int(read_len),
                     _ptr(offsets))

    # ------------------------------------------------------------ hot path
    @_on_ctx_stream
    def gen_reads(self, nbytes: int, seed: int = SEED_DEFAULT, first_byte: int = 0, out: torch.Tensor | None = None):
        """Deterministic synthetic ACGT stream (kmx_gen_reads)."""
        if out is None:
            out = self.empty(nbytes, torch.uint8)
        self._ck(self.lib.kmx_gen_reads(self._h, seed & (2**64 - 1), first_byte, _ptr(out), int(nbytes)))
        return out

    @_on_ctx_stream
    def canonical_reduce_async(self, bases, n_reads, read_len, k, hasher=HASH_NONE, hasher_k=0, flags=0, offsets=None,
                               out: torch.Tensor | None = None) -> torch.Tensor:
        """kmx_canonical_reduce; returns the device-resident summary (4 x int64 viewable as u64)."""
        if out is None:
            out = self.empty(4, torch.int64)
        r = self._reads(bases, n_reads, read_len, offsets)
        self._ck(self.lib.kmx_canonical_reduce(self._h, C.byref(r), k, hasher, hasher_k, flags, _ptr(out)))
        return out

    @_on_ctx_stream
    def canonical_reduce(self, bases, n_reads, read_len, k, hasher=HASH_NONE, hasher_k=0, flags=0, offsets=None) -> Summary:
        out = self.canonical_reduce_async(bases, n_reads, read_len, k, hasher, hasher_k, flags, offsets)
        v = u64_numpy(out)
        return Summary(int(v[0]), int(v[1]), int(v[2]), int(v[3]))

    @_on_ctx_stream
    def canonical_reduce_host(self, bases, n_reads, read_len, k, hasher=HASH_NONE, hasher_k=0, flags=0, offsets=None) -> Summary:
        """kmx_canonical_reduce_host: the summary in host memory when the call returns (one launch for clean uniform reads)."""
        out = Summary()
        r = self._reads(bases, n_reads, read_len, offsets)
        self._ck(self.lib.kmx_canonical_reduce_host(self._h, C.byref(r), k, hasher, hasher_k, flags, C.byref(out)))
        return out

    def win_offsets(self, n_reads, read_len, k, offsets=None) -> np.ndarray:
        if offsets is None:
            w = max(read_len - k + 1, 0)
            return np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(w)
        lens = np.diff(np.asarray(offsets).astype(np.int64))
        return np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))]).astype(np.uint64)

    @_on_ctx_stream
    def canonical_windows(self, bases, n_reads, read_len, k, offsets=None, host_offsets=None, want=("fw", "rc", "canon", "flags")):
        """kmx_canonical_windows -> dict of device tensors (u64 words as int64, flags uint8)."""
        wo_host = self.win_offsets(n_reads, read_len, k, host_offsets)
        total = int(wo_host[-1])
        d_wo = self.to_device(wo_host) if offsets is not None else None
        outs = {n: (self.empty(total, torch.uint8) if n == "flags" else self.empty(total, torch.int64)) for n in want}
        r = self._reads(bases, n_reads, read_len, offsets)
        self._ck(self.lib.kmx_canonical_windows(self._h, C.byref(r), _ptr(d_wo), k, _ptr(outs.get("fw")), _ptr(outs.get("rc")),
                                                _ptr(outs.get("canon")), _ptr(outs.get("flags"))))
        return outs

    @_on_ctx_stream
    def canonical_reduce2(self, bases, n_reads, read_len, k, with_hash=False, offsets=None) -> Summary2:
        out = self.empty(5, torch.int64)
        r = self._reads(bases, n_reads, read_len, offsets)
        self._ck(self.lib.kmx_canonical_reduce2(self._h, C.byref(r), k, int(with_hash), _ptr(out)))
        v = u64_numpy(out)
        return Summary2(*[int(x) for x in v])

    @_on_ctx_stream
    def canonical_windows2(self, bases, n_reads, read_len, k, offsets=None, host_offsets=None):
        wo_host = self.win_offsets(n_reads, read_len, k, host_offsets)
        total = int(wo_host[-1])
        d_wo = self.to_device(wo_host) if offsets is not None else None
        outs = {n: self.empty(2 * total, torch.int64) for n in ("fw", "rc", "canon")}
        outs["flags"] = self.empty(total, torch.uint8)
        r = self._reads(bases, n_reads, read_len, offsets)
        self._ck(self.lib.kmx_canonical_windows2(self._h, C.byref(r), _ptr(d_wo), k, _ptr(outs["fw"]), _ptr(outs["rc"]),
                                                 _ptr(outs["canon"]), _ptr(outs["flags"])))
        return outs

    @_on_ctx_stream
    def histogram(self, bases, n_reads, read_len, k, hasher, hasher_k, log2_buckets, offsets=None,
                  counts: torch.Tensor | None = None) -> torch.Tensor:
        if counts is None:
            counts = torch.zeros(1 << log2_buckets, dtype=torch.int64, device=self.device)
        r = self._reads(bases, n_reads, read_len, offsets)
        self._ck(self.lib.kmx_histogram(self._h, C.byref(r), k, hasher, hasher_k, log2_buckets, _ptr(counts)))
        return counts

    # --------------------------------------------------------- element-wise
    @_on_ctx_stream
    def kmers_from_bytes(self, seqs: torch.Tensor, n: int, k: int) -> torch.Tensor:
        out = self.empty(n, torch.int64)
        bad = C.c_uint64()
        st = self.lib.kmx_kmers_from_bytes(self._h, _ptr(seqs) if n else None, n, k, _ptr(out) if n else None, C.byref(bad))
        if st == _lib.E_INVALID_BASE:
            e = KmxError(st, f"invalid base at byte {bad.value}")
            e.first_bad = bad.value
            raise e
        self._ck(st)
        return out

    @_on_ctx_stream
    def revcomp_words(self, words: torch.Tensor, k: int) -> torch.Tensor:
        out = torch.empty_like(words)
        self._ck(self.lib.kmx_revcomp_words(self._h, _ptr(words), words.numel(), k, _ptr(out)))
        return out

    @_on_ctx_stream
    def canonical_words(self, words: torch.Tensor, k: int):
        canon = torch.empty_like(words)
        isc = self.empty(words.numel(), torch.uint8)
        self._ck(self.lib.kmx_canonical_words(self._h, _ptr(words), words.numel(), k, _ptr(canon), _ptr(isc)))
        return canon, isc

    @_on_ctx_stream
    def hash_words(self, words: torch.Tensor, hasher: int, hasher_k: int) -> torch.Tensor:
        out = torch.empty_like(words)
        self._ck(self.lib.kmx_hash_words(self._h, _ptr(words), words.numel(), hasher, hasher_k, _ptr(out)))
        return out

    @_on_ctx_stream
    def hash_words_sip13(self, words: torch.Tensor, key0: int = 0, key1: int = 0) -> torch.Tensor:
        """hash_one(&DefaultHasher / RandomState, kmer): SipHash-1-3 of each word (kmx_hash_words_sip13)"""
        out = torch.empty_like(words)
        self._ck(self.lib.kmx_hash_words_sip13(self._h, _ptr(words), words.numel(), key0 & (2**64 - 1), key1 & (2**64 - 1), _ptr(out)))
        return out

    @_on_ctx_stream
    def match_words(self, fw, rc, other) -> torch.Tensor:
        out = self.empty(fw.numel(), torch.uint8)
        self._ck(self.lib.kmx_match_words(self._h, _ptr(fw), _ptr(rc), _ptr(other), fw.numel(), _ptr(out)))
        return out

    @_on_ctx_stream
    def ck_shift(self, fw, rc, bases, k, append=True) -> torch.Tensor:
        dropped = self.empty(fw.numel(), torch.uint8)
        fn = self.lib.kmx_ck_append_bases if append else self.lib.kmx_ck_prepend_bases
        self._ck(fn(self._h, _ptr(fw), _ptr(rc), _ptr(bases), fw.numel(), k, _ptr(dropped)))
        return dropped

    @_on_ctx_stream
    def encode_kmers(self, seqs: torch.Tensor, n: int, seq_len: int, enc_byte: int, words_per_kmer: int) -> torch.Tensor:
        out = self.empty(n * words_per_kmer, torch.int64)
        self._ck(self.lib.kmx_encode_kmers(self._h, _ptr(seqs) if seqs.numel() else None, n, seq_len, enc_byte,
                                           words_per_kmer, _ptr(out)))
        return out

    @_on_ctx_stream
    def encode_windows(self, bases, n_reads, read_len, k, enc_byte, words_per_kmer) -> torch.Tensor:
        nwin = max(read_len - k + 1, 0)
        out = self.empty(n_reads * nwin * words_per_kmer, torch.int64)
        r = self._reads(bases, n_reads, read_len, None)
        self._ck(self.lib.kmx_encode_windows(self._h, C.byref(r), k, enc_byte, words_per_kmer, _ptr(out)))
        return out

    @_on_ctx_stream
    def encoding_rev_comp(self, words: torch.Tensor, K: int, enc_byte: int, words_per_kmer: int) -> torch.Tensor:
        out = torch.empty_like(words)
        self._ck(self.lib.kmx_encoding_rev_comp(self._h, _ptr(words), words.numel() // words_per_kmer, K, enc_byte,
                                                words_per_kmer, _ptr(out)))
        return out

    # ---- SeqVector (src/naive_impl/seq_vector.rs): 2-bit packed sequences on the device
    @_on_ctx_stream
    def seqvec_from_bytes(self, data: torch.Tensor, n: int | None = None) -> torch.Tensor:
        """SeqVector::from(&[u8]) (seq_vector.rs:230-242): ceil(n/32) u64 words (as int64 tensor), base i at bits [2i,2i+1]"""
        n = data.numel() if n is None else n
        words = torch.zeros((n + 31) // 32 + 2, dtype=torch.int64, device=self.device)[: (n + 31) // 32]
        return self.seqvec_push_chars(words, 0, data, n)

    @_on_ctx_stream
    def seqvec_push_chars(self, words: torch.Tensor, n_before: int, data: torch.Tensor, n: int | None = None) -> torch.Tensor:
        """SeqVector::push_chars (seq_vector.rs:141-161): append n ASCII bases after the n_before already stored"""
        n = data.numel() if n is None else n
        bad = C.c_uint64()
        st = self.lib.kmx_seqvec_push_chars(self._h, _ptr(words) if words.numel() else None, n_before, _ptr(data) if n else None, n, C.byref(bad))
        if st == _lib.E_INVALID_BASE:
            e = KmxError(st, f"invalid base at byte {bad.value}")
            e.first_bad = bad.value
            raise e
        self._ck(st)
        return words

    @_on_ctx_stream
    def seqvec_to_bytes(self, words: torch.Tensor, n_bases: int) -> torch.Tensor:
        out = self.empty(n_bases, torch.uint8)
        self._ck(self.lib.kmx_seqvec_to_bytes(self._h, _ptr(words) if n_bases else None, n_bases, _ptr(out) if n_bases else None))
        return out

    @_on_ctx_stream
    def seqvec_get_kmers(self, words: torch.Tensor, n_bases: int, pos: torch.Tensor, k: int) -> torch.Tensor:
        out = self.empty(pos.numel(), torch.int64)
        n = pos.numel()
        self._ck(self.lib.kmx_seqvec_get_kmers(self._h, _ptr(words) if n else None, n_bases, _ptr(pos) if n else None, n, k, _ptr(out) if n else None))
        return out

    @_on_ctx_stream
    def seqvec_iter_kmers(self, words: torch.Tensor, n_bases: int, k: int, start: int = 0, end: int | None = None) -> torch.Tensor:
        end = n_bases if end is None else end
        cnt = max(0, end - start - k + 1)
        out = self.empty(cnt, torch.int64)
        self._ck(self.lib.kmx_seqvec_iter_kmers(self._h, _ptr(words), n_bases, start, end, k, _ptr(out) if cnt else None))
        return out

    @_on_ctx_stream
    def seqvec_canonical_reduce(self, words: torch.Tensor, n_reads: int, read_len: int, k: int, hasher: int = 0, hasher_k: int = 0,
                                flags: int = 0, out: torch.Tensor | None = None, sync: bool = True):
        """canonical k-mer scan of the reads stored back to back in a SeqVector (read r = slice [r*L, (r+1)*L))"""
        out = self.empty(4, torch.int64) if out is None else out
        self._ck(self.lib.kmx_seqvec_canonical_reduce(self._h, _ptr(words) if n_reads else None, n_reads, read_len, k, hasher, hasher_k, flags, _ptr(out)))
        if not sync:
            return out
        v = u64_numpy(out)
        return Summary(int(v[0]), int(v[1]), int(v[2]), int(v[3]))

    # ---- minimizers
    @_on_ctx_stream
    def minimizer_words(self, words: torch.Tensor, k: int, width: int, hasher: int, hasher_k: int = 0):
        """Kmer::minimizer_word (kmer.rs:170-192) per k-mer word -> (mmer words int64, offsets int32)"""
        n = words.numel()
        mm, off = self.empty(n, torch.int64), self.empty(n, torch.int32)
        self._ck(self.lib.kmx_minimizer_words(self._h, _ptr(words) if n else None, n, k, width, hasher, hasher_k,
                                              _ptr(mm) if n else None, _ptr(off) if n else None))
        return mm, off

    @_on_ctx_stream
    def seqvec_minimizers(self, words: torch.Tensor, n_reads: int, read_len: int, k: int, w: int, hasher: int, hasher_k: int = 0):
        """SeqVecMinimizerIter over every read slice -> (word int64, pos int32), slot r*(L-k+1)+i"""
        tot = n_reads * max(read_len - k + 1, 0)
        mw, mp = self.empty(tot, torch.int64), self.empty(tot, torch.int32)
        self._ck(self.lib.kmx_seqvec_minimizers(self._h, _ptr(words) if n_reads else None, n_reads, read_len, k, w, hasher, hasher_k,
                                                _ptr(mw) if tot else None, _ptr(mp) if tot else None))
        return mw, mp

    @_on_ctx_stream
    def minimizers(self, bases, n_reads: int, read_len: int, k: int, w: int, hasher: int, hasher_k: int = 0, offsets=None, win_offsets=None,
                   check: bool = True):
        """kmx_minimizers: SeqVecMinimizerIter over every READ (ASCII; uniform, or ragged with offsets + win_offsets) ->
        (word int64, pos int32).  check: ask for the first read with an invalid byte (KmxError KMX_E_INVALID_BASE if there is one)"""
        if offsets is None:
            tot = n_reads * max(read_len - k + 1, 0)
        else:
            tot = int(win_offsets[-1].item()) if n_reads else 0
        mw, mp = self.empty(tot, torch.int64), self.empty(tot, torch.int32)
        r = self._reads(bases, n_reads, read_len, offsets)
        bad = C.c_uint64(0)
        self._ck(self.lib.kmx_minimizers(self._h, C.byref(r), _ptr(win_offsets) if win_offsets is not None else None, k, w, hasher, hasher_k,
                                         _ptr(mw) if tot else None, _ptr(mp) if tot else None, C.byref(bad) if check else None))
        return mw, mp

    @_on_ctx_stream
    def fastx_parse(self, text: torch.Tensor, fmt: int = 0, max_reads: int | None = None):
        """kmx_fastx_parse: FASTA/FASTQ file image (uint8, on the device) -> (bases uint8[n_bases], offsets int64[n_reads+1]).
        Two calls: the counts, then the emit into exactly sized buffers.  With `max_reads` (a bound on the number of records the
        caller vouches for): ONE call into buffers sized for the bound; KmxError (KMX_E_NOMEM) if the image holds more records."""
        n = int(text.numel())
        nr, nb = C.c_uint64(0), C.c_uint64(0)
        if max_reads is not None:
            bases = self.empty(max(n, 1), torch.uint8)
            offsets = self.empty(max_reads + 1, torch.int64)
            self._ck(self.lib.kmx_fastx_parse(self._h, _ptr(text) if n else None, n, fmt, _ptr(bases), _ptr(offsets), max_reads, C.byref(nr), C.byref(nb)))
            return bases[:nb.value], offsets[:nr.value + 1]
        self._ck(self.lib.kmx_fastx_parse(self._h, _ptr(text) if n else None, n, fmt, None, None, 0, C.byref(nr), C.byref(nb)))
        bases = self.empty(max(nb.value, 1), torch.uint8)
        offsets = self.empty(nr.value + 1, torch.int64)
        # (KMX_FASTX_SAME_TEXT: the emit reuses the chunk summaries of the counting call just made on the same image)
        self._ck(self.lib.kmx_fastx_parse(self._h, _ptr(text) if n else None, n, fmt | _lib.FASTX_SAME_TEXT, _ptr(bases), _ptr(offsets),
                                          nr.value, C.byref(nr), C.byref(nb)))
        return bases[:nb.value], offsets

    def fastx_reads(self, text: torch.Tensor, fmt: int = 0):
        """file image -> (bases, n_reads, read_len, offsets) the way the scan calls like it best: the uniform layout
        (offsets None, read_len = the length) when every read has the same length, else the ragged layout with the
        longest read as the length bound ("Real FASTQ / FASTA, end to end")"""
        bases, offsets = self.fastx_parse(text, fmt)
        n = int(offsets.numel()) - 1
        mn, mx = self.reads_length_range(offsets)
        if n > 0 and mn == mx:
            return bases, n, mx, None
        return bases, n, mx, offsets

    @_on_ctx_stream
    def reads_length_range(self, offsets: torch.Tensor) -> tuple[int, int]:
        """kmx_reads_length_range: (shortest, longest) read of a ragged batch (offsets: int64[n_reads+1] on the device)"""
        n = int(offsets.numel()) - 1
        mn, mx = C.c_uint32(0), C.c_uint32(0)
        self._ck(self.lib.kmx_reads_length_range(self._h, _ptr(offsets) if n > 0 else None, max(n, 0), C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    @_on_ctx_stream
    def encoding_decode(self, words: torch.Tensor, enc_byte: int, words_per_kmer: int) -> torch.Tensor:
        n = words.numel() // words_per_kmer
        out = self.empty(n * 32 * words_per_kmer, torch.uint8)
        self._ck(self.lib.kmx_encoding_decode(self._h, _ptr(words), n, enc_byte, words_per_kmer, _ptr(out)))
        return out

    # ---- decode / display direction
    @_on_ctx_stream
    def sub_kmer_words(self, words: torch.Tensor, k: int, pos: int, width: int) -> torch.Tensor:
        """Kmer::sub_kmer_word (kmer.rs:156-162) per word"""
        out = torch.empty_like(words)
        n = words.numel()
        self._ck(self.lib.kmx_sub_kmer_words(self._h, _ptr(words) if n else None, n, k, pos, width, _ptr(out) if n else None))
        return out

    @_on_ctx_stream
    def kmers_to_strings(self, words: torch.Tensor, k: int) -> torch.Tensor:
        """String::from(Kmer) (kmer.rs:196-207): uint8[n*k], lower case"""
        n = words.numel()
        out = self.empty(n * k, torch.uint8)
        self._ck(self.lib.kmx_kmers_to_strings(self._h, _ptr(words) if n else None, n, k, _ptr(out) if n * k else None))
        return out

    @_on_ctx_stream
    def bitmers_to_bytes(self, mers: torch.Tensor, length: int) -> torch.Tensor:
        """kmer::bitmer_to_bytes (src/kmer.rs:71-91): uint8[n*len], upper case"""
        n = mers.numel()
        out = self.empty(n * length, torch.uint8)
        self._ck(self.lib.kmx_bitmers_to_bytes(self._h, _ptr(mers) if n else None, n, length, _ptr(out) if n * length else None))
        return out

    # ---- Encoding<P, B> for any utils::Data word type (byte image of [P; B])
    @_on_ctx_stream
    def encode_kmers_p(self, seqs: torch.Tensor, n: int, seq_len: int, enc_byte: int, word_bits: int, words_per_kmer: int) -> torch.Tensor:
        out = self.empty(n * (word_bits // 8) * words_per_kmer, torch.uint8)
        self._ck(self.lib.kmx_encode_kmers_p(self._h, _ptr(seqs) if seqs.numel() else None, n, seq_len, enc_byte, word_bits,
                                             words_per_kmer, _ptr(out) if out.numel() else None))
        return out

    @_on_ctx_stream
    def encoding_rev_comp_p(self, arrays: torch.Tensor, K: int, enc_byte: int, word_bits: int, words_per_kmer: int) -> torch.Tensor:
        out = torch.empty_like(arrays)
        n = arrays.numel() // ((word_bits // 8) * words_per_kmer)
        self._ck(self.lib.kmx_encoding_rev_comp_p(self._h, _ptr(arrays) if n else None, n, K, enc_byte, word_bits, words_per_kmer,
                                                  _ptr(out) if n else None))
        return out

    @_on_ctx_stream
    def encoding_decode_p(self, arrays: torch.Tensor, enc_byte: int, word_bits: int, words_per_kmer: int) -> torch.Tensor:
        n = arrays.numel() // ((word_bits // 8) * words_per_kmer)
        out = self.empty(arrays.numel() * 4, torch.uint8)
        self._ck(self.lib.kmx_encoding_decode_p(self._h, _ptr(arrays) if n else None, n, enc_byte, word_bits, words_per_kmer,
                                                _ptr(out) if n else None))
        return out

    # ---- measurement helper
    @_on_ctx_stream
    def calib_stream_read(self, buf: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """kmx_calib_stream_read: read-only pass over `buf` with the scan's load shape (async; returns the 1-word xor fold)"""
        out = self.empty(1, torch.int64) if out is None else out
        self._ck(self.lib.kmx_calib_stream_read(self._h, _ptr(buf) if buf.numel() else None, buf.numel() * buf.element_size(), _ptr(out)))
        return out


class Comm:
    """kmx_comm: the RCCL communicator behind the C ABI (include/kmx.h, "multi-GPU exchange").  One per Context.

    `exchange_id(id_bytes_or_None) -> bytes` hands rank 0's unique id to the other ranks; with torch.distributed
    initialised the default does it with one broadcast."""

    def __init__(self, ctx: Context, n_ranks: int, rank: int, exchange_id=None):
        self.ctx = ctx
        lib = ctx.lib
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        # rank 0's status travels WITH the id (one extra byte): if kmx_comm_get_unique_id fails there -- no RCCL on the loader
        # path, say -- every rank learns it from the same broadcast and raises, instead of rank 0 raising alone while the
        # others wait in ncclCommInitRank for a peer that never comes.
        st0 = lib.kmx_comm_get_unique_id(buf) if rank == 0 else 0
        if n_ranks > 1:
            if exchange_id is None:
                exchange_id = self._torch_broadcast
            raw = exchange_id((bytes(buf) + bytes([st0 & 0xFF])) if rank == 0 else None)
            if len(raw) > _lib.COMM_ID_BYTES:
                st0 = raw[_lib.COMM_ID_BYTES]
                raw = raw[:_lib.COMM_ID_BYTES]
            buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(raw)
        if st0 != 0:
            raise KmxError(int(st0), "kmx_comm_get_unique_id failed on rank 0 (is librccl.so.1 on the loader path?)")
        h = C.c_void_p()
        ctx._ck(lib.kmx_comm_create(ctx._h, buf, n_ranks, rank, C.byref(h)))
        self._h = h
        self.n_ranks, self.rank = n_ranks, rank

    def _torch_broadcast(self, raw):
        import torch.distributed as dist

        t = torch.zeros(_lib.COMM_ID_BYTES + 1, dtype=torch.uint8)   # the id + rank 0's status byte
        if raw is not None:
            t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
        if dist.get_backend() == "nccl":
            t = t.to(self.ctx.device)
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    def close(self):
        if getattr(self, "_h", None):
            self.ctx.lib.kmx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self) -> int:
        return self.ctx.lib.kmx_comm_size(self._h)

    def histogram_allreduce(self, counts: torch.Tensor) -> torch.Tensor:
        """in-place sum over ranks of the 2^b u64 counters (ncclAllReduce, ncclUint64/ncclSum) on the context's stream"""
        counts.record_stream(self.ctx.stream)
        self.ctx._ck(self.ctx.lib.kmx_histogram_allreduce(self._h, _ptr(counts), counts.numel()))
        return counts

    def summary_allreduce(self, summary: torch.Tensor) -> torch.Tensor:
        """in-place combine of the device-resident 32-byte kmx_summary of every rank"""
        summary.record_stream(self.ctx.stream)
        self.ctx._ck(self.ctx.lib.kmx_summary_allreduce(self._h, _ptr(summary)))
        return summary
