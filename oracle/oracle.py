"""ctypes front-end of the CPU ORACLE (oracle/kmx_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from kmers_amd/ (the shipped HIP path has no
CPU fallback and fails loudly when its extension is missing).

The oracle restates the reference's naive_impl / encoding path in plain C; parity
is pinned by the reference's own known-answer tests (tests/golden/reference_kats.json).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

OK, E_INVALID_BASE, E_TOO_LONG, E_ARG = 0, 1, 2, 3
NO_MATCH, IDENTITY_MATCH, TWIN_MATCH = 0, 1, 2
INVALID_CODE = 0xFFFFFFFFFFFFFFFF


class OracleError(Exception):
    """Raised where the reference would panic (status carried in .status)."""

    def __init__(self, status: int, what: str = ""):
        super().__init__(f"oracle status {status} {what}")
        self.status = status


class Kmer(C.Structure):
    _fields_ = [("k", C.c_uint8), ("data", C.c_uint64)]


class CanonicalKmer(C.Structure):
    _fields_ = [("fw", Kmer), ("rc", Kmer)]


class Iter(C.Structure):
    _fields_ = [
        ("seq", C.c_void_p),
        ("seq_len", C.c_int32),
        ("km", CanonicalKmer),
        ("pos", C.c_int32),
        ("invalid", C.c_int),
        ("last_invalid", C.c_int32),
        ("k", C.c_int32),
    ]


class DQMer(C.Structure):   # seq_vector/minimizers.rs:8-13
    _fields_ = [("lmer", C.c_uint64), ("pos", C.c_size_t), ("hash", C.c_uint64)]


class MMIter(C.Structure):  # SeqVecMinimizerIter, seq_vector/minimizers.rs:39-46
    _fields_ = [("dq", DQMer * 96), ("head", C.c_size_t), ("len", C.c_size_t), ("k", C.c_size_t), ("w", C.c_size_t),
                ("curr_km_i", C.c_size_t), ("words", C.c_void_p), ("n_bases", C.c_size_t), ("start", C.c_size_t),
                ("slice_len", C.c_size_t), ("hasher_k", C.c_size_t)]

    def dq_hashes(self):
        return [self.dq[(self.head + i) % 96].hash for i in range(self.len)]


class Summary(C.Structure):
    _fields_ = [("n_valid", C.c_uint64), ("sum_canon", C.c_uint64), ("xor_hash", C.c_uint64), ("sum_fw", C.c_uint64)]


class Summary2(C.Structure):
    _fields_ = [
        ("n_valid", C.c_uint64),
        ("sum_lo", C.c_uint64),
        ("sum_hi", C.c_uint64),
        ("xor_lo", C.c_uint64),
        ("xor_hi", C.c_uint64),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/libkmx_oracle.so (portable flags) if missing; return its path.
    KMX_ORACLE_SANITIZE=1 (tests/test_sanitizers.py: a python started with libasan preloaded): the ASan + UBSan build."""
    if os.environ.get("KMX_ORACLE_SANITIZE") == "1":
        so = os.path.join(_HERE, "libkmx_oracle_asan.so")
        src = os.path.join(_HERE, "kmx_oracle.c")
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libkmx_oracle_asan.so"], stdout=subprocess.DEVNULL)
        return so
    so = os.path.join(_HERE, "libkmx_oracle.so")
    src = os.path.join(_HERE, "kmx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libkmx_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_native() -> str | None:
    """Host-tuned (-march=native) copy built on THIS machine into a temp dir; None if gcc is absent."""
    out = os.path.join(tempfile.gettempdir(), f"kmx_oracle_native_{os.getuid()}")
    os.makedirs(out, exist_ok=True)
    try:
        subprocess.check_call(["make", "-C", _HERE, "native", f"OUT={out}"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    except (subprocess.CalledProcessError, FileNotFoundError):
        return None
    return os.path.join(out, "libkmx_oracle_native.so")


def _bind(lib):
    u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    sig = {
        "kmo_encode_binary_u8": (C.c_uint64, [C.c_uint8]),
        "kmo_encode_binary": (C.c_int, [C.c_uint8, u64p]),
        "kmo_encode_complement_binary_u8": (C.c_uint64, [C.c_uint8]),
        "kmo_complement_base": (C.c_uint64, [C.c_uint64]),
        "kmo_is_valid_nuc": (C.c_int, [C.c_uint64]),
        "kmo_mask_table": (C.c_uint64, [C.c_uint]),
        "kmo_kmer_from_u64": (Kmer, [C.c_uint64, C.c_uint8]),
        "kmo_kmer_from_bytes": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(Kmer)]),
        "kmo_kmer_append_base": (C.c_uint64, [C.POINTER(Kmer), C.c_uint64]),
        "kmo_kmer_prepend_base": (C.c_uint64, [C.POINTER(Kmer), C.c_uint64]),
        "kmo_kmer_append_base_u8": (C.c_uint64, [C.POINTER(Kmer), C.c_uint8]),
        "kmo_kmer_prepend_base_u8": (C.c_uint64, [C.POINTER(Kmer), C.c_uint8]),
        "kmo_revcomp_word": (C.c_uint64, [C.c_uint64, C.c_uint8]),
        "kmo_kmer_to_reverse_complement": (Kmer, [Kmer]),
        "kmo_kmer_cmp": (C.c_int, [Kmer, Kmer]),
        "kmo_kmer_eq": (C.c_int, [Kmer, Kmer]),
        "kmo_kmer_is_canonical": (C.c_int, [Kmer]),
        "kmo_kmer_to_canonical": (Kmer, [Kmer]),
        "kmo_sub_kmer_word": (C.c_int, [C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t, u64p]),
        "kmo_kmer_to_string": (C.c_size_t, [Kmer, C.c_char_p]),
        "kmo_ck_blank_of_size": (CanonicalKmer, [C.c_uint8]),
        "kmo_ck_from_u64": (CanonicalKmer, [C.c_uint64, C.c_uint8]),
        "kmo_ck_from_kmer": (CanonicalKmer, [Kmer]),
        "kmo_ck_from_bytes": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(CanonicalKmer)]),
        "kmo_ck_swap": (None, [C.POINTER(CanonicalKmer)]),
        "kmo_ck_is_fw_canonical": (C.c_int, [C.POINTER(CanonicalKmer)]),
        "kmo_ck_append_base": (C.c_uint64, [C.POINTER(CanonicalKmer), C.c_uint64]),
        "kmo_ck_prepend_base": (C.c_uint64, [C.POINTER(CanonicalKmer), C.c_uint64]),
        "kmo_ck_append_base_u8": (C.c_uint64, [C.POINTER(CanonicalKmer), C.c_uint8]),
        "kmo_ck_prepend_base_u8": (C.c_uint64, [C.POINTER(CanonicalKmer), C.c_uint8]),
        "kmo_ck_get_canonical_word": (C.c_uint64, [C.POINTER(CanonicalKmer)]),
        "kmo_ck_get_word_equivalency": (C.c_int, [C.POINTER(CanonicalKmer), C.c_uint64]),
        "kmo_iter_from_u8_slice": (None, [C.POINTER(Iter), C.c_char_p, C.c_size_t, C.c_uint8]),
        "kmo_iter_exhausted": (C.c_int, [C.POINTER(Iter)]),
        "kmo_iter_inc": (C.c_int, [C.POINTER(Iter)]),
        "kmo_iter_inc_by": (C.c_int, [C.POINTER(Iter), C.c_size_t]),
        "kmo_lex_hash_u64": (C.c_uint64, [C.c_uint64, C.c_size_t]),
        "kmo_siphash": (C.c_uint64, [C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]),
        "kmo_siphash13_u64": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64]),
        "kmo_nuc2internal": (C.c_uint8, [C.c_uint8]),
        "kmo_rev_encoding": (C.c_uint8, [C.c_uint8]),
        "kmo_naive_nuc2bits": (C.c_uint8, [C.c_uint8, C.c_uint8]),
        "kmo_naive_bits2nuc": (C.c_uint8, [C.c_uint8, C.c_uint8]),
        "kmo_naive_complement": (C.c_uint8, [C.c_uint8, C.c_uint8]),
        "kmo_naive_encode": (C.c_int, [C.c_uint8, C.c_char_p, C.c_size_t, u8p, C.c_size_t]),
        "kmo_naive_decode": (None, [C.c_uint8, u8p, C.c_size_t, u8p]),
        "kmo_naive_rev_comp": (C.c_int, [C.c_uint8, C.c_size_t, u8p, C.c_size_t]),
        "kmo_xor10_encode": (C.c_int, [C.c_char_p, C.c_size_t, u8p, C.c_size_t]),
        "kmo_xor10_decode": (None, [u8p, C.c_size_t, u8p]),
        "kmo_xor10_rev_comp": (C.c_int, [C.c_size_t, u8p, C.c_size_t]),
        "kmo_xor10_rev_comp_b1_quirk": (C.c_uint64, [C.c_uint64, C.c_uint]),
        "kmo_word_for_k": (C.c_size_t, [C.c_size_t, C.c_size_t]),
        "kmo_generic_get": (C.c_uint64, [u8p, C.c_size_t, C.c_size_t]),
        "kmo_generic_get_prefix": (C.c_int, [u8p, C.c_size_t, C.c_uint, C.c_size_t, u64p]),
        "kmo_bitmer_to_bytes": (None, [C.c_uint64, C.c_size_t, u8p]),
        "kmo_canonical_reduce": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint8, C.c_size_t,
                                           C.POINTER(Summary)]),
        "kmo_canonical_windows": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint8,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "kmo_compute_naive": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, u64p]),
        "kmo_compute_naive_canonical": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, u64p]),
        "kmo_canonical_reduce2": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint8, C.c_int,
                                            C.POINTER(Summary2)]),
        "kmo_canonical_windows2": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint8,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "kmo_seqvec_push_chars": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
        "kmo_seqvec_get_kmer_u64": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, u64p]),
        "kmo_seqvec_to_bytes": (None, [C.c_void_p, C.c_size_t, C.c_void_p]),
        "kmo_seqvec_iter_kmers": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
        "kmo_seqvec_canonical_reduce": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint8, C.c_size_t, C.POINTER(Summary)]),
        "kmo_mm_hash": (C.c_uint64, [C.c_uint64, C.c_size_t]),
        "kmo_minimizer_word": (C.c_int, [C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t, u64p, C.POINTER(C.c_size_t)]),
        "kmo_mmiter_enqueue": (None, [C.POINTER(MMIter), DQMer]),
        "kmo_mmiter_new": (C.c_int, [C.POINTER(MMIter), C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
        "kmo_mmiter_next": (C.c_int, [C.POINTER(MMIter), u64p, C.POINTER(C.c_size_t)]),
        "kmo_seqvec_minimizers": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
        "kmo_fastx_parse": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t)]),
        "kmo_splitmix64": (C.c_uint64, [C.c_uint64]),
        "kmo_gen_reads": (None, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]),
        "kmo_bucket_of": (C.c_uint64, [C.c_uint64, C.c_uint]),
        "kmo_histogram": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint8, C.c_size_t, C.c_uint,
                                    C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_LIB = None


def lib(native: bool = False):
    """Load (building if needed) the oracle library."""
    global _LIB
    if native:
        p = build_native()
        if p is not None:
            return _bind(C.CDLL(p))
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
    return _LIB


# ------------------------------------------------------------------ helpers --

def _u8(a) -> np.ndarray:
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def kmer_from_bytes(s: bytes) -> Kmer:
    km = Kmer()
    st = lib().kmo_kmer_from_bytes(s, len(s), C.byref(km))
    if st != OK:
        raise OracleError(st, "Kmer::from")
    return km


def kmer_to_string(km: Kmer) -> str:
    buf = C.create_string_buffer(40)
    lib().kmo_kmer_to_string(km, buf)
    return buf.value.decode()


def ck_from_bytes(s: bytes) -> CanonicalKmer:
    ck = CanonicalKmer()
    st = lib().kmo_ck_from_bytes(s, len(s), C.byref(ck))
    if st != OK:
        raise OracleError(st, "CanonicalKmer::from")
    return ck


def canonical_reduce(reads, n_reads, read_len, k, hasher_k=0, offsets=None, native_lib=None) -> Summary:
    reads = _u8(reads)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
    out = Summary()
    L = native_lib or lib()
    st = L.kmo_canonical_reduce(_ptr(reads), n_reads, read_len, _ptr(off), k, hasher_k, C.byref(out))
    if st != OK:
        raise OracleError(st, "canonical_reduce")
    return out


def win_offsets_for(n_reads, read_len, k, offsets=None) -> np.ndarray:
    """Exclusive prefix sum of per-read window counts (n_reads+1 entries)."""
    if offsets is None:
        w = max(read_len - k + 1, 0)
        return (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(w)).astype(np.uint64)
    lens = np.diff(np.asarray(offsets, dtype=np.int64))
    w = np.maximum(lens - k + 1, 0)
    return np.concatenate([[0], np.cumsum(w)]).astype(np.uint64)


def canonical_windows(reads, n_reads, read_len, k, offsets=None):
    """-> (fw, rc, canon, flags) dense arrays, one slot per (read, pos)."""
    reads = _u8(reads)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
    wo = win_offsets_for(n_reads, read_len, k, off)
    total = int(wo[-1])
    fw = np.zeros(total, np.uint64)
    rc = np.zeros(total, np.uint64)
    canon = np.zeros(total, np.uint64)
    flags = np.zeros(total, np.uint8)
    st = lib().kmo_canonical_windows(_ptr(reads), n_reads, read_len, _ptr(off), _ptr(wo), k, _ptr(fw), _ptr(rc),
                                     _ptr(canon), _ptr(flags))
    if st != OK:
        raise OracleError(st, "canonical_windows")
    return fw, rc, canon, flags


def canonical_reduce2(reads, n_reads, read_len, k, with_hash=False, offsets=None) -> Summary2:
    reads = _u8(reads)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
    out = Summary2()
    st = lib().kmo_canonical_reduce2(_ptr(reads), n_reads, read_len, _ptr(off), k, int(with_hash), C.byref(out))
    if st != OK:
        raise OracleError(st, "canonical_reduce2")
    return out


def canonical_windows2(reads, n_reads, read_len, k, offsets=None):
    reads = _u8(reads)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
    wo = win_offsets_for(n_reads, read_len, k, off)
    total = int(wo[-1])
    fw = np.zeros(2 * total, np.uint64)
    rc = np.zeros(2 * total, np.uint64)
    canon = np.zeros(2 * total, np.uint64)
    flags = np.zeros(total, np.uint8)
    st = lib().kmo_canonical_windows2(_ptr(reads), n_reads, read_len, _ptr(off), _ptr(wo), k, _ptr(fw), _ptr(rc),
                                      _ptr(canon), _ptr(flags))
    if st != OK:
        raise OracleError(st, "canonical_windows2")
    return fw.reshape(-1, 2), rc.reshape(-1, 2), canon.reshape(-1, 2), flags


class SeqVector:
    """host mirror of seq_vector.rs::SeqVector on top of the C restatement (words: np.uint64 array, n: bases)"""

    def __init__(self, data: bytes = b"", capacity: int = 0):
        self.n = 0
        self.words = np.zeros(max((max(capacity, len(data)) + 31) // 32, 1) + 1, dtype=np.uint64)
        if data:
            self.push_chars(data)

    def push_chars(self, data: bytes) -> None:
        need = (self.n + len(data) + 31) // 32 + 1
        if need > self.words.size:
            self.words = np.concatenate([self.words, np.zeros(need - self.words.size, dtype=np.uint64)])
        buf = _u8(data)
        bad = C.c_size_t(0)
        st = lib().kmo_seqvec_push_chars(_ptr(self.words), self.n, _ptr(buf), buf.size, C.byref(bad))
        if st != 0:
            e = OracleError(st, f"invalid base at byte {bad.value}")
            e.first_bad = bad.value
            raise e
        self.n += buf.size

    def __len__(self) -> int:
        return self.n

    def get_kmer_u64(self, pos: int, k: int) -> int:
        out = C.c_uint64()
        st = lib().kmo_seqvec_get_kmer_u64(_ptr(self.words), self.n, pos, k, C.byref(out))
        if st != 0:
            raise OracleError(st, "get_kmer_u64 outside the vector")
        return out.value

    def to_bytes(self) -> bytes:
        out = np.zeros(self.n, dtype=np.uint8)
        lib().kmo_seqvec_to_bytes(_ptr(self.words), self.n, _ptr(out))
        return out.tobytes()

    def iter_kmers(self, k: int, start: int = 0, end: int | None = None) -> np.ndarray:
        end = self.n if end is None else end
        out = np.zeros(max(end - start - k + 1, 0), dtype=np.uint64)
        cnt = lib().kmo_seqvec_iter_kmers(_ptr(self.words), self.n, start, end, k, _ptr(out))
        return out[:cnt]

    def canonical_reduce(self, n_reads: int, read_len: int, k: int, hasher_k: int = 0) -> "Summary":
        s = Summary()
        st = lib().kmo_seqvec_canonical_reduce(_ptr(self.words), n_reads, read_len, k, hasher_k, C.byref(s))
        if st != 0:
            raise OracleError(st, "seqvec_canonical_reduce")
        return s


def minimizer_word(word: int, k: int, width: int, hasher_k: int = 0) -> tuple[int, int]:
    """Kmer::minimizer_word (kmer.rs:170-192); hasher_k == 0: identity hasher, else LexHasher(hasher_k)"""
    mm, off = C.c_uint64(), C.c_size_t()
    st = lib().kmo_minimizer_word(word, k, width, hasher_k, C.byref(mm), C.byref(off))
    if st != 0:
        raise OracleError(st, "minimizer_word")
    return mm.value, off.value


def seqvec_iter_minimizers(sv: "SeqVector", k: int, w: int, hasher_k: int = 0, start: int = 0, end: int | None = None):
    """SeqVecMinimizerIter over sv.slice(start, end): list of (word, pos)"""
    end = sv.n if end is None else end
    it = MMIter()
    st = lib().kmo_mmiter_new(C.byref(it), _ptr(sv.words), sv.n, start, end, k, w, hasher_k)
    if st != 0:
        raise OracleError(st, "SeqVecMinimizerIter::new")
    out, word, pos = [], C.c_uint64(), C.c_size_t()
    while lib().kmo_mmiter_next(C.byref(it), C.byref(word), C.byref(pos)):
        out.append((word.value, pos.value))
    return out


def seqvec_minimizers(sv: "SeqVector", n_reads: int, read_len: int, k: int, w: int, hasher_k: int = 0):
    W = max(read_len - k + 1, 0)
    words = np.zeros(n_reads * W, dtype=np.uint64)
    pos = np.zeros(n_reads * W, dtype=np.uint32)
    st = lib().kmo_seqvec_minimizers(_ptr(sv.words), n_reads, read_len, k, w, hasher_k, _ptr(words), _ptr(pos))
    if st != 0:
        raise OracleError(st, "seqvec_minimizers")
    return words, pos


def fastx_parse(text: bytes, fmt: int = 0):
    """FASTA/FASTQ record splitting (BUILD-DEFINED, the reference has no parser): returns
    (bases u8[n_bases], offsets u64[n_reads+1]), read r = bases[offsets[r]:offsets[r+1]].  fmt 1 = FASTQ (line i is a
    read iff i % 4 == 1), 2 = FASTA ('>' lines start a record; the other lines up to the next '>' are its sequence),
    0 = by the first byte.  Lines end at '\\n', every '\\r' on a sequence line is dropped, the last line may lack its '\\n'.
    Written with bytes.split -- a different algorithm from kmo_fastx_parse, which it is tested against."""
    text = bytes(text)
    if not text:
        return np.zeros(0, np.uint8), np.zeros(1, np.uint64)
    if fmt == 0:
        fmt = 1 if text[:1] == b"@" else 2 if text[:1] == b">" else 3
    if fmt not in (1, 2) or text[:1] != (b"@" if fmt == 1 else b">"):
        raise ValueError("not a FASTA/FASTQ text")
    lines = text.split(b"\n")
    if text.endswith(b"\n"):
        lines.pop()          # what follows the last newline is not a line
    reads = []
    if fmt == 1:
        reads = [ln.replace(b"\r", b"") for ln in lines[1::4]]
    else:
        for ln in lines:
            if ln[:1] == b">":
                reads.append([])
            else:
                reads[-1].append(ln.replace(b"\r", b""))
        reads = [b"".join(r) for r in reads]
    offsets = np.zeros(len(reads) + 1, np.uint64)
    if reads:
        offsets[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    return np.frombuffer(b"".join(reads), dtype=np.uint8).copy(), offsets


def fastx_parse_c(text: bytes, fmt: int = 0):
    """kmo_fastx_parse (the C state machine): same result as fastx_parse"""
    text = bytes(text)
    buf = np.frombuffer(text, dtype=np.uint8)
    nr, nb = C.c_size_t(0), C.c_size_t(0)
    p = buf.ctypes.data if len(buf) else None
    if lib().kmo_fastx_parse(p, len(buf), fmt, None, None, C.byref(nr), C.byref(nb)) != 0:
        raise ValueError("not a FASTA/FASTQ text")
    bases = np.zeros(max(nb.value, 1), np.uint8)
    offsets = np.zeros(nr.value + 1, np.uint64)
    lib().kmo_fastx_parse(p, len(buf), fmt, bases.ctypes.data, offsets.ctypes.data, C.byref(nr), C.byref(nb))
    return bases[:nb.value], offsets


def gen_reads(seed: int, first_byte: int, nbytes: int) -> np.ndarray:
    out = np.empty(nbytes, np.uint8)
    lib().kmo_gen_reads(seed, first_byte, _ptr(out), nbytes)
    return out


def histogram(reads, n_reads, read_len, k, hasher_k, log2_buckets, offsets=None) -> np.ndarray:
    reads = _u8(reads)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
    counts = np.zeros(1 << log2_buckets, np.uint64)
    st = lib().kmo_histogram(_ptr(reads), n_reads, read_len, _ptr(off), k, hasher_k, log2_buckets, _ptr(counts))
    if st != OK:
        raise OracleError(st, "histogram")
    return counts


def compute_naive(b, K: int) -> int:
    b = _u8(b)
    out = C.c_uint64()
    st = lib().kmo_compute_naive(_ptr(b), b.size, K, C.byref(out))
    if st != OK:
        raise OracleError(st, "compute_naive")
    return out.value


def naive_encode(enc: int, seq: bytes, nbytes: int) -> np.ndarray:
    arr = np.zeros(nbytes, np.uint8)
    st = lib().kmo_naive_encode(enc, seq, len(seq), arr.ctypes.data_as(C.POINTER(C.c_uint8)), nbytes)
    if st != OK:
        raise OracleError(st, "Naive::encode")
    return arr


def naive_decode(enc: int, arr: np.ndarray) -> bytes:
    arr = _u8(arr)
    out = np.zeros(arr.size * 4, np.uint8)
    lib().kmo_naive_decode(enc, arr.ctypes.data_as(C.POINTER(C.c_uint8)), arr.size,
                           out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.tobytes()


def naive_rev_comp(enc: int, K: int, arr: np.ndarray) -> np.ndarray:
    arr = _u8(arr).copy()
    st = lib().kmo_naive_rev_comp(enc, K, arr.ctypes.data_as(C.POINTER(C.c_uint8)), arr.size)
    if st != OK:
        raise OracleError(st, "Naive::rev_comp")
    return arr


def xor10_encode(seq: bytes, nbytes: int) -> np.ndarray:
    arr = np.zeros(nbytes, np.uint8)
    st = lib().kmo_xor10_encode(seq, len(seq), arr.ctypes.data_as(C.POINTER(C.c_uint8)), nbytes)
    if st != OK:
        raise OracleError(st, "Xor10::encode")
    return arr


def xor10_rev_comp(K: int, arr: np.ndarray) -> np.ndarray:
    arr = _u8(arr).copy()
    st = lib().kmo_xor10_rev_comp(K, arr.ctypes.data_as(C.POINTER(C.c_uint8)), arr.size)
    if st != OK:
        raise OracleError(st, "Xor10::rev_comp")
    return arr


def xor10_decode(arr: np.ndarray) -> bytes:
    arr = _u8(arr)
    out = np.zeros(arr.size * 4, np.uint8)
    lib().kmo_xor10_decode(arr.ctypes.data_as(C.POINTER(C.c_uint8)), arr.size,
                           out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.tobytes()


def words(arr: np.ndarray, p_bits: int) -> list[int]:
    """View the flat little-endian bit array as a list of P-bit words (python ints; handles u128)."""
    b = _u8(arr).tobytes()
    n = p_bits // 8
    return [int.from_bytes(b[i:i + n], "little") for i in range(0, len(b), n)]


def sub_kmer_word(word: int, k: int, pos: int, width: int) -> int:
    """Kmer::sub_kmer_word (kmer.rs:156-162); the reference's asserts surface as OracleError"""
    out = C.c_uint64()
    st = lib().kmo_sub_kmer_word(word, k, pos, width, C.byref(out))
    if st != OK:
        raise OracleError(st, "sub_kmer_word")
    return out.value


def bitmer_to_bytes(mer: int, length: int) -> bytes:
    """kmer::bitmer_to_bytes (src/kmer.rs:71-91)"""
    out = np.zeros(max(length, 1), np.uint8)
    lib().kmo_bitmer_to_bytes(mer, length, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out[:length].tobytes()
