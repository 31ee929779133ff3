"""ctypes binding of the libkmx C ABI (include/kmx.h).

There is deliberately NO fallback: if libkmx.so is missing, or no HIP device is visible when
a context is created, this raises.  The CPU oracle under oracle/ is test infrastructure and
is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The one library this package loads.  Nothing in the environment changes it: development builds (tools/dev_variant.py) are
# selected by the tool or test that wants one (tools/devlib.py, `pytest --kmx-lib`), in code, before load().
LIB_PATH = os.path.join(_HERE, "libkmx.so")
COMM_ID_BYTES = 128

# status codes (include/kmx.h)
OK, E_ARG, E_K_RANGE, E_HIP, E_INVALID_BASE, E_TOO_LONG, E_NOMEM = range(7)
HASH_NONE, HASH_LEX, HASH_IDENTITY = 0, 1, 2
NO_MATCH, IDENTITY_MATCH, TWIN_MATCH = 0, 1, 2
WIN_VALID, WIN_FW_CANONICAL = 1, 2
REDUCE_SUM_FW = 1


class KmxError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"kmx status {status}: {msg}")
        self.status = status


class Reads(C.Structure):
    _fields_ = [("d_bases", C.c_void_p), ("n_reads", C.c_uint64), ("read_len", C.c_uint32), ("d_offsets", C.c_void_p)]


class Summary(C.Structure):
    _fields_ = [("n_valid", C.c_uint64), ("sum_canon", C.c_uint64), ("xor_hash", C.c_uint64), ("sum_fw", C.c_uint64)]


class Summary2(C.Structure):
    _fields_ = [("n_valid", C.c_uint64), ("sum_lo", C.c_uint64), ("sum_hi", C.c_uint64), ("xor_lo", C.c_uint64),
                ("xor_hi", C.c_uint64)]


_vp, _u64, _u32, _u8, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_int
_RP = C.POINTER(Reads)

# every symbol include/kmx.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "kmx_version": (_int, []),
    "kmx_strerror": (C.c_char_p, [_int]),
    "kmx_ctx_create": (_int, [_int, C.POINTER(_vp)]),
    "kmx_ctx_create_on_stream": (_int, [_int, _vp, C.POINTER(_vp)]),
    "kmx_ctx_destroy": (None, [_vp]),
    "kmx_ctx_synchronize": (_int, [_vp]),
    "kmx_ctx_device": (_int, [_vp]),
    "kmx_ctx_set_work_buffer_limit": (_int, [_vp, C.c_size_t]),
    "kmx_ctx_work_buffer_info": (_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "kmx_last_error": (C.c_char_p, [_vp]),
    "kmx_malloc": (_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "kmx_free": (_int, [_vp, _vp]),
    "kmx_memcpy_h2d": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "kmx_memcpy_d2h": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "kmx_memset": (_int, [_vp, _vp, _int, C.c_size_t]),
    "kmx_canonical_reduce": (_int, [_vp, _RP, _u32, _u32, _u32, _u32, _vp]),
    "kmx_canonical_reduce_host": (_int, [_vp, _RP, _u32, _u32, _u32, _u32, _vp]),
    "kmx_canonical_windows": (_int, [_vp, _RP, _vp, _u32, _vp, _vp, _vp, _vp]),
    "kmx_canonical_reduce2": (_int, [_vp, _RP, _u32, _u32, _vp]),
    "kmx_canonical_windows2": (_int, [_vp, _RP, _vp, _u32, _vp, _vp, _vp, _vp]),
    "kmx_histogram": (_int, [_vp, _RP, _u32, _u32, _u32, _u32, _vp]),
    "kmx_gen_reads": (_int, [_vp, _u64, _u64, _vp, _u64]),
    "kmx_kmers_from_bytes": (_int, [_vp, _vp, _u64, _u32, _vp, C.POINTER(_u64)]),
    "kmx_revcomp_words": (_int, [_vp, _vp, _u64, _u32, _vp]),
    "kmx_canonical_words": (_int, [_vp, _vp, _u64, _u32, _vp, _vp]),
    "kmx_hash_words": (_int, [_vp, _vp, _u64, _u32, _u32, _vp]),
    "kmx_hash_words_sip13": (_int, [_vp, _vp, _u64, _u64, _u64, _vp]),
    "kmx_match_words": (_int, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "kmx_ck_append_bases": (_int, [_vp, _vp, _vp, _vp, _u64, _u32, _vp]),
    "kmx_ck_prepend_bases": (_int, [_vp, _vp, _vp, _vp, _u64, _u32, _vp]),
    "kmx_encode_kmers": (_int, [_vp, _vp, _u64, _u32, _u8, _u32, _vp]),
    "kmx_encode_windows": (_int, [_vp, _RP, _u32, _u8, _u32, _vp]),
    "kmx_encoding_rev_comp": (_int, [_vp, _vp, _u64, _u32, _u8, _u32, _vp]),
    "kmx_encoding_decode": (_int, [_vp, _vp, _u64, _u8, _u32, _vp]),
    "kmx_seqvec_push_chars": (_int, [_vp, _vp, _u64, _vp, _u64, C.POINTER(_u64)]),
    "kmx_seqvec_to_bytes": (_int, [_vp, _vp, _u64, _vp]),
    "kmx_seqvec_get_kmers": (_int, [_vp, _vp, _u64, _vp, _u64, _u32, _vp]),
    "kmx_seqvec_iter_kmers": (_int, [_vp, _vp, _u64, _u64, _u64, _u32, _vp]),
    "kmx_seqvec_canonical_reduce": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u32, _u32, _vp]),
    "kmx_minimizer_words": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u32, _vp, _vp]),
    "kmx_seqvec_minimizers": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "kmx_minimizers": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "kmx_fastx_parse": (_int, [_vp, _vp, _u64, _u32, _vp, _vp, _u64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kmx_sub_kmer_words": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "kmx_kmers_to_strings": (_int, [_vp, _vp, _u64, _u32, _vp]),
    "kmx_bitmers_to_bytes": (_int, [_vp, _vp, _u64, _u32, _vp]),
    "kmx_encode_kmers_p": (_int, [_vp, _vp, _u64, _u32, _u8, _u32, _u32, _vp]),
    "kmx_encoding_rev_comp_p": (_int, [_vp, _vp, _u64, _u32, _u8, _u32, _u32, _vp]),
    "kmx_encoding_decode_p": (_int, [_vp, _vp, _u64, _u8, _u32, _u32, _vp]),
    "kmx_comm_get_unique_id": (_int, [_vp]),
    "kmx_comm_create": (_int, [_vp, _vp, _int, _int, C.POINTER(_vp)]),
    "kmx_comm_destroy": (None, [_vp]),
    "kmx_comm_size": (_int, [_vp]),
    "kmx_comm_rank": (_int, [_vp]),
    "kmx_histogram_allreduce": (_int, [_vp, _vp, _u64]),
    "kmx_summary_allreduce": (_int, [_vp, _vp]),
    "kmx_calib_stream_read": (_int, [_vp, _vp, _u64, _vp]),
    "kmx_reads_length_range": (_int, [_vp, _vp, _u64, _vp, _vp]),
}

FASTX_AUTO, FASTX_FASTQ, FASTX_FASTA = 0, 1, 2
FASTX_SAME_TEXT = 0x100   # or-ed into the format of the emit call that follows the counting call on the same image

_LIB = None


def load(import_torch_first: bool = True):
    """dlopen kmers_amd/libkmx.so.  torch (if installed) is imported first so that libkmx's
    libamdhip64.so.7 dependency resolves to the HIP runtime torch already loaded -- two HIP
    runtimes in one process would not share device allocations."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m kmers_amd.build` (hipcc, gfx950). "
            "kmers_amd has no CPU fallback.")
    if import_torch_first:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI mismatch with include/kmx.h
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(lib, ctx, status: int):
    if status == OK:
        return
    msg = lib.kmx_strerror(status).decode()
    if status == E_HIP and ctx:
        msg += " -- " + lib.kmx_last_error(ctx).decode()
    raise KmxError(status, msg)
