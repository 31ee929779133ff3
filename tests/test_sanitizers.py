"""CPU sanitizer runs (GPU ASan is not available on the pool, so this is where the sanitizers go):
  * the oracle -- the plain-C restatement every parity test leans on -- rebuilt with -fsanitize=address,undefined
    (oracle/Makefile: libkmx_oracle_asan.so) and put through the reference's own known-answer tests and the FASTX
    cross-checks in a python that has libasan preloaded: an out-of-bounds read or a signed overflow in the checker
    would otherwise hide behind matching sums;
  * the C++ host layer (include/kmx.hpp) compiled with the same flags: without a GPU it must still fail loudly and
    cleanly (no leak of a half-built context, no UB on the error path)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    p = subprocess.run([gcc, f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip(f"{name} not installed")
    return os.path.realpath(p)


def _sanitizer_env():
    env = dict(os.environ)
    env["LD_PRELOAD"] = _runtime("libasan.so") + ":" + _runtime("libubsan.so")
    # python itself leaks by design; everything else is fatal
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=66"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1:exitcode=67"
    return env


def test_oracle_under_asan_ubsan_passes_the_reference_kats():
    env = _sanitizer_env()
    env["KMX_ORACLE_SANITIZE"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_fastx.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in r.stdout
    # and it really was the sanitized library
    assert os.path.exists(os.path.join(ROOT, "oracle", "libkmx_oracle_asan.so"))


def test_oracle_scans_under_asan_ubsan_on_ragged_and_dirty_input():
    """the entry points the GPU parity tests compare against, on the shapes that index hardest: ragged offsets with empty and
    sub-k reads, invalid bytes, two-word k, histogram, SeqVector, minimizers -- sized to run in seconds under ASan"""
    env = _sanitizer_env()
    env["KMX_ORACLE_SANITIZE"] = "1"
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from oracle import oracle as o
rng = np.random.default_rng(7)
alpha = np.frombuffer(b"ACGTacgtNn", np.uint8)
lens = rng.integers(0, 200, size=400); lens[::17] = 0; lens[3::29] = 5
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
host = alpha[rng.integers(0, 10, int(off[-1]))].copy()
for k in (1, 2, 13, 21, 31):
    s = o.canonical_reduce(host, len(lens), 0, k, hasher_k=k, offsets=off)
    w = o.canonical_windows(host, len(lens), 0, k, offsets=off)
    h = o.histogram(host, len(lens), 0, k, k, 12, offsets=off)
    assert int(h.sum()) == s.n_valid
for k in (33, 47, 63, 64):
    o.canonical_reduce2(host, len(lens), 0, k, with_hash=True, offsets=off)
    o.canonical_windows2(host, len(lens), 0, k, offsets=off)
u = alpha[rng.integers(0, 8, 150 * 300)].copy()
for k in (5, 31):
    o.canonical_reduce(u, 300, 150, k, hasher_k=k)
o.canonical_reduce2(u, 300, 150, 63)
sv = o.SeqVector(bytes(u[:900]).upper())
sv.push_chars(b"acgtACGT" * 7)
sv.canonical_reduce(6, 150, 31, 31)
[sv.get_kmer_u64(p, 31) for p in (0, 1, 33, len(sv) - 31)]
print("SCANS_OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0 and "SCANS_OK" in r.stdout, out[-4000:]


def test_cpp_host_layer_under_asan_ubsan_fails_cleanly_without_a_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here: the GPU run of this binary is tests/test_cpp_host_layer.py")
    lib = os.path.join(ROOT, "kmers_amd", "libkmx.so")
    if not os.path.exists(lib):
        pytest.skip("libkmx.so not built")
    _runtime("libasan.so")
    exe = str(tmp_path / "test_kmx_hpp_san")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_kmx_hpp.cpp"),
                           "-L", os.path.join(ROOT, "kmers_amd"), "-lkmx", "-L/opt/rocm/lib",
                           "-Wl,-rpath," + os.path.join(ROOT, "kmers_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=1:exitcode=66"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:exitcode=67"
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    out = r.stdout + r.stderr
    assert "runtime error:" not in out and "AddressSanitizer" not in out and "LeakSanitizer" not in out, out[-3000:]
    assert r.returncode not in (0, 66, 67), (r.returncode, out[-2000:])     # kmx::Context throws: no GPU, no fallback
