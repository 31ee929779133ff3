"""include/kmx.hpp (C++ mirror of the reference's Kmer / CanonicalKmer / Encoding surface) against the
reference's own test cases: compiled here with g++, executed on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_kmx_hpp.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_kmx_hpp")


def _build():
    lib = os.path.join(ROOT, "kmers_amd", "libkmx.so")
    assert os.path.exists(lib), "build libkmx.so first (python -m kmers_amd.build)"
    deps = [SRC, os.path.join(ROOT, "include", "kmx.hpp"), os.path.join(ROOT, "include", "kmx.h"), lib]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC,
                               "-L", os.path.join(ROOT, "kmers_amd"), "-lkmx", "-L/opt/rocm/lib",
                               "-Wl,-rpath," + os.path.join(ROOT, "kmers_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])
    return EXE


def test_cpp_host_layer_compiles_and_fails_loudly_without_gpu():
    exe = _build()
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0  # kmx::Context throws: there is no CPU fallback behind the header either
    assert "all C++ host-layer checks passed" not in r.stdout


@pytest.mark.gpu
def test_cpp_host_layer_reference_tests_on_gpu():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all C++ host-layer checks passed" in r.stdout
