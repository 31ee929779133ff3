"""In-tree build of libkmx.so (hipcc, gfx950 only).  `python -m kmers_amd.build [--force]`.

The .so is written next to this file (kmers_amd/libkmx.so): it is git-ignored but travels
to the GPU box with the repo snapshot, so nothing is compiled there.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libkmx.so")
SOURCES = ["kmx_bitslice.hip", "kmx_bitslice_k21.hip", "kmx_bitslice_k13_17.hip", "kmx_bitslice_k18_23.hip", "kmx_bitslice_k24_27.hip",
           "kmx_bitslice_k28_30.hip", "kmx_bitslice_k33_39.hip", "kmx_bitslice_k41_47.hip", "kmx_bitslice_k49_55.hip",
           "kmx_bitslice_k57_61.hip", "kmx_bitslice_ragged.hip", "kmx_scan.hip", "kmx_generic.hip", "kmx_elem.hip", "kmx_seqvec.hip", "kmx_fastx.hip", "kmx_api.hip"]
HEADERS = [os.path.join(CSRC, "kmx_device.h"), os.path.join(CSRC, "kmx_bitslice_kernel.h"), os.path.join(HERE, "..", "include", "kmx.h")]
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-fno-gpu-rdc", "-munsafe-fp-atomics"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libkmx is HIP-only (gfx950) and has no other build")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool, extra: list[str]) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + HEADERS):
        cmd = [hipcc(), *CXXFLAGS, *extra, "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, extra: list[str] | None = None) -> str:
    os.makedirs(OBJ, exist_ok=True)
    extra = extra or []
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, extra), SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs,
               "-Wl,-rpath,/opt/rocm/lib", "-Wl,-soname,libkmx.so"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
