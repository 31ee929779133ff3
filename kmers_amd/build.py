"""In-tree build of libkmx.so (hipcc, gfx950 only).  `python -m kmers_amd.build [--force]`.

The .so is written next to this file (kmers_amd/libkmx.so): it is git-ignored but travels
to the GPU box with the repo snapshot, so nothing is compiled there.

Development builds (patched copies of the sources, extra -D switches) are made OUTSIDE this package by tools/dev_variant.py;
nothing here, and nothing in kmers_amd/_lib.py, knows about them.  The build records its flags in a stamp file, so objects left
behind by different flags are rebuilt instead of being taken for current.
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
# (the slowest translation units first: the pool starts them in this order, and the build is as long as its longest tail)
SOURCES = ["kmx_sweep.hip", "kmx_hist.hip", "kmx_hist32.hip", "kmx_bitslice.hip", "kmx_scan.hip", "kmx_bitslice_k21.hip", "kmx_bitslice_k9_12.hip", "kmx_bitslice_k13_17.hip", "kmx_bitslice_k18_23.hip",
           "kmx_bitslice_k24_27.hip", "kmx_bitslice_k28_30.hip", "kmx_bitslice_k33_39.hip", "kmx_bitslice_k41_47.hip", "kmx_bitslice_k49_55.hip",
           "kmx_bitslice_k57_61.hip", "kmx_bitslice_k34_40.hip", "kmx_bitslice_k42_48.hip", "kmx_bitslice_k50_56.hip", "kmx_bitslice_k58_64.hip",
           "kmx_bitslice_ragged_k9_12.hip", "kmx_bitslice_ragged_k13_16.hip", "kmx_bitslice_ragged_k17_20.hip", "kmx_bitslice_ragged_k21_24.hip", "kmx_bitslice_ragged_k25_28.hip",
           "kmx_bitslice_ragged_k29_31.hip", "kmx_bitslice_ragged2_k33_36.hip", "kmx_bitslice_ragged2_k37_40.hip", "kmx_bitslice_ragged2_k41_44.hip", "kmx_bitslice_ragged2_k45_48.hip", "kmx_bitslice_ragged2_k49_52.hip", "kmx_bitslice_ragged2_k53_56.hip", "kmx_bitslice_ragged2_k57_60.hip", "kmx_bitslice_ragged2_k61_64.hip", "kmx_generic.hip", "kmx_segments.hip", "kmx_elem.hip", "kmx_seqvec.hip", "kmx_minimizers.hip", "kmx_fastx.hip", "kmx_comm.hip", "kmx_api.hip"]
HEADERS = [os.path.join(CSRC, "kmx_device.h"), os.path.join(CSRC, "kmx_hist_part.h"), os.path.join(CSRC, "kmx_bitslice_kernel.h"), os.path.join(CSRC, "kmx_internal.h"), os.path.join(CSRC, "kmx_scan_kernel.h"),
           os.path.join(HERE, "..", "include", "kmx.h")]
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-fno-gpu-rdc", "-munsafe-fp-atomics"]
JOBS = int(os.environ.get("KMX_BUILD_JOBS", str(min(8, os.cpu_count() or 4))))


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


def _headers_of(src: str) -> list[str]:
    """the headers a source depends on (the two kernel headers are each included by their own few translation units)"""
    out = []
    for h in HEADERS:
        base = os.path.basename(h)
        if base == "kmx_bitslice_kernel.h" and not (src.startswith("kmx_bitslice") or src == "kmx_sweep.hip"):
            continue
        if base in ("kmx_scan_kernel.h", "kmx_hist_part.h") and src not in ("kmx_scan.hip", "kmx_hist.hip", "kmx_hist32.hip"):
            continue
        if base == "kmx_hist_part.h" and src == "kmx_scan.hip":
            continue
        if os.path.exists(h):
            out.append(h)
    return out


def _split_usage(stderr: str) -> tuple[str, str]:
    """hipcc's stderr -> (one line per kernel: name and resources, everything that is not a resource remark)"""
    import re
    kernels, cur, other = [], None, []
    for ln in stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis=kernel-resource-usage\]", ln)
        if not m:
            # the source line and the caret clang prints under a remark, and the include stack above one in a header
            if re.match(r"^\s*\d*\s*\|", ln) or ln.startswith("In file included from") or not ln.strip():
                continue
            other.append(ln)
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = [body.split(":", 1)[1].strip()]
            kernels.append(cur)
        elif cur is not None:
            cur.append(body.strip())
    return "\n".join(" | ".join(k) for k in kernels) + "\n", "\n".join(other)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + _headers_of(src)):
        # -Rpass-analysis=kernel-resource-usage: registers / scratch / occupancy of every kernel, kept next to the object
        # (<src>.usage.txt; tests/test_kernel_resources.py reads them: a hot kernel that starts using scratch, or stops
        # inlining a lambda, shows there before it shows in a benchmark)
        cmd = [hipcc(), *CXXFLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        usage, other = _split_usage(r.stderr)
        with open(os.path.splitext(obj)[0] + ".usage.txt", "w") as f:
            f.write(usage)
        if other.strip():
            sys.stderr.write(other)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "flags.txt")
    flags_now = " ".join(CXXFLAGS)
    try:
        with open(stamp) as f:
            if f.read() != flags_now:
                force = True
    except OSError:
        force = True   # objects of unknown provenance (e.g. left by an older build script)
    with ThreadPoolExecutor(max_workers=JOBS) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    with open(stamp, "w") as f:
        f.write(flags_now)
    if force or _stale(LIB, objs):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs,
               "-Wl,-rpath,/opt/rocm/lib", "-ldl", f"-Wl,-soname,{os.path.basename(LIB)}"]   # (RCCL: dlopen at the first kmx_comm_* call)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
