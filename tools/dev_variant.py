"""Development variants of libkmx, built OUTSIDE the product tree.

    python tools/dev_variant.py NAME --only a.hip,b.hip [--sub 'REGEX=>TEXT' ...] [--patch FILE ...] [-DX=1 ...] [--keep-asm]

copies kmers_amd/csrc into tools/_variants/NAME/csrc, applies the substitutions (python `re.sub`, MULTILINE, on every
copied header and source; a substitution that matches nothing is an error) and unified diffs (`patch -p1` relative to
kmers_amd/csrc), compiles the sources named by --only from the copy with the product's flags, links them with the DEFAULT
objects of every other source (kmers_amd/csrc/_obj, built first if stale) into tools/_variants/NAME/libkmx.so and leaves each
compiled kernel's resource usage (and with --keep-asm the gfx950 assembly) next to it.

The product never looks for such a library: tools and tests pick one explicitly (`tools/devlib.py`, `pytest --kmx-lib PATH`).
tools/_variants/ is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmers_amd import build as kb  # noqa: E402


def main(argv: list[str]) -> str:
    name = argv[0]
    only: list[str] = []
    subs: list[tuple[str, str]] = []
    patches: list[str] = []
    extra: list[str] = []
    keep_asm = False
    i = 1
    while i < len(argv):
        a = argv[i]
        if a == "--only":
            only = argv[i + 1].split(",")
            i += 2
        elif a == "--sub":
            pat, _, rep = argv[i + 1].partition("=>")
            subs.append((pat, rep))
            i += 2
        elif a == "--patch":
            patches.append(os.path.abspath(argv[i + 1]))
            i += 2
        elif a == "--keep-asm":
            keep_asm = True
            i += 1
        else:
            extra.append(a)
            i += 1
    if not only:
        raise SystemExit("--only a.hip[,b.hip]: the sources the variant recompiles")
    kb.build()   # the default objects the variant links against
    vdir = os.path.join(ROOT, "tools", "_variants", name)
    csrc = os.path.join(vdir, "csrc")
    shutil.rmtree(vdir, ignore_errors=True)
    os.makedirs(csrc)
    for f in os.listdir(kb.CSRC):
        if f.endswith((".h", ".hip")):
            shutil.copy2(os.path.join(kb.CSRC, f), os.path.join(csrc, f))
    # (the sources say #include "../../include/kmx.h")
    os.makedirs(os.path.join(ROOT, "tools", "_variants", "include"), exist_ok=True)
    shutil.copy2(os.path.join(ROOT, "include", "kmx.h"), os.path.join(ROOT, "tools", "_variants", "include", "kmx.h"))
    for p in patches:
        subprocess.run(["patch", "-p1", "-i", p], cwd=csrc, check=True)
    for pat, rep in subs:
        hits = 0
        for f in os.listdir(csrc):
            path = os.path.join(csrc, f)
            with open(path) as fh:
                text = fh.read()
            new, n = re.subn(pat, rep, text, flags=re.MULTILINE)
            if n:
                hits += n
                with open(path, "w") as fh:
                    fh.write(new)
        if hits == 0:
            raise SystemExit(f"substitution matched nothing: {pat!r}")
        print(f"  sub {pat!r}: {hits} site(s)")
    def one(s: str) -> str:
        if s not in only:
            return os.path.join(kb.OBJ, os.path.splitext(s)[0] + ".o")
        obj = os.path.join(vdir, os.path.splitext(s)[0] + ".o")
        cmd = [kb.hipcc(), *kb.CXXFLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, s), "-o", obj]
        if keep_asm:
            cmd += ["-save-temps=obj"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=vdir)
        if r.returncode != 0:
            raise SystemExit(f"hipcc failed for {s}:\n{r.stderr}")
        usage, other = kb._split_usage(r.stderr)
        with open(os.path.splitext(obj)[0] + ".usage.txt", "w") as f:
            f.write(usage)
        if other.strip():
            sys.stderr.write(other + "\n")
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=kb.JOBS) as ex:
        objs = list(ex.map(one, kb.SOURCES))
    lib = os.path.join(vdir, "libkmx.so")
    cmd = [kb.hipcc(), "-shared", "-fPIC", f"--offload-arch={kb.ARCH}", "-o", lib, *objs, "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-Wl,-soname,libkmx.so"]
    subprocess.run(cmd, check=True)
    if keep_asm:   # keep the device assembly only
        for f in os.listdir(vdir):
            if f.endswith((".bc", ".hipi", ".hipfb", ".out", ".cui")) or (f.endswith((".s", ".o")) and "host" in f):
                os.remove(os.path.join(vdir, f))
    return lib


if __name__ == "__main__":
    print(main(sys.argv[1:]))
