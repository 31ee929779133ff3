"""dev tool: where a kernel touches scratch.  python tools/asm_loop_scratch.py FILE.s [SUBSTRING]
For every kernel whose mangled name holds SUBSTRING: the scratch loads / stores that lie inside a loop (in a block the assembly
labels "in Loop" / "Loop Header"), i.e. the ones executed per tile -- a reload there waits with s_waitcnt vmcnt(0) behind
everything in flight (the next tile's rows).  `scan()` is what tests/test_kernel_resources.py uses."""
import re
import sys


def scan(path, want=""):
    """{mangled kernel name: (reloads outside loops, reloads inside loops, stores inside loops)}"""
    name, in_loop = None, False
    hits = {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            name = m.group(1) if want in m.group(1) else None
            in_loop = False
            if name:
                hits[name] = [0, 0, 0]
            continue
        if name is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", ln) or "%bb." in ln:
            in_loop = "in Loop" in ln or "Loop Header" in ln
        elif ln.lstrip().startswith(";") and ("in Loop" in ln or "Loop Header" in ln):
            in_loop = True
        if "scratch_load" in ln:
            hits[name][1 if in_loop else 0] += 1
        if "scratch_store" in ln and in_loop:
            hits[name][2] += 1
        if "s_endpgm" in ln:
            name = None
    return {k: tuple(v) for k, v in hits.items()}


if __name__ == "__main__":
    for k, (out_l, in_l, in_s) in scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "").items():
        print(f"{re.sub(r'^_ZN3kmx[0-9]+', '', k)[:64]:66s} reloads outside loops {out_l:2d}  inside {in_l:2d}  stores inside {in_s:2d}")
