"""dev tool: where a kernel touches scratch.  python tools/asm_loop_scratch.py FILE.s [SUBSTRING]
For every kernel whose mangled name holds SUBSTRING: the scratch loads / stores that lie inside a loop (in a block the assembly
labels "in Loop" / "Loop Header"), i.e. the ones executed per tile -- a reload there waits with s_waitcnt vmcnt(0) behind
everything in flight (the next tile's rows).  `scan()` is what tests/test_kernel_resources.py uses.
Round 6: a reload in the block that marks a dirty tile's reads -- recognised by its buffer stores, the only ones of the kernel:
mark_dirty_reads in kmx_bitslice_kernel.h -- is counted apart: a tile with an invalid byte pays it, a clean one does not."""
import re
import sys

DIRTY_SPAN = 140   # lines of assembly from the first LDS look-up of mark_dirty_reads to its stores (11..17 look-ups, the ballot)


def scan(path, want="", split_dirty=False):
    """{mangled kernel name: (reloads outside loops, reloads inside loops, stores inside loops)}; split_dirty: a fourth number, the
    reloads inside loops that sit within DIRTY_SPAN lines ahead of (or 8 behind) a buffer store, taken OUT of the second"""
    name, in_loop = None, False
    hits = {}
    lines = open(path).read().splitlines(True)
    marks = [i for i, ln in enumerate(lines) if "buffer_store_dwordx2" in ln] if split_dirty else []
    def dirty(i):
        return any(m - DIRTY_SPAN <= i <= m + 8 for m in marks)
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            name = m.group(1) if want in m.group(1) else None
            in_loop = False
            if name:
                hits[name] = [0, 0, 0, 0]
            continue
        if name is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", ln) or "%bb." in ln:
            in_loop = "in Loop" in ln or "Loop Header" in ln
        elif ln.lstrip().startswith(";") and ("in Loop" in ln or "Loop Header" in ln):
            in_loop = True
        if "scratch_load" in ln:
            hits[name][(3 if split_dirty and dirty(i) else 1) if in_loop else 0] += 1
        if "scratch_store" in ln and in_loop:
            hits[name][2] += 1
        if "s_endpgm" in ln:
            name = None
    return {k: tuple(v if split_dirty else v[:3]) for k, v in hits.items()}


if __name__ == "__main__":
    for k, (out_l, in_l, in_s, in_d) in scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", True).items():
        print(f"{re.sub(r'^_ZN3kmx[0-9]+', '', k)[:64]:66s} reloads outside loops {out_l:2d}  inside {in_l:2d} (+ {in_d} marking a dirty tile)  stores inside {in_s:2d}")
