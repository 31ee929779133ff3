"""dev tool: the full drains inside a kernel's loops.  python tools/asm_loop_waits.py FILE.s [SUBSTRING] [-v]
For every kernel whose mangled name holds SUBSTRING: each `s_waitcnt vmcnt(0)` that lies in a block the assembly labels "in Loop" /
"Loop Header", with the block it is in, the memory instruction ahead of it and the first instruction behind it that reads a
register -- on the tile loop's main path such a wait drains the next tile's rows (round 6: the ticket atomic's dead high register
was handed out behind one).  A wait inside a rare branch (the partial last row, a dirty tile, the roll) is fine;
which block is which is for the reader to judge: -v prints ten lines of context."""
import re
import sys


def scan(path, want=""):
    """{mangled name: [(line number, block label, previous vmem instruction, next instruction)]}"""
    lines = open(path).read().splitlines()
    name, in_loop, block = None, False, ""
    out = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            name = m.group(1) if want in m.group(1) else None
            in_loop, block = False, ""
            if name:
                out[name] = []
            continue
        if name is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            block = m.group(1)
            in_loop = "in Loop" in ln or "Loop Header" in ln
        elif ln.lstrip().startswith(";") and ("in Loop" in ln or "Loop Header" in ln):
            in_loop = True
        if in_loop and re.search(r"s_waitcnt\s+vmcnt\(0\)", ln):
            prev = next((lines[j].strip() for j in range(i - 1, max(i - 400, 0), -1)
                         if re.match(r"\s*(global_|buffer_|scratch_|flat_)", lines[j])), "?")
            nxt = next((lines[j].strip() for j in range(i + 1, min(i + 20, len(lines)))
                        if re.match(r"\s*[vsd]\w*_", lines[j]) and "s_waitcnt" not in lines[j] and "s_nop" not in lines[j]), "?")
            out[name].append((i + 1, block, prev.split(";")[0].strip(), nxt.split(";")[0].strip()))
        if "s_endpgm" in ln:
            name = None
    return out


if __name__ == "__main__":
    verbose = "-v" in sys.argv
    args = [a for a in sys.argv[1:] if a != "-v"]
    lines = open(args[0]).read().splitlines()
    for k, hits in scan(args[0], args[1] if len(args) > 1 else "").items():
        print(f"{re.sub(r'^_ZN3kmx[0-9]+', '', k)[:80]}: {len(hits)} x vmcnt(0) inside loops")
        for (n, block, prev, nxt) in hits:
            print(f"   line {n:6d} {block:12s} after [{prev[:60]}]  then [{nxt[:60]}]")
            if verbose:
                print("\n".join("        " + x for x in lines[max(n - 6, 0):n + 4]))
