"""dev tool: a coarse register-pressure curve of one kernel in a gfx950 assembly file (hipcc -S): for every N-th instruction the number
   of VGPRs whose first and last mention span it, with the nearest source landmark (barriers, sc1 accesses).
   usage: asm_pressure.py FILE.s KERNEL_SUBSTRING [STEP]"""
import re
import sys

path, want = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if want in l and l.split(":")[0].strip().startswith("_Z") and ":" in l and not l.startswith("\t"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = [l for l in lines[start:end] if l.strip() and not l.strip().startswith((";", "."))]
first, last = {}, {}
for i, l in enumerate(body):
    code = l.split(";")[0]
    for m in re.finditer(r"\bv(\d+)\b", code):
        r = int(m.group(1)); first.setdefault(r, i); last[r] = i
    for m in re.finditer(r"v\[(\d+):(\d+)\]", code):
        for r in range(int(m.group(1)), int(m.group(2)) + 1):
            first.setdefault(r, i); last[r] = i
print("instructions", len(body), "vgprs mentioned", len(first))
for i in range(0, len(body), step):
    live = sum(1 for r in first if first[r] <= i <= last[r])
    tags = [w for w in ("s_barrier", "sc1", "global_load_dwordx4", "ds_write_b128", "global_store_dwordx4", "s_sleep", "scratch_") if any(w in b for b in body[i:i + step])]
    print(f"{i:6d} {live:4d}  {' '.join(tags)}")
