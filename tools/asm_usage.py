"""dev tool: per-kernel registers / scratch / occupancy out of a gfx950 assembly file kept by tools/dev_variant.py --keep-asm
    python tools/asm_usage.py FILE.s [substring of the mangled name]"""
import re
import sys

name = None
want = sys.argv[2] if len(sys.argv) > 2 else ""
vals = {}
for ln in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):\s", ln)
    if m:
        name = m.group(1)
        vals = {}
        continue
    m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|codeLenInByte)[ :=]+(\d+)", ln)
    if m and name:
        vals[m.group(1)] = int(m.group(2))
        if m.group(1) == "Occupancy":
            if want in name:
                short = re.sub(r"^_ZN3kmx\d+", "", name)[:70]
                print(f"{short:72s} vgpr {vals.get('NumVgprs', 0):3d} agpr {vals.get('NumAgprs', 0):3d} scratch {vals.get('ScratchSize', 0):4d} occ {vals['Occupancy']} code {vals.get('codeLenInByte', 0)}")
            name = None
