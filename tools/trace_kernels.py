"""dev tool: per-kernel durations out of a rocprofv3 --kernel-trace run, grouped by consecutive launches of the SAME call
    rocprofv3 --kernel-trace -d OUT -o t -- python3 tools/bench_dirty.py ...;  python tools/trace_kernels.py OUT [min_us]
prints, per kernel name (shortened), the number of launches and the median / min duration in microseconds"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
per = defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        name = re.sub(r"\(.*", "", name)
        name = name.replace("kmx::", "").replace("void ", "")
        per[name].append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
for name, v in sorted(per.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    us = sorted(x[1] for x in v)
    if us[len(us) // 2] < min_us and us[-1] < min_us:
        continue
    print(f"{len(us):6d} x  median {us[len(us)//2]:9.1f} us  min {us[0]:9.1f}  max {us[-1]:9.1f}   {name[:110]}")
