"""dev tool: the device timeline of the LAST `ncalls` repetitions in a rocprofv3 --kernel-trace run: every kernel's start (relative to the first of its call), duration
    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 tools/bench_small_batches.py 31 150 100000;  python tools/trace_timeline.py OUT [n_last]"""
import csv, glob, os, re, sys
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("kmx::", "").replace("void ", "")
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), name))
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
prev_end = t0
for s, e, name in rows:
    print(f"start {(s - t0) / 1e3:9.1f} us   gap {(s - prev_end) / 1e3:7.1f}   dur {(e - s) / 1e3:8.1f} us   {name[:90]}")
    prev_end = e
