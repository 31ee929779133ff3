#!/bin/bash
# dev tool: kernel-trace A/B of the second-pass machinery on clean input (main-pass duration and dispatch gaps)
R=$GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  python -c "from kmers_amd import build; build.build(force=True, extra=['-DKMX_BS_DIRTY=$v'])" >/dev/null 2>&1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abt$v -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
  python3 - <<PY
import csv, statistics
rows = list(csv.DictReader(open('$R/gpurun_out/abt$v/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
p0 = [r for r in rows if 'scan_bitsliced' in r['Kernel_Name'] and ', 1>' not in r['Kernel_Name']]
d = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in p0]
big = [x for x in d if x > 1.0][5:]
per = [(int(b['Start_Timestamp'])-int(a['Start_Timestamp']))/1e6 for a,b in zip(p0,p0[1:])]
per = [x for x in per if 1.0 < x < 10][5:]
print("dirty=$v  main kernel median %.3f ms  min %.3f | start-to-start median %.3f ms" % (statistics.median(big), min(big), statistics.median(per)))
PY
done
