#!/bin/bash
# scratch: the GPU session of the moment
timeout 300 python3 tools/bench_small_hist.py 2>&1 | grep "n ="
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/smallh_trace -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_small_hist.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/smallh_trace/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last 14 kernels of each batch size: split by gen_reads kernels
idx=[i for i,r in enumerate(rows) if 'gen_reads' in r['Kernel_Name']]
idx.append(len(rows))
for a,b in zip(idx[:-1],idx[1:]):
    seg=rows[a:b][-12:]
    t0=int(seg[0]['Start_Timestamp'])
    print('--- batch')
    for r in seg:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        print(f"  +{(s-t0)/1e3:8.1f} us  dur {(e-s)/1e3:8.1f} us  grid {r['Grid_Size_X']:>9}  {r['Kernel_Name'][:80]}")
PY
