#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r6w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "WRITE_SIZE" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$i -o t -- python3 $R/tools/bench_windows2.py 10000000 150 > $out/pmc_$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $out > $out/summary.txt 2>&1
