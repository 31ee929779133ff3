#!/bin/bash
# dev tool: VALU instruction and co-issue counters of the bit-sliced kernel for several build variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  (cd $R && python -c "from kmers_amd import build; build.build(force=True, extra='$v'.split())" >/dev/null 2>&1)
  rm -rf $R/gpurun_out/pv; mkdir -p $R/gpurun_out/pv
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pv -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "== [$v]"; (cd $R && python3 tools/pmc_summary.py gpurun_out/pv | grep -A6 "scan_bitsliced" | head -7)
done
