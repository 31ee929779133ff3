#!/bin/bash
# dev tool (GPU box), round 6: counter passes over tools/bench_dirty.py at ONE share of dirty reads -- what the scan and the sweep
# fetch and issue per call (one --pmc group per run, no tracing flags mixed in).   usage: tools/pmc_dirty.sh OUTDIR FRAC "CTRS" ["CTRS" ...]
cd /tmp && export TMPDIR=/tmp
out=$1; frac=$2; shift; shift
mkdir -p $GRAFT_REPO_ROOT/$out
i=0
for grp in "$@"; do
  i=$((i+1))
  export FRACS=$frac
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$i -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_dirty.py > $GRAFT_REPO_ROOT/$out/pmc_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $out | grep -E "==|scan_bitsliced|sweep_flagged|      "
