#!/bin/bash
# dev tool (GPU box): kernel trace + counter passes over tools/bench_fastq_parse.py (one --pmc group per run, no tracing flags mixed in)
# usage: tools/pmc_fastq.sh OUTDIR ["CTR_A CTR_B ..." ...]
cd /tmp && export TMPDIR=/tmp
out=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$out
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_fastq_parse.py 256 8 > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$i -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_fastq_parse.py 256 4 > $GRAFT_REPO_ROOT/$out/pmc_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $out > $out/summary.txt 2>&1
grep -A12 "fastx" $out/summary.txt | head -150
