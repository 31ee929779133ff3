#!/bin/bash
# dev tool (GPU box): rocprofv3 counter passes over a short bench run (one --pmc group per run, no tracing flags mixed in)
# usage: tools/pmc_pass.sh OUTDIR "CTR_A CTR_B ..." ["CTR_C ..."] ...     (KMX_PMC_BENCH_ARGS: extra bench.py arguments)
cd /tmp && export TMPDIR=/tmp
out=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$out
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$i -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --sustain-steps 0 $KMX_PMC_BENCH_ARGS > $GRAFT_REPO_ROOT/$out/pmc_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $out | grep -A14 "scan_bitsliced"
