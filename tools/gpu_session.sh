#!/bin/bash
# scratch: the GPU session of the moment
mkdir -p gpurun_out/r6a
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 0.0 0.001 0.02 0.1; do
  rm -rf /tmp/tr_$f
  FRACS=$f rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$f -o t -- python3 $R/tools/bench_dirty.py > /tmp/log_$f 2>&1
  echo "== frac $f"; python3 $R/tools/trace_kernels.py /tmp/tr_$f 2.0 | head -8
done 2>&1 | tee $R/gpurun_out/r6a/dirty_trace.txt
