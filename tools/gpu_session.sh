#!/bin/bash
# scratch: the GPU session of the moment
mkdir -p gpurun_out/r6c
R=$GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py -x -q 2>&1 | tail -3
python3 tools/bench_dirty.py 2>&1 | tee gpurun_out/r6c/dirty_bench.txt
cd /tmp && export TMPDIR=/tmp
for f in 0.001 0.005 0.02 0.1; do
  rm -rf /tmp/tr
  FRACS=$f rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 $R/tools/bench_dirty.py > /tmp/log 2>&1
  echo "== frac $f"; python3 $R/tools/trace_kernels.py /tmp/tr 2.0 | grep -E "scan_bits|sweep"
done 2>&1 | tee $R/gpurun_out/r6c/dirty_trace.txt
