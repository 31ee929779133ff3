#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6w
for v in default w2ring8 default w2ring8; do echo "[$v]"; KMX_DEV_LIB=$v timeout 300 python3 tools/bench_windows2.py 10000000 150 2>/dev/null | grep -E "canon only" | head -5; done > gpurun_out/r6w/ring8.txt 2>&1
