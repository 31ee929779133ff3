#!/bin/bash
# scratch: the GPU session of the moment
mkdir -p gpurun_out/r6e
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
HIST=20 python3 tools/bench_dirty.py 2>&1 | tee gpurun_out/r6e/dirty_bench.txt
