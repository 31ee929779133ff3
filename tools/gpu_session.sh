#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6v
HIST=20 python3 tools/bench_dirty.py > gpurun_out/r6v/dirty_bench_final.txt 2>&1
python3 bench.py --config 4 --no-cpu-baseline --no-traffic --steps 5 --warmup 2 --sustain-steps 20 2>/dev/null | python3 tools/bench_line.py "[hist20]" > gpurun_out/r6v/hist20_final.txt
