#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6f
for k in 9 10 11 12; do python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 200 -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"; done > gpurun_out/r6f/k_small.txt
for spec in "161 93000000" "170 88000000" "176 85000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > gpurun_out/r6f/len_11.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r6f/pytest_full.txt 2>&1
tail -5 gpurun_out/r6f/pytest_full.txt
