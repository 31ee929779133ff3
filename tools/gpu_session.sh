#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6s
for spec in "257 58000000" "280 53000000" "300 50000000" "320 47000000" "340 44000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > gpurun_out/r6s/len_seg11.txt
python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len 300 --reads-per-gpu 50000000 -k 21 2>/dev/null | python3 tools/bench_line.py "L=300 k=21" >> gpurun_out/r6s/len_seg11.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r6s/pytest_full.txt 2>&1
tail -2 gpurun_out/r6s/pytest_full.txt
