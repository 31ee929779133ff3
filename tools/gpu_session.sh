#!/bin/bash
mkdir -p gpurun_out/s16
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s16/pytest_full.txt 2>&1
tail -3 gpurun_out/s16/pytest_full.txt
for k in 31 21 33 41 47 55 63 64; do python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 200 -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"; done | tee gpurun_out/s16/k.txt
python3 bench.py --packed --no-cpu-baseline --no-traffic --sustain-steps 200 2>/dev/null | python3 tools/bench_line.py "packed k=31"
python3 bench.py --packed -k 23 --no-cpu-baseline --no-traffic --sustain-steps 200 2>/dev/null | python3 tools/bench_line.py "packed k=23"
python3 tools/bench_ragged.py 100000000 31 2>/dev/null | head -2
python3 tools/bench_dirty.py 2>/dev/null | head -4
