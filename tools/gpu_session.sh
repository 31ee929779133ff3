#!/bin/bash
# scratch: the GPU session of the moment
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python3 tools/bench_small_batches.py 31 150 1000,10000,30000,100000,1000000,4000000 2>&1 | grep "n ="
timeout 300 python3 tools/bench_small_hist.py 2>&1 | grep "n ="
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "[headline]"; done
