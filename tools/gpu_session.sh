#!/bin/bash
# scratch: the GPU session of the moment
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python3 tools/bench_small_hist.py 2>&1 | grep "n ="
python3 bench.py --config 4 --no-cpu-baseline --no-traffic --steps 5 --warmup 2 --sustain-steps 20 2>/dev/null | python3 tools/bench_line.py "[hist20]"
timeout 300 python3 tools/bench_hist.py 2>&1 | grep -v amdgpu | tail -12
timeout 300 python3 tools/bench_windows.py 2>&1 | grep -v amdgpu | tail -8
