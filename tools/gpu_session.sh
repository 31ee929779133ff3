#!/bin/bash
# scratch: the GPU session of the moment
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python3 tools/bench_dirty.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_dirty_bench.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "[headline]"
timeout 300 python3 tools/bench_long_ragged.py 2>&1 | grep -v amdgpu | tail -8
