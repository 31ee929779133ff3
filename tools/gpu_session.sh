#!/bin/bash
mkdir -p gpurun_out/s11
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s11/pytest_full.txt 2>&1
tail -5 gpurun_out/s11/pytest_full.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/s11/b.json 2>/dev/null; wc -l gpurun_out/s11/b.json; python3 tools/bench_line.py default < gpurun_out/s11/b.json
