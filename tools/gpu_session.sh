#!/bin/bash
# scratch: the GPU session of the moment
mkdir -p gpurun_out/r6d
timeout 1500 python3 -m pytest tests/test_gpu_round6.py -x -q -k minimizers 2>&1 | tail -15
python3 tools/bench_minimizers.py 2>&1 | tee gpurun_out/r6d/minimizers_bench.txt
