#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6z
KMX_FUZZ_N=12000 timeout 2000 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r6z/fuzz12000.txt 2>&1
tail -2 gpurun_out/r6z/fuzz12000.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6z/bench_driver_cmd2.json 2> /dev/null
python3 tools/bench_line.py "[driver cmd, final build]" < gpurun_out/r6z/bench_driver_cmd2.json
python3 tools/bench_small_batches.py > gpurun_out/r6z/small_final.txt 2>&1
