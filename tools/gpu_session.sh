#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6sb
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r6sb/pytest_full.txt 2>&1
tail -5 gpurun_out/r6sb/pytest_full.txt
python3 tools/bench_small_batches.py > gpurun_out/r6sb/small6.txt 2>&1
python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 300 > gpurun_out/r6sb/bench6.json 2> /dev/null
