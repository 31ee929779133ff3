#!/bin/bash
# scratch: the GPU session of the moment
bash tools/profile_round.sh gpurun_out/r05b > gpurun_out/r05b.log 2>&1
tail -3 gpurun_out/r05b.log
python3 tools/bench_line.py "[default]" < gpurun_out/r05b/bench_default.json
