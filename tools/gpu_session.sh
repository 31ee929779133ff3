#!/bin/bash
# scratch: the GPU session of the moment
bash tools/profile_round.sh gpurun_out/r05e > gpurun_out/r05e.log 2>&1
python3 tools/bench_line.py "[default]" < gpurun_out/r05e/bench_default.json
timeout 300 python3 tools/bench_fastq_pipeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_fastq_pipeline.txt
