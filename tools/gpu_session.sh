#!/bin/bash
# scratch: the GPU session of the moment
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
KMX_FUZZ_N=20000 timeout 2400 python3 -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
bash tools/profile_round.sh gpurun_out/r05c > gpurun_out/r05c.log 2>&1
python3 tools/bench_line.py "[default]" < gpurun_out/r05c/bench_default.json
timeout 300 python3 tools/bench_small_batches.py 2>&1 | grep "n =" | tee gpurun_out/small_batches.txt
timeout 300 python3 tools/bench_fastq_pipeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_fastq_pipeline.txt
timeout 300 python3 tools/bench_dirty.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_dirty_bench.txt
