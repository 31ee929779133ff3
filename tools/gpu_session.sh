#!/bin/bash
mkdir -p gpurun_out/s13
timeout 1800 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "bench_two_ranks" > gpurun_out/s13/pytest.txt 2>&1
tail -30 gpurun_out/s13/pytest.txt
