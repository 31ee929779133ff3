#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6v
timeout 1200 python3 -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "windows2_on_dirty" > gpurun_out/r6v/pytest_w2d.txt 2>&1
tail -4 gpurun_out/r6v/pytest_w2d.txt
