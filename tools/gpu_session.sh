#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6v
timeout 1500 python3 -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "windows_on_dirty" > gpurun_out/r6v/pytest_wd.txt 2>&1
tail -5 gpurun_out/r6v/pytest_wd.txt
