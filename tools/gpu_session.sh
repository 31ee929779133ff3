#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6v
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r6v/pytest_full4.txt 2>&1
tail -2 gpurun_out/r6v/pytest_full4.txt
python3 tools/bench_windows_dirty.py > gpurun_out/r6v/windows_dirty_after5.txt 2>&1
