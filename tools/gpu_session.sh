#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6s
timeout 900 python3 -m pytest tests/test_cpp_host_layer.py -m gpu -x -q > gpurun_out/r6s/pytest_cpp.txt 2>&1
tail -3 gpurun_out/r6s/pytest_cpp.txt
