#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6m
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "minimizer or seqvec or fuzz" > gpurun_out/r6m/pytest_min.txt 2>&1
tail -2 gpurun_out/r6m/pytest_min.txt
python3 tools/bench_minimizers.py > gpurun_out/r6m/bench_after3.txt 2>&1
