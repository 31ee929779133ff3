#!/bin/bash
mkdir -p gpurun_out/s7
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s7/pytest_full.txt 2>&1
tail -12 gpurun_out/s7/pytest_full.txt
