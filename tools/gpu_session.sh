#!/bin/bash
# scratch: the GPU session of the moment
timeout 1500 python3 -m pytest tests/test_gpu_round6.py -x -q -k "histogram" 2>&1 | tail -5
timeout 2500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "dirty" 2>&1 | tail -5
timeout 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q 2>&1 | tail -3
