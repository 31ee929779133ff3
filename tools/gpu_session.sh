python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "aligned" 2>&1 | tail -3
