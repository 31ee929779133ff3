python -m pytest tests/test_gpu_fastx.py -x -q -m gpu -k "picks" 2>&1 | tail -3
