KMX_FUZZ_N=9000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "windows or histogram" 2>&1 | tail -4
