KMX_FUZZ_N=3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "seqvec" 2>&1 | tail -6
