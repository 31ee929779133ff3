mkdir -p gpurun_out/r3fz
KMX_FUZZ_N=6000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3fz/fuzz.txt
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3fz/pytest.txt
