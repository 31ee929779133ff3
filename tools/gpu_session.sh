mkdir -p gpurun_out/r3fz
timeout 900 python -m pytest tests/test_gpu_fastx.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3fz/fastx.txt
