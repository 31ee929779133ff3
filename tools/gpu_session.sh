mkdir -p gpurun_out/r3lr
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "rolled" 2>&1 | tail -8 > gpurun_out/r3lr/pytest2.txt
