mkdir -p gpurun_out/r3j
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "comm_create or rccl" 2>&1 | tail -40 > gpurun_out/r3j/pytest_r3b.txt
