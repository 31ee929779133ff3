mkdir -p gpurun_out/r3r
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "2_31 or thirteen" 2>&1 | tail -15 > gpurun_out/r3r/pytest_a.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3r/pytest_all.txt
