mkdir -p gpurun_out/r3p
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "windows2" 2>&1 | tail -15 > gpurun_out/r3p/pytest_w2.txt
python -m pytest tests -x -q -m gpu -k "windows2 or reduce2 or two_word or win" 2>&1 | tail -5 > gpurun_out/r3p/pytest_w2_all.txt
python3 tools/bench_windows2.py > gpurun_out/r3p/windows2.txt 2>&1
