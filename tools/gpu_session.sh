mkdir -p gpurun_out/r3x
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r3x/pytest.txt
python3 tools/bench_hist.py 100000000 20,24 > gpurun_out/r3x/hist.txt 2>&1
