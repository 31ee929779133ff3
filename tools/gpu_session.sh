# scratch: the command list of the last gpurun call (round 3, run 14: two-level histogram with the mid-block flush)
mkdir -p gpurun_out/r3n
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "histogram" 2>&1 | tail -5 > gpurun_out/r3n/pytest_hist.txt
python3 tools/bench_hist.py 100000000 20,23,24,26,28 > gpurun_out/r3n/hist_bench.txt 2>&1
